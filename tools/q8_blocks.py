"""[r6] Which STAGES of the trunk can run on 8-bit operands and still hold the configs[4] bar on frames the calibration never saw?
  Test build: fpt_set_q8_blocks(mask) -- bit 0 encodeA.2-3 (128 ch), bit 1 encodeAB.0-1 (256 ch), bit 2 encodeAB.2
(3x3 / s2), bit 3 encodeAB.3-4 (512 ch); stages outside the mask keep their f16 weights and kernels (fp_nn.hip: run_trunk_q8).

Per mask and mesh (textured / untextured), 1280x720, N = 252: calibrate on 16 frames of the scene family, measure on 6 OTHER frames
against the f16 path of the same model: share of the 252 refined poses within 1 mm / 1 deg, p95, common-mode shift, winner index;
and the Register wall time (host frames) next to f16's.  STRICT bar: every scene >= 95 % and common mode < 0.3 mm.

   python tools/q8_blocks.py [--masks 15,2,8,10] [--prec int8|fp8] [--held 6] [--size 1280 720] [--weights disc|RECORD.npz]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, _lib, synthetic as syn, weights as W  # noqa: E402
from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_FP8, FP_PREC_INT8  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--masks", default="15,2,8,1,4,10,3,11,14")
ap.add_argument("--prec", default="int8")
ap.add_argument("--held", type=int, default=6)
ap.add_argument("--k", type=int, default=16)
ap.add_argument("--size", nargs=2, type=int, default=[1280, 720])
ap.add_argument("--meshes", default="textured,untextured")
ap.add_argument("--weights", default="disc", help="disc = the discriminating synthetic set; a path = a calibration record (.npz) applied instead, e.g. the output of oracle/fit_readouts.py")
args = ap.parse_args()
_lib.use_test_lib()
L = _lib.lib()
Wd, H = args.size
PREC = FP_PREC_INT8 if args.prec == "int8" else FP_PREC_FP8
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
if args.weights != "disc":
    cal = W.load_calibration(args.weights)
W.pack_synthetic("refiner", rp, 9, cal)
W.pack_synthetic("scorer", sp, 9, cal)
NAMES = {1: "128", 2: "256", 4: "b2", 8: "512"}


def rot_deg(a, b):
    dR = np.einsum("nij,nkj->nik", a[:, :3, :3].astype(np.float64), b[:, :3, :3].astype(np.float64))
    return np.degrees(np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1)))


def wall_ms(m, sc, name, n=8):
    for _ in range(3):
        m.Register(sc.rgb, sc.depth, sc.mask, name)
    t0 = time.perf_counter()
    for _ in range(n):
        m.Register(sc.rgb, sc.depth, sc.mask, name)
    return (time.perf_counter() - t0) / n * 1e3


for mesh_kind in args.meshes.split(","):
    mesh = syn.make_mesh(textured=mesh_kind == "textured")
    calib = syn.calibration_scenes(mesh, args.k, W=Wd, H=H)
    held = syn.heldout_scenes(mesh, args.held, W=Wd, H=H)
    m = FoundationPose(mesh, syn.intrinsics(Wd, H), rp, sp)
    ref = []
    for sc in held:
        ok, p16, i16, s16, r16, _ = m.register_detailed(sc.rgb, sc.depth, sc.mask, mesh.name)
        assert ok, m.last_error
        ref.append((r16, i16, p16))
    t16 = wall_ms(m, held[0], mesh.name)
    m.close()
    print(f"== {mesh_kind} {Wd}x{H} {args.prec} weights={args.weights}: f16 Register {t16:.2f} ms (host frames)", flush=True)
    for mask in [int(x) for x in args.masks.split(",")]:
        L.fpt_set_q8_blocks(mask)
        m = FoundationPose(mesh, syn.intrinsics(Wd, H), rp, sp)
        try:
            t0 = time.perf_counter()
            m.calibrate_frames(calib, mesh.name, PREC)
            t_cal = time.perf_counter() - t0
            m.set_precision(PREC)
            rows = []
            for k, sc in enumerate(held):
                ok, p8, i8, s8, r8, _ = m.register_detailed(sc.rgb, sc.depth, sc.mask, mesh.name)
                assert ok, m.last_error
                r16, i16, p16 = ref[k]
                dmm = np.linalg.norm(r8[:, :3, 3] - r16[:, :3, 3], axis=1) * 1e3
                dd = rot_deg(r8, r16)
                cm = np.linalg.norm((r8[:, :3, 3] - r16[:, :3, 3]).mean(0)) * 1e3
                wmm = np.linalg.norm(p8[:3, 3] - p16[:3, 3]) * 1e3
                wdeg = rot_deg(p8[None], p16[None])[0]
                rows.append(dict(frac=float(np.mean((dmm < 1) & (dd < 1))), p95=float(np.percentile(dmm, 95)), mx=float(dmm.max()), cm=float(cm),
                                 d95=float(np.percentile(dd, 95)), same=i8 == i16, wmm=float(wmm), wdeg=float(wdeg)))
            t8 = wall_ms(m, held[0], mesh.name)
        finally:
            m.close()
        frac = np.array([r["frac"] for r in rows]); cmv = np.array([r["cm"] for r in rows])
        strict = bool((frac >= 0.95).all() and (cmv < 0.3).all())
        stages = "+".join(NAMES[b] for b in (1, 2, 4, 8) if mask & b)
        print(f"mask {mask:2d} [{stages:14s}] {t8:6.2f} ms ({(1 - t8 / t16) * 100:+5.1f} % vs f16; calib {t_cal:.1f} s)  share mean {frac.mean() * 100:5.1f} min {frac.min() * 100:5.1f}  "
              f"cm mean {cmv.mean():.2f} max {cmv.max():.2f} mm  p95 max {max(r['p95'] for r in rows):.2f} mm / {max(r['d95'] for r in rows):.2f} deg  "
              f"same winner {sum(r['same'] for r in rows)}/{len(rows)}  STRICT {'PASS' if strict else 'fail'}", flush=True)
        print("      per scene: " + " | ".join(f"{r['frac'] * 100:5.1f}% cm {r['cm']:.2f} p95 {r['p95']:.2f} w {r['wmm']:.1f}mm/{r['wdeg']:.1f}d" for r in rows), flush=True)
L.fpt_set_q8_blocks(15)
