"""Multi-frame INT8 / FP8 calibration (fp_calibrate_begin / _add_frame / _finish) measured on HELD-OUT scenes: refined poses of the 8-bit
path against the f16 path of the same model, per held-out scene (share within 1 mm / 1 deg, p95, common-mode shift), for K = 1, 4, 8, 16
calibration scenes.
   python tools/q8_multi.py [W H] [--prec int8|fp8] [--ks 1,4,8,16]"""
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
ap.add_argument("size", nargs="*", type=int, default=[640, 480])
ap.add_argument("--prec", default="int8")
ap.add_argument("--ks", default="1,4,8,16")
ap.add_argument("--untextured", action="store_true")
ap.add_argument("--opts", default="", help="test build: sweeps,tok,out of the calibration (default 2,1,1)")
ap.add_argument("--headroom", type=float, default=0.0, help="test build: INT8 scale = |max| * headroom / 255 (default 1.25)")
ap.add_argument("--wq", default="", help="test build: wclip,efr,imgbias switches of the INT8 weight quantiser / per-image compensation (default 1,2,1)")
ap.add_argument("--amax", action="store_true", help="per-activation |max| of every held-out scene relative to the calibration record")
args = ap.parse_args()
if args.opts or args.headroom or args.wq:
    import ctypes
    _lib.use_test_lib()
    L = _lib.lib()
    if args.opts:
        L.fpt_set_calib_opts(*[int(x) for x in args.opts.split(",")])
    if args.wq:
        L.fpt_set_q8_wq(*[int(x) for x in args.wq.split(",")])
    if args.headroom:
        L.fpt_set_q8_headroom.argtypes = [ctypes.c_float]
        L.fpt_set_q8_headroom(args.headroom)
Wd, H = args.size
PREC = FP_PREC_INT8 if args.prec == "int8" else FP_PREC_FP8
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
mesh = syn.make_mesh(textured=not args.untextured)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp, 9, cal)
W.pack_synthetic("scorer", sp, 9, cal)


def rot_deg(a, b):
    dR = np.einsum("nij,nkj->nik", a[:, :3, :3].astype(np.float64), b[:, :3, :3].astype(np.float64))
    return np.degrees(np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1)))


calib = syn.calibration_scenes(mesh, max(16, max(int(x) for x in args.ks.split(","))), W=Wd, H=H)
held = syn.heldout_scenes(mesh, int(os.environ.get("HELD", "4")), W=Wd, H=H)
if (Wd, H) == (640, 480):      # the two "other scenes" of round 4's cross-frame test (same noise / background seeds as its calibration scene)
    held += [syn.make_scene(mesh, t=(-0.03, 0.02, 0.62), rot_seed=9), syn.make_scene(mesh, t=(0.04, -0.03, 0.80), rot_seed=4)]


def amax_of(model, scenes):
    model.set_precision(FP_PREC_F16)
    model.calibrate_frames(scenes, mesh.name, PREC)
    return np.frombuffer(model.get_calibration_blob(PREC), np.float32, 2 * 15 * 512, 16).reshape(2, 15, 512).copy()
m = FoundationPose(mesh, syn.intrinsics(Wd, H), rp, sp)
ref = {}
for k, sc in enumerate(held):
    ok, p16, i16, s16, r16, _ = m.register_detailed(sc.rgb, sc.depth, sc.mask, mesh.name)
    assert ok, m.last_error
    ref[k] = (r16, i16, s16)
for K in [int(x) for x in args.ks.split(",")]:
    m.set_precision(FP_PREC_F16)
    t0 = time.perf_counter()
    m.calibrate_frames(calib[:K], mesh.name, PREC)
    t_cal = time.perf_counter() - t0
    if args.amax:
        a_cal = np.frombuffer(m.get_calibration_blob(PREC), np.float32, 2 * 15 * 512, 16).reshape(2, 15, 512).copy()
        for k, sc in enumerate(held):
            a = amax_of(m, [sc])
            r = a / np.maximum(a_cal, 1e-9)
            live = a_cal > a_cal.max(axis=2, keepdims=True) / 1024
            print(f"  held {k}: |max| ratio scene / record per activation (refiner): " + " ".join(f"{np.percentile(r[0, i][live[0, i]], 99):.2f}" for i in range(1, 14)) +
                  f" | share of live channels over the record: {np.mean(r[0][live[0]] > 1.0) * 100:.1f} %, over 1.25x: {np.mean(r[0][live[0]] > 1.25) * 100:.1f} %", flush=True)
        m.set_precision(FP_PREC_F16)
        m.calibrate_frames(calib[:K], mesh.name, PREC)
    m.set_precision(PREC)
    out = []
    for k, sc in enumerate(held):
        ok, p8, i8, s8, r8, _ = m.register_detailed(sc.rgb, sc.depth, sc.mask, mesh.name)
        assert ok, m.last_error
        r16, i16, s16 = ref[k]
        dmm = np.linalg.norm(r8[:, :3, 3] - r16[:, :3, 3], axis=1) * 1e3
        dd = rot_deg(r8, r16)
        cm = np.linalg.norm((r8[:, :3, 3] - r16[:, :3, 3]).mean(0)) * 1e3
        out.append(f"{np.mean((dmm < 1) & (dd < 1)) * 100:5.1f}% p95 {np.percentile(dmm, 95):.2f} max {dmm.max():.2f} mm cm {cm:.2f} | {np.percentile(dd, 95):.2f} deg win {i8}/{i16}")
    print(f"{args.prec} {Wd}x{H} K={K:2d} ({t_cal:.1f} s): " + " || ".join(out), flush=True)
m.close()
