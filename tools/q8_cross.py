"""INT8 trunk variants (test build): same-frame and CROSS-frame refined-pose error against the f16 path, and the Register time.
   python tools/q8_cross.py   ->  one line per setting of the test hook g_i8_stream (0 = f16 residual stream, the product; 1 = 8-bit stream)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_INT8
import tempfile
L = _lib.lib()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
mesh = syn.make_mesh()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp, 9, cal); W.pack_synthetic("scorer", sp, 9, cal)
s1 = syn.make_scene(mesh)
scenes = {"same": s1, "cross": syn.make_scene(mesh, t=(-0.03, 0.02, 0.62), rot_seed=9), "cross2": syn.make_scene(mesh, t=(0.04, -0.03, 0.80), rot_seed=4)}
def rot_deg(a, b):
    dR = np.einsum("nij,nkj->nik", a[:, :3, :3].astype(np.float64), b[:, :3, :3].astype(np.float64))
    return np.degrees(np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1)))
settings = [int(a) for a in sys.argv[1:]] or [0, 1, 0, 1]
for mode in settings:
    L.fpt_set_i8_stream(mode)
    m = FoundationPose(mesh, syn.intrinsics(), rp, sp)
    m.calibrate(s1.rgb, s1.depth, s1.mask, mesh.name, FP_PREC_INT8)
    out = []
    for nm, sc in scenes.items():
        m.set_precision(FP_PREC_F16)
        ok, p16, i16, s16, r16, _ = m.register_detailed(sc.rgb, sc.depth, sc.mask, mesh.name)
        m.set_precision(FP_PREC_INT8)
        ok, p8, i8, s8, r8, _ = m.register_detailed(sc.rgb, sc.depth, sc.mask, mesh.name)
        dmm = np.linalg.norm(r8[:, :3, 3] - r16[:, :3, 3], axis=1) * 1e3; dd = rot_deg(r8, r16)
        cm = np.linalg.norm((r8[:, :3, 3] - r16[:, :3, 3]).mean(0)) * 1e3
        out.append(f"{nm}: {np.mean((dmm < 1) & (dd < 1)) * 100:5.1f}% p95 {np.percentile(dmm, 95):.2f} mm (common-mode {cm:.2f}) {np.percentile(dd, 95):.2f} deg")
    for _ in range(3): m.Register(s1.rgb, s1.depth, s1.mask, mesh.name)
    t0 = time.perf_counter()
    for _ in range(20): m.Register(s1.rgb, s1.depth, s1.mask, mesh.name)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"8-bit residual stream {mode}: " + " | ".join(out) + f" | Register {ms:.2f} ms (640x480, host frames)", flush=True)
    m.close()
