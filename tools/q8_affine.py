"""Experiment: can a per-hypothesis AFFINE output correction of the INT8 refiner (fitted on the calibration frames' hypotheses) predict the
scene-dependent part of its error on held-out frames?  Same input blobs through the f16 and the INT8 refiner; error e = out16 - out8 [6];
ridge regression of e on feature sets F of the calibration hypotheses; residual on held-out hypotheses.
   python tools/q8_affine.py"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W  # noqa: E402
from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_INT8  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
mesh = syn.make_mesh()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp, 9, cal); W.pack_synthetic("scorer", sp, 9, cal)
m = FoundationPose(mesh, syn.intrinsics(), rp, sp)
calib = syn.calibration_scenes(mesh, 16)
held = syn.heldout_scenes(mesh, 8)
m.calibrate_frames(calib, mesh.name, FP_PREC_INT8)


def outs(sc):
    m.set_precision(FP_PREC_F16)
    m.upload_frame(sc.rgb, sc.depth)
    poses = m.get_hyp_poses(sc.mask)
    a, b = m.render_and_transform(mesh.name, poses, 1.2)
    t16, r16 = m.refiner_infer(a, b)
    m.set_precision(FP_PREC_INT8)
    t8, r8 = m.refiner_infer(a, b)
    stats = np.concatenate([a.mean((1, 2)), b.mean((1, 2)), (a[..., 3:] != 0).mean((1, 2)), (b[..., 3:] != 0).mean((1, 2))], 1)   # [N, 18]
    return np.concatenate([t16, r16], 1), np.concatenate([t8, r8], 1), stats


C = [outs(s) for s in calib]
Hd = [outs(s) for s in held]
o16c, o8c, stc = (np.concatenate([c[i] for c in C]) for i in range(3))
scale_mm = mesh.diameter / 2 * 1e3


def feats(o8, st, kind):
    one = np.ones((len(o8), 1), np.float32)
    if kind == "bias": return one
    if kind == "out8": return np.concatenate([o8, one], 1)
    if kind == "out8+stats": return np.concatenate([o8, st, one], 1)
    if kind == "out8^2": return np.concatenate([o8, o8[:, :, None].repeat(6, 2).reshape(len(o8), -1) * np.tile(o8, (1, 6)), st, one], 1)
    raise ValueError(kind)


for kind in ("bias", "out8", "out8+stats", "out8^2"):
    F = feats(o8c, stc, kind).astype(np.float64)
    E = (o16c - o8c).astype(np.float64)
    lam = 1e-3 * np.trace(F.T @ F) / F.shape[1]
    A = np.linalg.solve(F.T @ F + lam * np.eye(F.shape[1]), F.T @ E)
    line = []
    for (o16, o8, st) in Hd:
        e0 = (o16 - o8)
        e1 = e0 - feats(o8, st, kind).astype(np.float64) @ A
        f = lambda e: (np.mean(np.linalg.norm(e[:, :3], axis=1) * scale_mm < 1.0) * 100, np.linalg.norm(e[:, :3].mean(0)) * scale_mm)
        line.append(f"{f(e0)[0]:5.1f}->{f(e1)[0]:5.1f}% cm {f(e0)[1]:.2f}->{f(e1)[1]:.2f}")
    ec = E - F @ A
    print(f"{kind:12s} (fit residual rms {np.sqrt((ec[:, :3] ** 2).mean()) * scale_mm:.3f} mm of {np.sqrt((E[:, :3] ** 2).mean()) * scale_mm:.3f}): " + " | ".join(line), flush=True)
m.close()
