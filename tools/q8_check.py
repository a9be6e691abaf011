"""8-bit precisions (FP8 e4m3 / INT8) under the discriminating weights on the GPU: de-meaned error of the HIP networks vs the torch
fp32 networks as a fraction of the between-hypothesis spread, refined-pose distances, Register winner -- and what each costs.
   python tools/q8_check.py [W H]"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_FP8, FP_PREC_INT8
from oracle import fp_oracle as fo, nets_torch as NT

Wd, Hd = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
mesh = syn.make_mesh(textured=os.environ.get("UNTEXTURED") is None); scene = syn.make_scene(mesh, W=Wd, H=Hd); om = fo.OracleMesh(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
rn = NT.build("refiner", W.pack_synthetic("refiner", rp, 9, cal)); sn = NT.build("scorer", W.pack_synthetic("scorer", sp, 9, cal))
model = FoundationPose(mesh, scene.K, rp, sp)
model.upload_frame(scene.rgb, scene.depth)
poses = model.get_hyp_poses(scene.mask)
def dm(x): return x - x.mean(0, keepdims=True)
sel = poses[::6]
a12, b12 = model.render_and_transform(mesh.name, sel, 1.2)
a11, b11 = model.render_and_transform(mesh.name, sel, 1.1)
with torch.no_grad():
    rt, rr = rn(torch.from_numpy(a12), torch.from_numpy(b12)); rt, rr = rt.numpy(), rr.numpy()
    ss = sn(torch.from_numpy(a11), torch.from_numpy(b11)).numpy()
def stage(name):
    t, r = model.refiner_infer(a12, b12)
    out = {}
    for nm, x, y in (("trans", t, rt), ("rot", r, rr)):
        sp_ = y.std(0)
        de = dm(x) - dm(y)
        print(f"{name} {nm}: common-mode/spread {np.abs((x - y).mean(0) / sp_).max():.2f}  de-meaned rms/spread {(np.sqrt((de**2).mean(0)) / sp_).max()*100:.1f}%  max {(np.abs(de).max(0)/sp_).max()*100:.1f}%  corr min {min(np.corrcoef(x[:, j], y[:, j])[0, 1] for j in range(3)):.4f}")
    dt = np.linalg.norm(t - rt, axis=1) * mesh.diameter / 2 * 1e3
    dr = np.degrees(np.linalg.norm((np.tanh(r) - np.tanh(rr)) * 0.349065850398865, axis=1))
    print(f"{name} refiner delta error: mm p95 {np.percentile(dt, 95):.3f} max {dt.max():.3f}; deg p95 {np.percentile(dr, 95):.3f} max {dr.max():.3f}")
    s = model.scorer_infer(a11, b11)
    print(f"{name} score: de-meaned rms/spread {np.sqrt(((dm(s) - dm(ss))**2).mean()) / ss.std() * 100:.1f}% max {np.abs(dm(s)-dm(ss)).max()/ss.std()*100:.1f}% corr {np.corrcoef(s, ss)[0,1]:.4f}; argmax hip {s.argmax()} torch {ss.argmax()} rank {list(np.argsort(-ss)).index(int(s.argmax()))}")
def timed(n=5):
    model.Register(scene.rgb, scene.depth, scene.mask, mesh.name); model.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    t0 = time.time()
    for _ in range(n): model.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    return (time.time() - t0) / n * 1e3
stage("f16")
ok, pose, idx16, sc16, ref16, _ = model.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
print(f"f16 Register {timed():.2f} ms (host frames)")
for prec, name in ((FP_PREC_INT8, "int8"), (FP_PREC_FP8, "fp8")):
    model.set_precision(FP_PREC_F16)
    t0 = time.time()
    model.calibrate(scene.rgb, scene.depth, scene.mask, mesh.name, prec)
    print(f"{name}: calibration {time.time() - t0:.2f} s")
    model.set_precision(prec)
    stage(name)
    ok, pose, idx, sc, refined, feats = model.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok, model.last_error
    dmm = np.linalg.norm(refined[:, :3, 3] - ref16[:, :3, 3], axis=1) * 1e3
    dR = np.einsum("nij,nkj->nik", refined[:, :3, :3], ref16[:, :3, :3])
    ddeg = np.degrees(np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1)))
    print(f"{name} Register: winner {idx} (f16 {idx16}); rank of winner in f16 order {list(np.argsort(-sc16)).index(idx)}; refined poses vs f16: mm p95 {np.percentile(dmm,95):.3f} max {dmm.max():.3f} deg p95 {np.percentile(ddeg,95):.3f} max {ddeg.max():.3f}; within 1mm/1deg: {np.mean((dmm < 1) & (ddeg < 1))*100:.1f}%; score corr vs f16 {np.corrcoef(sc, sc16)[0,1]:.4f}")
    print(f"{name} Register {timed():.2f} ms (host frames)")
model.close()
