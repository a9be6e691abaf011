"""Measure what the discriminating weight set (tests/golden/disc_calib_seed9.npz) is worth on the GPU: spread of the oracle's
outputs over the hypotheses vs the HIP path's deviation from the oracle, per precision.  Sets the tolerances of
tests/test_discriminative_gpu.py.   python tools/disc_noise.py [n_register]"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_BF16, FP_PREC_FP8
from oracle import fp_oracle as fo, nets_torch as NT

cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
mesh = syn.make_mesh(); scene = syn.make_scene(mesh); om = fo.OracleMesh(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
rn = NT.build("refiner", W.pack_synthetic("refiner", rp, 9, cal)); sn = NT.build("scorer", W.pack_synthetic("scorer", sp, 9, cal))
model = FoundationPose(mesh, scene.K, rp, sp)
model.upload_frame(scene.rgb, scene.depth)
poses = model.get_hyp_poses(scene.mask)
def dm(x): return x - x.mean(0, keepdims=True)
def stage(prec, name):
    model.set_precision(prec)
    sel = poses[::6]
    a, b = model.render_and_transform(mesh.name, sel, 1.2)
    t, r = model.refiner_infer(a, b)
    with torch.no_grad(): rt, rr = rn(torch.from_numpy(a), torch.from_numpy(b))
    rt, rr = rt.numpy(), rr.numpy()
    for nm, x, y in (("trans", t, rt), ("rot", r, rr)):
        print(f"{name} {nm}: torch mean {np.abs(y.mean(0)).max():.4f} spread(std) {y.std(0).min():.4f}  |hip-torch| max {np.abs(x-y).max():.2e}  de-meaned max {np.abs(dm(x)-dm(y)).max():.2e}  = {np.abs(dm(x)-dm(y)).max()/y.std(0).min()*100:.2f}% of spread")
    a, b = model.render_and_transform(mesh.name, sel, 1.1)
    s = model.scorer_infer(a, b)
    with torch.no_grad(): ss = sn(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    o = np.sort(ss)[::-1]
    print(f"{name} score: torch std {ss.std():.3f} range {np.ptp(ss):.3f} gap {o[0]-o[1]:.3f} |hip-torch| max {np.abs(s-ss).max():.2e} de-meaned {np.abs(dm(s)-dm(ss)).max():.2e} = {np.abs(dm(s)-dm(ss)).max()/ss.std()*100:.2f}% of spread; argmax hip {s.argmax()} torch {ss.argmax()}")
stage(FP_PREC_F16, "f16")
stage(FP_PREC_BF16, "bf16")
model.set_precision(FP_PREC_F16)
model.calibrate_fp8(scene.rgb, scene.depth, scene.mask, mesh.name)
stage(FP_PREC_FP8, "fp8")
# full Register
t0 = time.time()
p0 = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)
def onets(net, p, ratio):
    a = fo.render(om, p, scene.K, scene.depth.shape, ratio); b = fo.crop(scene.rgb, scene.depth, scene.K, p, ratio, mesh.diameter)
    with torch.no_grad(): return net(torch.from_numpy(a), torch.from_numpy(b))
t, r = onets(rn, p0, 1.2)
ref = fo.refine_post_process(p0, t.numpy(), r.numpy(), mesh.diameter)
os_ = onets(sn, ref, 1.1).numpy()
print("oracle register %.1fs: score std %.3f top3 %s gaps %s" % (time.time() - t0, os_.std(), np.argsort(-os_)[:3], -np.diff(np.sort(os_)[::-1][:4])))
for prec, name in ((FP_PREC_F16, "f16"), (FP_PREC_BF16, "bf16"), (FP_PREC_FP8, "fp8")):
    model.set_precision(prec)
    ok, pose, idx, sc, refined, feats = model.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok, model.last_error
    dpos = np.abs(syn.to_colmajor(refined) - ref).max()
    print(f"register {name}: winner {idx} (oracle {os_.argmax()}), rank of hip winner in oracle order {list(np.argsort(-os_)).index(idx)}, |scores-oracle| de-meaned max {np.abs(dm(sc)-dm(os_)).max():.3e} ({np.abs(dm(sc)-dm(os_)).max()/os_.std()*100:.2f}% of spread), refined pose max|d| {dpos:.2e}, spearman {np.corrcoef(np.argsort(np.argsort(sc)), np.argsort(np.argsort(os_)))[0,1]:.4f}")
model.close()
