"""enc_tail_kernel (the row-wise tail of both refiner heads in one launch) against the five-launch form: outputs and time.
    python tools/ab_enc_tail.py
Uses the test library (fpt_set_enc_tail); discriminating weights so that differences show."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device='cuda')   # (torch's runtime must see the GPU before the library's does)
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
_lib.use_test_lib()
L = _lib.lib()
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
from foundationpose_cpp_amd.weights import load_calibration
cal = load_calibration(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "disc_calib_seed9.npz"))
W.pack_synthetic("refiner", rp, 9, cal); W.pack_synthetic("scorer", sp, 9, cal)   # the discriminating set of tests/conftest.py
m = FoundationPose(mesh, scene.K, rp, sp)
m.upload_frame(scene.rgb, scene.depth)
poses = m.get_hyp_poses(scene.mask)
for step in (6, 1):
    ps = np.stack([syn.perturb_pose(p, deg=3.0, trans=0.006, seed=100 + i) for i, p in enumerate(poses[::step])])
    a, b = m.render_and_transform(mesh.name, ps, 1.2)
    out = {}
    for v in (0, 1, 0, 1):
        L.fpt_set_enc_tail(v)
        out.setdefault(v, []).append(m.refiner_infer(a, b))
    for v in (0, 1):
        assert all(np.array_equal(x, y) for x, y in zip(out[v][0], out[v][1])), f"enc_tail={v}: not reproducible"
    for name, x, y in zip(("trans", "rot"), out[0][0], out[1][0]):
        sp_ = x.std(0)
        print(f"N={len(ps)} {name}: spread {sp_}, max |fused - five launches| / spread {np.abs(x - y).max(0) / sp_}, rms {np.sqrt(((x - y) ** 2).mean(0)) / sp_}, nan {np.isnan(y).any()}")
for v in (0, 1, 0, 1):
    L.fpt_set_enc_tail(v)
    for _ in range(2):
        m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    m.profile(True); m.profile_reset()
    for _ in range(6):
        m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    r = m.profile_report(); m.profile(False)
    st = {}
    for k, x in r.items():
        st[k.split("/")[0]] = st.get(k.split("/")[0], 0.0) + x["ms"] / 6
    keys = ("gemm_qkv", "attention", "gemm_512", "layernorm", "layernorm_mean", "enc_tail", "small_linear")
    print(f"enc_tail={v}: all kernels {sum(st.values()):.3f} ms/step  " + "  ".join(f"{k} {st.get(k, 0.0):.3f}" for k in keys), flush=True)
import time
for v in (0, 1, 0, 1):
    L.fpt_set_enc_tail(v)
    L.fpt_model_use_graphs(m.handle, 0); m.Register(scene.rgb, scene.depth, scene.mask, mesh.name); L.fpt_model_use_graphs(m.handle, 1)   # (drops the captured graph)
    for _ in range(4): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    t0 = time.perf_counter()
    for _ in range(20): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    print(f"enc_tail={v}: wall {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per Register", flush=True)
# slices of a sharded Register (wall clock per slice): small batches do not fill the chip, there the launch count matters
import torch
from foundationpose_cpp_amd.distributed import HipShardBackend
dev = torch.device("cuda", 0)
rgb, depth, mask = (torch.from_numpy(x).to(dev) for x in (scene.rgb, scene.depth, scene.mask))
H, Wd = scene.depth.shape
be = HipShardBackend(m, dev)
for count in (32, 63, 126):
    packed, _ = be.buffers(count, 1)
    def one():
        be.shard_begin_packed(rgb, depth, mask, H, Wd, mesh.name, 1, 0, count, packed, count)
        m.synchronize()
    for v in (0, 1, 0, 1):
        L.fpt_set_enc_tail(v)
        L.fpt_model_use_graphs(m.handle, 0); one(); L.fpt_model_use_graphs(m.handle, 1)
        for _ in range(4): one()
        t0 = time.perf_counter()
        for _ in range(30): one()
        print(f"slice of {count}, enc_tail={v}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms", flush=True)
