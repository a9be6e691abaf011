#!/usr/bin/env python3
"""A/B of the attention kernel's softmax reference (lazy vs exact running maximum) on the 8-rank shard emulation of
tests/test_discriminative_gpu.py::test_sharded_register_agrees_on_the_winner: deviation of the gathered pooled features of shards of 32
from the unsharded Register's, relative to the between-hypothesis spread.   python tools/ab_shard_att.py [variants, default 1,10: 1 = shipped (exact running maximum), 10 = lazy reference]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
from foundationpose_cpp_amd.distributed import HipShardBackend, shard_range
cal = W.load_calibration(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "disc_calib_seed9.npz"))
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp, 9, cal); W.pack_synthetic("scorer", sp, 9, cal)
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
m = FoundationPose(mesh, syn.intrinsics(), rp, sp)
dev = torch.device("cuda", 0)
rgb, depth, mask = (torch.from_numpy(x).to(dev) for x in (scene.rgb, scene.depth, scene.mask))
be = HipShardBackend(m, dev)
L = _lib.lib()
res = {}
for v in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,10").split(",")]:
    L.fpt_set_att_variant(v)
    ok, pose, idx, scores, refined, feats = m.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok
    world = 8; per = -(-252 // world)
    packed, gathered = be.buffers(per, world)
    for r in range(world):
        b0, c = shard_range(252, world, r)
        be.shard_begin_packed(rgb, depth, mask, 480, 640, mesh.name, 1, b0, c, packed, per)
        be.before_collective(); gathered[r * per:(r + 1) * per].copy_(packed); be.after_collective()
    p16, idx_w = be.shard_finish_packed(gathered, 252)
    rows = gathered[:252].cpu().numpy()
    spread = (feats - feats.mean(0, keepdims=True)).std()
    dd = rows[:, :512] - feats
    worst = np.unravel_index(np.abs(dd).argmax(), dd.shape)
    print(f"variant {v}: max {np.abs(dd).max() / spread:.3f} rms {np.sqrt((dd ** 2).mean()) / spread:.4f} of the spread {spread:.4f}; worst at row {worst[0]} ch {worst[1]}; "
          f"rows with |d| > 0.2 spread: {np.unique(np.where(np.abs(dd) > 0.2 * spread)[0]).tolist()[:20]}; winner {idx_w} vs {idx}; |feat| rms {np.sqrt((feats ** 2).mean()):.3f}")
    res[v] = (feats.copy(), rows[:, :512].copy())
vs = list(res)
if len(vs) == 2:
    a, b = res[vs[0]], res[vs[1]]
    print(f"unsharded features, variant {vs[0]} vs {vs[1]}: max {np.abs(a[0] - b[0]).max():.5f}; sharded: max {np.abs(a[1] - b[1]).max():.5f}")
