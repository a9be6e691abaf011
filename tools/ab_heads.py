"""Same-box A/B of the transformer-head kernels of round 5's last session: the launch chain (gemm_k32 x 3 + layernorm + layernorm_mean per
head, gemm_k32 QKV) against enc_tail_kernel + qkv_tile_kernel, on the graph-replayed Register (N = 252) and Track.  Boxes of the pool
differ by several per cent (the 3x3 layers run at the chip's power limit), so only a same-box comparison says what the code changed.
    python tools/ab_heads.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
hyp = syn.perturb_pose(scene.gt_pose)
for v in (0, 1, 0, 1):
    L.fpt_set_enc_tail(v); L.fpt_set_qkv_tile(v)
    m = FoundationPose(mesh, scene.K, rp, sp)     # fresh model: its graphs are captured under this setting
    for _ in range(5): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    t0 = time.perf_counter()
    for _ in range(30): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    reg = (time.perf_counter() - t0) / 30 * 1e3
    for _ in range(10): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
    t0 = time.perf_counter()
    for _ in range(300): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
    trk = (time.perf_counter() - t0) / 300 * 1e6
    print(f"{'tile kernels (enc_tail + qkv_tile)' if v else 'launch chain (round-5 start)      '}: Register {reg:.3f} ms, Track {trk:.1f} us (host frames)", flush=True)
    m.close()
