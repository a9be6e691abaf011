"""Per-rank latency of a sharded Register on ONE GPU: fp_register_shard_begin over `count` of 252 (or 1008) hypotheses +
fp_register_shard_finish over all of them -- what each rank of a strong-scaled run executes besides the all-gather.

    python tools/time_shard.py            # counts 32 (252/8), 63 (252/4), 126 (252/2 or 1008/8), 252
"""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
from foundationpose_cpp_amd.distributed import HipShardBackend

dev = torch.device("cuda", 0)
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
rgb, depth, mask = (torch.from_numpy(a).to(dev) for a in (scene.rgb, scene.depth, scene.mask))
H, Wd = scene.depth.shape
be = HipShardBackend(m, dev)
for n_total, counts in ((252, (32, 63, 126, 252)), (1008, (126,))):
    m.set_inplane_steps(n_total // 42)
    feat_all = torch.zeros((n_total, 512), device=dev); pose_all = torch.zeros((n_total, 16), device=dev)
    pose_all[:, 0] = pose_all[:, 5] = pose_all[:, 10] = pose_all[:, 15] = 1.0
    for count in counts:
        for it in range(7):
            if it == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            f, p = be.shard_begin(rgb, depth, mask, H, Wd, mesh.name, 1, 0, count)
            feat_all[:count] = f; pose_all[:count] = p
            be.shard_finish(feat_all, pose_all)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"N={n_total} shard of {count}: {ms:.3f} ms per Register per rank -> {n_total / ms * 1e3:.0f} hyp/s aggregate at {n_total // count if n_total % count == 0 else round(n_total / count)} ranks (excluding the all-gather)")
