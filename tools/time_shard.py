"""Per-rank latency of a sharded Register on ONE GPU: the packed shard protocol (fp_register_shard_begin_packed over `count` of
252 or 1008 hypotheses, event-ordered hand-off to torch's stream and back, fp_register_shard_finish_packed over all rows) --
what each rank of a strong-scaled run executes besides the all-gather itself -- and the host time the calls themselves take.

    python tools/time_shard.py            # counts 32 (252/8), 63 (252/4), 126 (252/2 or 1008/8), 252
"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
from foundationpose_cpp_amd.distributed import HipShardBackend
if len(sys.argv) >= 3:   # A/B: python tools/time_shard.py HOOK VALUE  (test build of the library)
    _lib.use_test_lib()
    getattr(_lib.lib(), sys.argv[1])(int(sys.argv[2]))
    print(f"{sys.argv[1]}({sys.argv[2]})")

dev = torch.device("cuda", 0)
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
rgb, depth, mask = (torch.from_numpy(a).to(dev) for a in (scene.rgb, scene.depth, scene.mask))
H, Wd = scene.depth.shape
be = HipShardBackend(m, dev)
for n_total, counts in ((252, (32, 63, 126, 252)), (1008, (126,)), (2016, (252,))):   # 2016 / 252 = what `bench.py --gpus 8` runs per rank (weak scaling)
    m.set_inplane_steps(n_total // 42)
    for count in counts:
        world = -(-n_total // count)
        packed, gathered = be.buffers(count, world)
        gathered.zero_()
        gathered[:, 512] = gathered[:, 517] = gathered[:, 522] = gathered[:, 527] = 1.0    # identity poses in the other ranks' rows
        host = 0.0
        for it in range(12):
            if it == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
            h0 = time.perf_counter()
            be.shard_begin_packed(rgb, depth, mask, H, Wd, mesh.name, 1, 0, count, packed, count)
            be.before_collective()
            gathered[:count].copy_(packed)          # stands in for the all-gather on torch's stream
            be.after_collective()
            host += time.perf_counter() - h0        # everything before the finish's single synchronisation is asynchronous
            be.shard_finish_packed(gathered, n_total)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"N={n_total} shard of {count}: {ms:.3f} ms per Register per rank ({host / 10 * 1e3:.3f} ms of it host time in the asynchronous calls = "
              f"{host / 10 * 1e3 / ms * 100:.1f} %) -> {n_total / ms * 1e3:.0f} hyp/s aggregate at {world} ranks (excluding the all-gather)")
