"""Per-kernel table of one sharded Register slice (fp_register_shard_begin over `count` hypotheses), in-library profiler.
    python tools/profile_shard.py 32
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_cpp_amd import _lib as _l0
if len(sys.argv) > 3: _l0.use_test_lib()
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
from foundationpose_cpp_amd.distributed import HipShardBackend
count = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
rgb, depth, mask = (torch.from_numpy(a).to(dev) for a in (scene.rgb, scene.depth, scene.mask))
H, Wd = scene.depth.shape
be = HipShardBackend(m, dev)
if len(sys.argv) > 3:   # optional test hook: python tools/profile_shard.py 32 fpt_set_rem_small 1
    from foundationpose_cpp_amd import _lib
    for hk, hv in zip(sys.argv[2::2], sys.argv[3::2]):   # any number of hook / value pairs
        getattr(_lib.lib(), hk)(int(hv))  # needs _lib.use_test_lib() (done at import above)
packed, _ = be.buffers(count, 1)
def one():
    be.shard_begin_packed(rgb, depth, mask, H, Wd, mesh.name, 1, 0, count, packed, count)
    m.synchronize()
for _ in range(2): one()
m.profile(True); m.profile_reset()
for _ in range(5): one()
r = m.profile_report()
tot = sum(v["ms"] for v in r.values()) / 5
print(f"slice of {count}: {tot:.3f} ms of kernels per Register")
for k, v in sorted(r.items(), key=lambda kv: -kv[1]["ms"])[:24]:
    print(f"{k:48s} calls/step {v['calls']/5:5.1f}  ms/step {v['ms']/5:7.3f}  us/call {v['ms']/v['calls']*1e3:7.1f}  TF/s {v['flops']/max(v['ms'],1e-9)/1e9:7.1f}")
