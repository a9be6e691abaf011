import os, sys, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
hyp = syn.perturb_pose(scene.gt_pose)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hyps = np.stack([hyp] * K); names = [mesh.name] * K
for _ in range(3): m.track_multi(scene.rgb, scene.depth, hyps, names)
m.profile(True); m.profile_reset()
for _ in range(10): m.track_multi(scene.rgb, scene.depth, hyps, names)
r = m.profile_report()
tot = 0
for k, v in sorted(r.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:55s} calls {v['calls']/10:5.1f}  us {v['ms']*1000/10:7.1f}"); tot += v["ms"] * 100
print("sum", round(tot, 1), "us")
