import os, sys, tempfile
sys.path.insert(0, '/root/repo')
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
hyp = syn.perturb_pose(scene.gt_pose)
for _ in range(3): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
m.profile(True); m.profile_reset()
for _ in range(20): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
r = m.profile_report()
tot = 0
for k, v in sorted(r.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:55s} calls/track {v['calls']/20:5.1f}  us/track {v['ms']*1000/20:7.1f}")
    tot += v["ms"] * 1000 / 20
print("sum of kernel time per Track:", round(tot, 1), "us")
import time
m.profile(False)
for _ in range(5): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
t0 = time.perf_counter()
for _ in range(200): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
print("wall per Track (host frame, graph):", (time.perf_counter() - t0) / 200 * 1e6, "us")
