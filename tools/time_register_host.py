"""Wall time of Register from HOST frames (the reference's calling convention) vs the kernel time."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
for _ in range(3): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
t0 = time.perf_counter()
for _ in range(20): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
print("Register from host frames: %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
hyp = syn.perturb_pose(scene.gt_pose)
for _ in range(5): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
t0 = time.perf_counter()
for _ in range(200): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
print("Track from host frames: %.3f ms per call" % ((time.perf_counter() - t0) / 200 * 1e3))
