import os, sys, tempfile
sys.path.insert(0, '/root/repo')
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
for _ in range(2): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
m.profile(True); m.profile_reset()
for _ in range(5): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
r = m.profile_report()
for k, v in sorted(r.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:50s} calls/step {v['calls']/5:5.1f}  ms/step {v['ms']/5:7.3f}  us/call {v['ms']/v['calls']*1e3:7.1f}  TF/s {v['flops']/max(v['ms'],1e-9)/1e9:7.1f}")
