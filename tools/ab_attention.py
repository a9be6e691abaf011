"""A/B of attention kernel variants inside the real Register pipeline (QKV freshly written by the GEMM)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
L = _lib.lib()
for v in (7, 1, 5, 3, 7, 1, 5, 3):
    L.fpt_set_att_variant(v)
    for _ in range(2):
        m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    m.profile(True); m.profile_reset()
    for _ in range(5):
        m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    r = m.profile_report()
    m.profile(False)
    att = sum(x["ms"] for k, x in r.items() if k.startswith("attention")) / 5
    tot = sum(x["ms"] for x in r.values()) / 5
    print(f"variant {v}: attention {att:.3f} ms/step   all kernels {tot:.3f} ms/step")
