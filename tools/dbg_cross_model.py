"""Does a Register / Track of one model modify any device buffer of ANOTHER (idle) model?  Sequential, deterministic."""
import sys, os, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
mesh = syn.make_mesh()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
scenes = [syn.make_scene(mesh), syn.make_scene(mesh, t=(-0.03, 0.02, 0.62), rot_seed=9)]
models = [FoundationPose(mesh, syn.intrinsics(), rp, sp) for _ in scenes]
NAMES = ["recs", "poses", "clip", "attr", "nn_in", "trans", "rot", "scores", "feat", "arena", "arena_f32"]
def dig(m):
    o = np.zeros(16, np.uint64); assert L.fpt_digest_buffers(m.handle, o.ctypes.data_as(C.c_void_p)) == 0, _lib.last_error(); return o
for m, s in zip(models, scenes):
    m.Register(s.rgb, s.depth, s.mask, mesh.name)
for victim, actor in ((0, 1), (1, 0)):
    before = dig(models[victim])
    s = scenes[actor]
    for k in range(3):
        models[actor].Register(s.rgb, s.depth, s.mask, mesh.name)
        models[actor].Track(s.rgb, s.depth, syn.perturb_pose(s.gt_pose), mesh.name)
    after = dig(models[victim])
    ch = [NAMES[i] for i in range(11) if before[i] != after[i]]
    print(f"model {actor} ran; buffers of idle model {victim} that changed: {ch}")
