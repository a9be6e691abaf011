"""debug: Track with degenerate hypotheses, one per process invocation (argv[1] = index)"""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
scene = syn.make_scene(mesh)
m1 = FoundationPose(mesh, scene.K, rp, sp)
base = syn.perturb_pose(scene.gt_pose)
T = [[0, 0, -0.5], [0, 1e30, 1e-5], [0, -1e30, 0.5], [0, 0, 1e-12], [0, 0, 0.05], [1e6, 0, 0.5], [0, 0, 1e4], [float("nan"), 0, 0.5]]
t = T[int(sys.argv[1])]
hyp = base.copy(); hyp[:3, 3] = t
print("t =", t, flush=True)
ok, p = m1.Track(scene.rgb, scene.depth, hyp, mesh.name)
print("  ->", ok, p[:3, 3] if ok else m1.last_error, flush=True)
