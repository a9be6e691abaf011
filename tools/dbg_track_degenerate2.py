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
hyp = base.copy(); hyp[:3, 3] = [0, 0, float(sys.argv[2]) if len(sys.argv) > 2 else 1e-12]
m1.upload_frame(scene.rgb, scene.depth)
step = sys.argv[1]
print(step, flush=True)
if step == "raster":
    tri, rast = m1.debug_rasterize(mesh.name, hyp[None], 1.2); print(tri.max(), np.isfinite(rast).all(), flush=True)
elif step == "crop":
    p = np.zeros((1, 160, 160, 6), np.float32)
    import ctypes as C
    m1._must(m1._L.fp_render_and_transform(m1._h, mesh.name.encode(), syn.to_colmajor(hyp[None]).ctypes.data_as(C.c_void_p), 1, C.c_float(1.2), None, p.ctypes.data_as(C.c_void_p), 0)); print(np.isfinite(p).all(), flush=True)
elif step == "render":
    p = np.zeros((1, 160, 160, 6), np.float32)
    import ctypes as C
    m1._must(m1._L.fp_render_and_transform(m1._h, mesh.name.encode(), syn.to_colmajor(hyp[None]).ctypes.data_as(C.c_void_p), 1, C.c_float(1.2), p.ctypes.data_as(C.c_void_p), None, 0)); print(np.isfinite(p).all(), flush=True)
print("done", flush=True)
