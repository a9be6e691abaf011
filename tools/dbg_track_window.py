"""debug: Track from host frames with random hypotheses (partial-row upload) -- prints progress so that a crash names its pose"""
import os, sys, tempfile
import torch  # before the HIP library: torch brings its own libamdhip64
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
rng = np.random.default_rng(12)
K = syn.intrinsics()
m1 = FoundationPose(mesh, K, rp, sp)
base = syn.perturb_pose(syn.pose_matrix(syn.random_rotation(3), [0, 0, 0.7]).astype(np.float32))
for k in range(60):
    rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    depth = rng.uniform(0.2, 2.0, (480, 640)).astype(np.float32)
    hyp = base.copy()
    tz = rng.uniform(0.3, 1.5)
    hyp[:3, 3] = [rng.uniform(-0.5, 0.5) * tz, rng.uniform(-1.2, 1.2) * tz, tz]
    print(k, hyp[:3, 3], flush=True)
    ok, p1 = m1.Track(rgb, depth, hyp, mesh.name)
    print("   ->", ok, p1[:3, 3] if ok else m1.last_error, flush=True)
if len(sys.argv) > 1:
    import ctypes as C, torch
    def _p(a): return a.ctypes.data_as(C.c_void_p)
    m2 = FoundationPose(mesh, K, rp, sp)
    out = np.zeros(16, np.float32)
    for k in range(60):
        rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        depth = rng.uniform(0.2, 2.0, (480, 640)).astype(np.float32)
        hyp = base.copy()
        tz = rng.uniform(0.3, 1.5)
        hyp[:3, 3] = [rng.uniform(-0.5, 0.5) * tz, rng.uniform(-1.2, 1.2) * tz, tz]
        print("two", k, hyp[:3, 3], flush=True)
        ok, p1 = m1.Track(rgb, depth, hyp, mesh.name)
        print("   m1 ->", ok, flush=True)
        r_d, d_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda()
        h16 = syn.to_colmajor(hyp[None])[0]
        rc = m2._L.fp_track_ex(m2.handle, C.c_void_p(r_d.data_ptr()), C.c_void_p(d_d.data_ptr()), 1, 480, 640, _p(h16), mesh.name.encode(), 1, _p(out))
        print("   m2 ->", rc, np.array_equal(p1, syn.from_colmajor(out[None])[0]), flush=True)
