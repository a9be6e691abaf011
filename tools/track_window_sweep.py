"""Track from host frames vs frames resident in device memory as a function of the observed-crop window (object distance):
what the packed-window upload costs per call.   python tools/track_window_sweep.py"""
import ctypes as C, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh(); K = syn.intrinsics()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, K, rp, sp)
rng = np.random.default_rng(1)
rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8); depth = rng.uniform(0.2, 2.0, (480, 640)).astype(np.float32)
r_d, d_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda()
base = syn.perturb_pose(syn.pose_matrix(syn.random_rotation(3), [0, 0, 0.7]).astype(np.float32))
out = np.zeros(16, np.float32)
def p(a): return a.ctypes.data_as(C.c_void_p)
for tz in (0.35, 0.5, 0.7, 1.0, 1.5, 3.0):
    hyp = base.copy(); hyp[:3, 3] = [0, 0, tz]
    h16 = syn.to_colmajor(hyp[None])[0]
    win = 2 * K[1, 1] * mesh.diameter * 0.6 / tz + 9
    res = []
    for host in (True, False):
        def one():
            if host: m.Track(rgb, depth, hyp, mesh.name)
            else: m._must(m._L.fp_track_ex(m.handle, C.c_void_p(r_d.data_ptr()), C.c_void_p(d_d.data_ptr()), 1, 480, 640, p(h16), mesh.name.encode(), 1, p(out)))
        for _ in range(20): one()
        t0 = time.perf_counter()
        for _ in range(400): one()
        res.append((time.perf_counter() - t0) / 400 * 1e6)
    print(f"tz {tz:4.2f} m: window ~{win:4.0f} px square ({win * win * 7 / 1e3:6.0f} KB)  host frame {res[0]:6.1f} us  device frame {res[1]:6.1f} us  difference {res[0] - res[1]:5.1f} us")
m.close()
