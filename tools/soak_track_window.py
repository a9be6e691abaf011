"""Soak of Track's packed-window upload: N host-frame Tracks on one model, eight different noise frames and a
random hypothesis per call, each compared bit for bit with a second model that reads the same frame whole from device memory.
    python tools/soak_track_window.py [calls]"""
import ctypes as C, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
mesh = syn.make_mesh(); K = syn.intrinsics()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m1 = FoundationPose(mesh, K, rp, sp); m2 = FoundationPose(mesh, K, rp, sp)
rng = np.random.default_rng(1)
base = syn.perturb_pose(syn.pose_matrix(syn.random_rotation(3), [0, 0, 0.7]).astype(np.float32))
out = np.zeros(16, np.float32); bad = 0
def p(a): return a.ctypes.data_as(C.c_void_p)
frames = [(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), rng.uniform(0.2, 2.0, (480, 640)).astype(np.float32)) for _ in range(8)]
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(dp).cuda()) for r, dp in frames]
t0 = time.time()
for k in range(calls):
    rgb, depth = frames[k % 8]
    hyp = base.copy(); tz = rng.uniform(0.35, 1.5)
    hyp[:3, 3] = [rng.uniform(-0.45, 0.45) * tz, rng.uniform(-0.4, 0.4) * tz, tz]
    ok, p1 = m1.Track(rgb, depth, hyp, mesh.name)
    assert ok, m1.last_error
    r_d, d_d = dev[k % 8]
    m2._must(m2._L.fp_track_ex(m2.handle, C.c_void_p(r_d.data_ptr()), C.c_void_p(d_d.data_ptr()), 1, 480, 640, p(syn.to_colmajor(hyp[None])[0]), mesh.name.encode(), 1, p(out)))
    if not np.array_equal(p1, syn.from_colmajor(out[None])[0]): bad += 1
print(f"soak: {calls} host-frame Tracks (packed window) vs whole device frames: {bad} mismatches in {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
