"""Objects tracked per second by ONE host thread with K models in flight (fp_track_submit / fp_track_wait) against K synchronous
Track calls in a row:  python tools/bench_multi_track.py [K ...]"""
import os, sys, tempfile, time
import torch   # before the library: both must share ONE HIP runtime (torch brings its own libamdhip64)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
hyp = syn.perturb_pose(scene.gt_pose)
for K in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    models = [FoundationPose(mesh, scene.K, rp, sp) for _ in range(K)]
    for _ in range(5):
        for m in models: m.Track(scene.rgb, scene.depth, hyp, mesh.name)
    t0 = time.perf_counter()
    for _ in range(100):
        for m in models: m.Track(scene.rgb, scene.depth, hyp, mesh.name)
    seq = K * 100 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(100):
        for m in models: m.track_submit(scene.rgb, scene.depth, hyp, mesh.name)
        for m in models: m.track_wait()
    pipe = K * 100 / (time.perf_counter() - t0)
    print(f"K = {K} objects: {seq:.0f} tracks/s one after the other, {pipe:.0f} tracks/s with all {K} in flight ({pipe / seq:.2f}x), host frames")
    # one frame resident in HBM shared by all K objects (a multi-object scene): the C ABI directly
    import ctypes as C, numpy as np
    rgb_d, depth_d = torch.from_numpy(scene.rgb).cuda(), torch.from_numpy(scene.depth).cuda()
    h16 = syn.to_colmajor(hyp); out = np.zeros(16, np.float32)
    H, Wd = scene.depth.shape
    def submit(m):
        m._must(m._L.fp_track_submit(m.handle, C.c_void_p(rgb_d.data_ptr()), C.c_void_p(depth_d.data_ptr()), 1, H, Wd,
                                     h16.ctypes.data_as(C.c_void_p), mesh.name.encode(), 1))
    def wait(m):
        m._must(m._L.fp_track_wait(m.handle, out.ctypes.data_as(C.c_void_p)))
    for _ in range(5):
        for m in models: submit(m)
        for m in models: wait(m)
    t0 = time.perf_counter()
    for _ in range(100):
        for m in models: submit(m)
        for m in models: wait(m)
    print(f"        frame resident in HBM, all {K} in flight: {K * 100 / (time.perf_counter() - t0):.0f} tracks/s")
    # fp_track_multi: the K objects as ONE batch of one model (geometry per object, one refine-net pass)
    m0 = models[0]
    hyps = np.tile(h16, (K, 1)); outs = np.zeros((K, 16), np.float32)
    names = (C.c_char_p * K)(*[mesh.name.encode()] * K)
    def multi():
        m0._must(m0._L.fp_track_multi(m0.handle, C.c_void_p(rgb_d.data_ptr()), C.c_void_p(depth_d.data_ptr()), 1, H, Wd, K,
                                      hyps.ctypes.data_as(C.c_void_p), C.cast(names, C.c_void_p), 1, outs.ctypes.data_as(C.c_void_p)))
    for _ in range(5): multi()
    t0 = time.perf_counter()
    for _ in range(200): multi()
    dt = (time.perf_counter() - t0) / 200
    print(f"        fp_track_multi, {K} objects in one batch: {dt * 1e3:.3f} ms per call = {K / dt:.0f} tracks/s")
    for m in models: m.close()
