"""A/B of a test hook on fp_track_multi(K):  python tools/ab_track_multi.py K HOOK v..."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import _lib
_lib.use_test_lib(); L = _lib.lib()
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
K, hook, values = int(sys.argv[1]), sys.argv[2], [int(v) for v in sys.argv[3:]]
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
hyp = syn.perturb_pose(scene.gt_pose); hyps = np.stack([hyp] * K); names = [mesh.name] * K
for v in values:
    getattr(L, hook)(v)
    m = FoundationPose(mesh, scene.K, rp, sp)
    for _ in range(5): m.track_multi(scene.rgb, scene.depth, hyps, names)
    t0 = time.perf_counter()
    for _ in range(200): m.track_multi(scene.rgb, scene.depth, hyps, names)
    print(f"K={K} {hook}({v}): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (host frame)")
    m.close()
