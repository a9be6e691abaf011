"""A/B of a library test hook on Track (N = 1, hipGraph replay), same process / same box.
    python tools/ab_track.py fpt_set_splitk_target 256 128 512"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
hook, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
hyp = syn.perturb_pose(scene.gt_pose)
_lib.use_test_lib()
L = _lib.lib()
for v in values:
    getattr(L, hook)(v)
    m = FoundationPose(mesh, scene.K, rp, sp)      # fresh model: the Track graph is captured under this setting
    rgb, depth = scene.rgb, scene.depth              # host frame: the upload of the crop window rides on every number
    for _ in range(10):
        m.Track(rgb, depth, hyp, mesh.name)
    t0 = time.perf_counter()
    for _ in range(300):
        m.Track(rgb, depth, hyp, mesh.name)
    print(f"{hook}({v}): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per Track")
    m.close()
