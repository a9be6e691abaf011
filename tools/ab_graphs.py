"""Wall-clock A/B of hipGraph replay (Register N = 252 and Track) on one model, same box."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
hyp = syn.perturb_pose(scene.gt_pose)
for on in (0, 1, 0, 1, 0, 1):
    L.fpt_model_use_graphs(m.handle, on)
    for _ in range(4): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    t0 = time.perf_counter()
    for _ in range(20): m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    reg = (time.perf_counter() - t0) / 20 * 1e3
    for _ in range(5): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
    t0 = time.perf_counter()
    for _ in range(200): m.Track(scene.rgb, scene.depth, hyp, mesh.name)
    trk = (time.perf_counter() - t0) / 200 * 1e3
    print(f"graphs {on}: Register {reg:.3f} ms   Track {trk:.3f} ms   (host frames)   state {L.fpt_model_graph_state(m.handle)}")
