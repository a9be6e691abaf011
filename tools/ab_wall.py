"""Wall-clock A/B of a library test hook on the graph-replayed Register (N = 252) and Track, same process / same box: the graphs are
re-captured after every switch.   python tools/ab_wall.py fpt_set_rem_side 0 1 0 1"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
hook, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
m = FoundationPose(mesh, scene.K, rp, sp)
if os.environ.get("FP_AB_INPLANE"):
    m.set_inplane_steps(int(os.environ["FP_AB_INPLANE"]))
first = None
for v in values:
    getattr(L, hook)(v)
    L.fpt_model_use_graphs(m.handle, 0)
    m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    L.fpt_model_use_graphs(m.handle, 1)
    for _ in range(4):
        ok, pose = m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    t0 = time.perf_counter()
    for _ in range(30):
        ok, pose = m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    reg = (time.perf_counter() - t0) / 30 * 1e3
    if first is None: first = np.array(pose)
    print(f"{hook}({v}): Register {reg:.3f} ms (host frames, graphs {L.fpt_model_graph_state(m.handle)})  pose == first run: {bool((np.array(pose) == first).all())}")
