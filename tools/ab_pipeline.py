"""A/B of a library test hook inside the real Register pipeline (N = 252), same process / same box.

    python tools/ab_pipeline.py fpt_set_att_variant 7 1 7 1
    python tools/ab_pipeline.py fpt_set_conv_ablate 0 16 0 16        # 16 = no streaming stores
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
hook, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp()
rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
_lib.use_test_lib()
L = _lib.lib()
m = FoundationPose(mesh, scene.K, rp, sp)
if os.environ.get("FP_AB_INPLANE"):   # hypotheses = 42 x in-plane steps (default 6 -> 252)
    m.set_inplane_steps(int(os.environ["FP_AB_INPLANE"]))
for v in values:
    getattr(L, hook)(v)
    for _ in range(2):
        m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    m.profile(True); m.profile_reset()
    for _ in range(6):
        m.Register(scene.rgb, scene.depth, scene.mask, mesh.name)
    r = m.profile_report()
    m.profile(False)
    st = {}
    for k, x in r.items():
        st[k.split("/")[0]] = st.get(k.split("/")[0], 0.0) + x["ms"] / 6
    top = sorted(st.items(), key=lambda kv: -kv[1])[:8]
    watch = os.environ.get("FP_AB_WATCH")
    extra = f"  [{watch} {st.get(watch, 0.0):.3f}]" if watch else ""
    print(f"{hook}({v}): all kernels {sum(st.values()):.3f} ms/step{extra}  " + "  ".join(f"{k} {t:.3f}" for k, t in top))
