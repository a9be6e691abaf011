"""A/B of the rasteriser's strip height (fpt_set_raster_strip_rows: 8 / 20 / 40 rows with 256 threads; 1080 / 1040 / 1020 = 80 / 40 /
20 rows with 1024 threads) at a given number of hypotheses:  python tools/ab_raster_strips.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
L = _lib.lib()
from foundationpose_cpp_amd import FoundationPose, synthetic as syn
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
m = FoundationPose(mesh, scene.K)
m.upload_frame(scene.rgb, scene.depth)
poses_all = m.get_hyp_poses(scene.mask)
for n in [int(a) for a in sys.argv[1:]] or [252]:
    poses = poses_all[:n]
    for rows in (8, 20, 40, 1020, 1040, 1080, 0):
        L.fpt_set_raster_strip_rows(rows)
        for _ in range(2): m.render_and_transform(mesh.name, poses, 1.2)
        m.profile(True); m.profile_reset()
        for _ in range(8): m.render_and_transform(mesh.name, poses, 1.2)
        r = m.profile_report(); m.profile(False)
        print(f"N={n} strip code {rows}: raster_shade {r['raster_shade']['ms'] / r['raster_shade']['calls'] * 1e3:.1f} us")
