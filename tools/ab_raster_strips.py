import sys, json, subprocess, os
sys.path.insert(0, '/root/repo')
import numpy as np, time, ctypes as C
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, _lib
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
m = FoundationPose(mesh, scene.K)
m.upload_frame(scene.rgb, scene.depth)
poses = m.get_hyp_poses(scene.mask)
_lib.use_test_lib()
L = _lib.lib()
for rows in (40, 20, 8, 40, 20, 8):
    L.fpt_set_raster_strip_rows(rows)
    m.profile(True); m.profile_reset()
    for _ in range(5):
        m.render_and_transform(mesh.name, poses, 1.2)
    r = m.profile_report()
    print(rows, {k: round(v['ms']/v['calls'],4) for k,v in r.items() if k in ('raster_shade','vertex','crop_warp')})
