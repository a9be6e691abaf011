"""What the float model is worth: triangle-id flips and tensor deltas between the contracted
(nvcc -fmad=true like) and the separately-rounded model of the rendering stage, CPU oracle, N = 252, both crop ratios."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import synthetic as syn
from oracle import fp_oracle as fo
mesh = syn.make_mesh(); scene = syn.make_scene(mesh); om = fo.OracleMesh(mesh)
poses = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)
for ratio in (1.2, 1.1):
    out = {}
    for fm in (False, True):
        fo.set_fmad(fm)
        out[fm] = fo.render(om, poses, scene.K, scene.depth.shape, ratio, debug=True)
    (a0, t0, _), (a1, t1, _) = out[False], out[True]
    flips = t0 != t1
    d = np.abs(a0 - a1)
    print(f"crop ratio {ratio}: triangle-id flips {int(flips.sum())} of {t0.size} pixels ({int((t0 > 0).sum())} foreground) in "
          f"{int(flips.reshape(252, -1).any(1).sum())} of 252 hypotheses; tensor: max |delta| {d.max():.3e} (rgb {d[..., :3].max():.3e}, xyz {d[..., 3:].max():.3e}), "
          f"pixels with any delta {int((d.max(-1) > 0).sum())}, mean |delta| {d.mean():.3e}")
fo.set_fmad(True)
