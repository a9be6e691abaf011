"""Render (pose_setup + vertex + raster_shade + crop) determinism while convolutions run on other streams.
Device-resident outputs, compared on the device every iteration."""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_conv_stress.restype = C.c_longlong
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
m = FoundationPose(mesh, scene.K)
m.upload_frame(scene.rgb, scene.depth)
poses = m.get_hyp_poses(scene.mask)
N = len(poses)
p16s = [np.ascontiguousarray(syn.to_colmajor(poses), np.float32)]
alt = poses.copy(); alt[:, :3, :3] = poses[::-1, :3, :3]; alt[:, 2, 3] += 0.01      # a second, different pose set
p16s.append(np.ascontiguousarray(syn.to_colmajor(alt), np.float32))
out = [torch.zeros(N, 160, 160, 6, device="cuda") for _ in range(6)]
def render(a, b, k=0):
    p16 = p16s[k]
    rc = L.fp_render_and_transform(m.handle, mesh.name.encode(), p16.ctypes.data_as(C.c_void_p), N, C.c_float(1.2),
                                   C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 1)
    assert rc == 0, _lib.last_error()
render(out[0], out[1], 0)
render(out[4], out[5], 1)
mode = sys.argv[1] if len(sys.argv) > 1 else "conv"
stop = False
def load():
    while not stop:
        if mode == "conv":
            L.fpt_conv_stress(126, 40, 256, 256, 1, 12000, 1, 0)
        elif mode == "conv512":
            L.fpt_conv_stress(252, 20, 512, 512, 1, 12000, 1, 0)
th = [threading.Thread(target=load) for _ in range(0 if mode == "none" else 2)]
[t.start() for t in th]
import time
time.sleep(0 if mode == "none" else 6.0)   # let the load threads finish their host-side setup
bad = 0
for it in range(3000):
    k = it & 1                                     # alternate pose sets like Register's refine / score renders
    render(out[2], out[3], k)
    ra, rb = (out[0], out[1]) if k == 0 else (out[4], out[5])
    da, db = int((out[2] != ra).sum()), int((out[3] != rb).sum())
    if da or db:
        bad += 1
        hy = torch.nonzero((out[2] != ra).flatten(1).any(1)).flatten().tolist()
        print(f"iter {it}: A differs in {da} values (hypotheses {hy[:10]}), B in {db}")
stop = True
[t.join() for t in th]
print("mode", mode, "bad renders", bad, "of 3000")
