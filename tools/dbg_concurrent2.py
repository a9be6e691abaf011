"""Race hunt: which ELEMENTS of the vertex-stage buffers differ when two models run concurrently (lock disabled)."""
import sys, os, threading, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_read_buffer.restype = C.c_longlong
L.fpt_read_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mesh = syn.make_mesh()
V = mesh.vertices.shape[0]
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
scenes = [syn.make_scene(mesh), syn.make_scene(mesh, t=(-0.03, 0.02, 0.62), rot_seed=9)]
models = [FoundationPose(mesh, syn.intrinsics(), rp, sp) for _ in scenes]
def p(a): return a.ctypes.data_as(C.c_void_p)
def dump(m, which, nbytes):
    a = np.zeros(nbytes // 4, np.uint32)
    n = L.fpt_read_buffer(m.handle, which, p(a), nbytes)
    assert n == nbytes, n
    return a
SZ = {0: 252 * 212, 1: 252 * V * 16, 2: 252 * V * 16, 4: 252 * 64}
def reg(m, s):
    ok, pose = m.Register(s.rgb, s.depth, s.mask, mesh.name)
    assert ok
    return {k: dump(m, k, SZ[k]) for k in SZ}
ref = [reg(m, s) for m, s in zip(models, scenes)]
ref2 = [reg(m, s) for m, s in zip(models, scenes)]
for i in range(2):
    for k in SZ: assert np.array_equal(ref[i][k], ref2[i][k]), ("sequential differs", i, k)
res = [[], []]
def worker(i):
    for k in range(ITERS):
        res[i].append(reg(models[i], scenes[i]))
th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
NAME = {0: "recs", 1: "clip", 2: "attr", 4: "poses"}
for i in range(2):
    for it, r in enumerate(res[i]):
        for k in SZ:
            bad = np.nonzero(r[k] != ref[i][k])[0]
            if bad.size == 0: continue
            by = bad * 4
            runs = np.split(by, np.nonzero(np.diff(by) != 4)[0] + 1)
            desc = [(int(x[0]), int(x[-1] - x[0] + 4)) for x in runs[:6]]
            print(f"model {i} iter {it} {NAME[k]}: {bad.size} dwords differ in {len(runs)} runs; first runs (byte offset, length): {desc}")
            if k in (1, 2):
                e = bad[0] // 4  # float4 index
                print(f"   first bad float4 index {e} = hypothesis {e // V} vertex {e % V}; got {r[k].view(np.float32)[e*4:e*4+4]} want {ref[i][k].view(np.float32)[e*4:e*4+4]}")
                # is the wrong hypothesis block equal to another hypothesis' / alignment of runs to 64 / 128 / 4096 bytes
                print("   run starts mod 128:", sorted(set(int(x[0]) % 128 for x in runs))[:8], " lengths:", sorted(set(int(x[-1] - x[0] + 4) for x in runs))[:8])
print("done")
