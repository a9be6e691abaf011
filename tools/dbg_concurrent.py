"""Debug helper: two models on two threads run Register concurrently; per call the score vector / winner / refined poses
are compared with the same model's sequential result."""
import sys, os, threading, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
if len(sys.argv) > 1:
    L.fpt_set_conv_variant(int(sys.argv[1]))
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mesh = syn.make_mesh()
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
scenes = [syn.make_scene(mesh), syn.make_scene(mesh, t=(-0.03, 0.02, 0.62), rot_seed=9)]
models = [FoundationPose(mesh, syn.intrinsics(), rp, sp) for _ in scenes]
def p(a): return a.ctypes.data_as(C.c_void_p)
def reg(m, s):
    feat, poses = C.c_void_p(), C.c_void_p()
    rc = L.fp_register_shard_begin(m.handle, p(s.rgb), p(s.depth), p(s.mask), 0, 480, 640, mesh.name.encode(), 1, 0, 252,
                                   C.byref(feat), C.byref(poses))
    assert rc == 0, _lib.last_error()
    out = np.zeros(16, np.float32); best = C.c_int(); sc = np.zeros(252, np.float32)
    rc = L.fp_register_shard_finish(m.handle, feat, poses, 252, p(out), C.byref(best), p(sc))
    assert rc == 0, _lib.last_error()
    dg = np.zeros(16, np.uint64)
    L.fpt_digests(m.handle, p(dg))
    return best.value, sc.copy(), out.copy(), dg
NAMES = ['recs', 'clip', 'attr', 'nn_in A', 'nn_in B', 'trans', 'rot', 'poses', 'clip2', 'attr2', 'nn_in2 A', 'nn_in2 B', 'feat', 'scores']
for m, s_ in zip(models, scenes):
    m.Register(s_.rgb, s_.depth, s_.mask, mesh.name)
    L.fpt_digests(m.handle, p(np.zeros(16, np.uint64)))   # first call switches the digests on
seq = [reg(m, s) for m, s in zip(models, scenes)]
seq2 = [reg(m, s) for m, s in zip(models, scenes)]
for i in range(2):
    assert seq[i][0] == seq2[i][0] and np.array_equal(seq[i][1], seq2[i][1]) and np.array_equal(seq[i][3], seq2[i][3]), "sequential runs differ"
    top = np.sort(seq[i][1])[::-1]
    print(f"model {i}: winner {seq[i][0]} top-2 score gap {top[0]-top[1]:.3e}")
BN = ['recs', 'poses', 'clip', 'attr', 'nn_in', 'trans', 'rot', 'scores', 'feat', 'arena', 'arena_f32', 'verts', 'normals', 'uvs', 'faces', 'tex']
def bufdig(m):
    o = np.zeros(16, np.uint64); assert L.fpt_digest_buffers(m.handle, p(o)) == 0; return o
static0 = [bufdig(m) for m in models]
res = [[], []]
def worker(i):
    for k in range(ITERS):
        res[i].append(reg(models[i], scenes[i]))
th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
bad = 0
for i in range(2):
    for k, (b, sc, out, dg) in enumerate(res[i]):
        if not np.array_equal(dg, seq[i][3]):
            first = [NAMES[j] for j in range(14) if dg[j] != seq[i][3][j]]
            print(f"model {i} iter {k}: stage digests differ: {first}")
        if not np.array_equal(sc, seq[i][1]):
            bad += 1
            dif = np.abs(sc - seq[i][1])
            print(f"model {i} iter {k}: winner {b} (seq {seq[i][0]}), scores differ in {int((dif > 0).sum())} of 252, max |d| {dif.max():.3e} at {int(dif.argmax())}")
print("bad", bad, "of", 2 * ITERS)
for i, m in enumerate(models):
    after = bufdig(m)
    ch = [BN[j] for j in range(11, 16) if after[j] != static0[i][j]]
    r2 = reg(m, scenes[i])
    print(f"model {i}: static mesh buffers changed: {ch}; sequential Register afterwards equals baseline: {np.array_equal(r2[1], seq[i][1])}")
