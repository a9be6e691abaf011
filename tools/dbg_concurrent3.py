"""Race hunt, aggressor bisect: thread A repeats the vertex+raster stage and checks the vertex buffers every iteration;
thread B runs one kind of work on ANOTHER model.  usage: dbg_concurrent3.py MODE [iters]"""
import sys, os, threading, tempfile, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W, _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_read_buffer.restype = C.c_longlong
L.fpt_read_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
MODE = sys.argv[1] if len(sys.argv) > 1 else "register"
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
NA = 64
mesh = syn.make_mesh()
V = mesh.vertices.shape[0]
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
scenes = [syn.make_scene(mesh), syn.make_scene(mesh, t=(-0.03, 0.02, 0.62), rot_seed=9)]
models = [FoundationPose(mesh, syn.intrinsics(), rp, sp) for _ in scenes]
def p(a): return a.ctypes.data_as(C.c_void_p)
def dump(m, which, nbytes):
    a = np.zeros(nbytes // 4, np.uint32)
    assert L.fpt_read_buffer(m.handle, which, p(a), nbytes) == nbytes
    return a
A, B = models
for m, s in zip(models, scenes):
    ok, _ = m.Register(s.rgb, s.depth, s.mask, mesh.name); assert ok
A.upload_frame(scenes[0].rgb, scenes[0].depth)
posesA = A.get_hyp_poses(scenes[0].mask)[:NA]
posesB = B.get_hyp_poses(scenes[1].mask)
L.fpt_attention_bench.restype = C.c_float
L.fpt_attention_bench.argtypes = [C.c_int] * 4
L.fpt_conv_stress.restype = C.c_longlong
L.fpt_conv_stress.argtypes = [C.c_int] * 8
L.fpt_vertex_dbg.restype = C.c_longlong
L.fpt_vertex_dbg.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
DBG = NA * V * 48
assert L.fpt_vertex_dbg(None, None, DBG) == 0
AGG_N = 63  # the aggressor must not use N == 64 (that launch size fills the debug buffer)
def stepA():
    A.debug_rasterize(mesh.name, posesA, 1.2)
    dbg = np.zeros(DBG // 4, np.float32)
    c, a = dump(A, 1, NA * V * 16), dump(A, 2, NA * V * 16)
    assert L.fpt_vertex_dbg(A.handle, p(dbg), DBG) == DBG
    return c, a, dbg.reshape(-1, 12)
ref = stepA(); ref2 = stepA()
assert np.array_equal(ref[0], ref2[0]) and np.array_equal(ref[1], ref2[1])
stop = False
def aggressor():
    s = scenes[1]
    if MODE == "refiner":
        a, b = B.render_and_transform(mesh.name, posesB[:AGG_N], 1.2)
    while not stop:
        if MODE == "register": B.Register(s.rgb, s.depth, s.mask, mesh.name)
        elif MODE == "track": B.Track(s.rgb, s.depth, posesB[0], mesh.name)
        elif MODE == "render": B.render_and_transform(mesh.name, posesB[:AGG_N], 1.2)
        elif MODE == "raster": B.debug_rasterize(mesh.name, posesB[:AGG_N], 1.2)
        elif MODE == "refiner": B.refiner_infer(a, b)
        elif MODE == "copy": B.upload_frame(s.rgb, s.depth)
        elif MODE == "attn": L.fpt_attention_bench(64, 400, 20, 1)
        elif MODE == "halo": L.fpt_conv_stress(64, 40, 128, 128, 0, 20, 1, 0)
        elif MODE == "bigpp": L.fpt_conv_stress(256, 20, 512, 512, 0, 5, 1, 0)
        elif MODE == "igemm": L.fpt_conv_stress(4, 40, 128, 128, 0, 50, 1, 0)
        elif MODE == "none": time.sleep(0.01)
th = threading.Thread(target=aggressor); th.start()
bad = 0
for it in range(ITERS):
    c, a, dbg = stepA()
    dc, da = np.nonzero(c != ref[0])[0], np.nonzero(a != ref[1])[0]
    if dc.size or da.size:
        bad += 1
        wonly = bool(da.size) and bool(np.all(da % 4 == 3))
        if da.size:
            idx = np.unique(da // 4)[:6]
            for e in idx:
                print(f"   float4 {e} v {e % V} lane {(e % V) % 64}: a.w got {a.view(np.float32)[e*4+3]:.6f} want {ref[1].view(np.float32)[e*4+3]:.6f}; dbg n,l2 {dbg[e, :4]} (ref {ref[2][e, :4]}) u,val {dbg[e, 4:8]} (ref {ref[2][e, 4:8]}) dt {dbg[e, 8]:.0f} ticks hwid {dbg[e, 9:10].view(np.uint32)[0]:#x}")
        dts = dbg[:, 8]
        badv = np.unique(np.concatenate([dc, da]) // 4)
        print(f"   wave duration (10 ns ticks): all median {np.median(dts):.0f} p99 {np.percentile(dts, 99):.0f} max {dts.max():.0f}; bad entries median {np.median(dts[badv]):.0f} min {dts[badv].min():.0f}; entries with dt > 1000: {(dts > 1000).sum()} of which bad {np.isin(np.nonzero(dts > 1000)[0], badv).sum()} ({badv.size} bad total)")
        lanes = np.bincount((badv % V) % 64, minlength=64)
        hw = dbg[badv, 9].view(np.uint32)
        tb = dbg[badv, 10].view(np.uint32)
        print("   bad lanes histogram (lane:count):", {int(l): int(c) for l, c in enumerate(lanes) if c})
        print("   distinct waves (hwid, t_begin):", len(set(zip(hw.tolist(), tb.tolist()))), " distinct CUs:", len(set(((h >> 8) & 0xf, (h >> 12) & 0x1, (h >> 13) & 0x7, (h >> 16) & 0xf) for h in hw.tolist())), " wave slots:", sorted(set(int(h & 0xf) for h in hw)), " t_begin spread (ticks):", int(tb.max()) - int(tb.min()))
        print(f"iter {it}: clip {dc.size} attr {da.size} dwords differ; attr .w only: {wonly}; first attr float4 {da[0] // 4 if da.size else -1} (hyp {da[0] // 4 // V if da.size else -1}, vertex {da[0] // 4 % V if da.size else -1}), span {(da[-1] - da[0]) // 4 + 1 if da.size else 0} float4")
stop = True; th.join()
print(f"MODE {MODE}: bad {bad} of {ITERS}")
