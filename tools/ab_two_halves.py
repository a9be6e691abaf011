"""Does a Register of 252 hypotheses finish sooner as K slices on K streams (K models on one GPU, one thread each) than as one
batch on one stream?  The slices' memory-path-bound phases (transformer heads, stem, render) can run under another slice's MFMA-bound
trunk, and a slice's partial last round of workgroups can be filled by the other stream.
    python tools/ab_two_halves.py [K ...]     (default 2 3 4)
Prints ms per (whole) Register for the one-stream form and for every K (begin only: the cross-hypothesis head is 0.1 ms either way).
"""
import os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
from foundationpose_cpp_amd.distributed import HipShardBackend

ks = [int(a) for a in sys.argv[1:]] or [2, 3, 4]
dev = torch.device("cuda", 0)
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
rgb, depth, mask = (torch.from_numpy(a).to(dev) for a in (scene.rgb, scene.depth, scene.mask))
H, Wd = scene.depth.shape
N = 252
kmax = max(ks)
models = [FoundationPose(mesh, scene.K, rp, sp) for _ in range(kmax)]
bes = [HipShardBackend(m, dev) for m in models]


def run(K, iters=20, warm=3, lock=False):
    per = (N + K - 1) // K
    bufs = [bes[r].buffers(per, 1)[0] for r in range(K)]
    bar = threading.Barrier(K + 1); step = threading.Barrier(K)
    def worker(r):
        b0 = r * per; cnt = max(0, min(per, N - b0))
        for it in range(warm + iters):
            if it == warm: bar.wait(); bar.wait()
            if lock and it >= warm: step.wait()
            bes[r].shard_begin_packed(rgb, depth, mask, H, Wd, mesh.name, 1, b0, cnt, bufs[r], per)
            models[r].synchronize()
            if lock and it >= warm: step.wait()
        bar.wait()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(K)]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait()
    bar.wait(); t1 = time.perf_counter()
    for t in th: t.join()
    return (t1 - t0) / iters * 1e3


for rep in range(2):
    print(f"one stream, 252 hypotheses: {run(1):.3f} ms per Register", flush=True)
    for K in ks:
        print(f"{K} streams x {(N + K - 1) // K} hypotheses: free-running {run(K):.3f} ms, all slices of a Register started together {run(K, lock=True):.3f} ms per Register", flush=True)
