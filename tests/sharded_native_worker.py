"""Worker of tests/test_sharded_native_gpu.py: the NATIVE sharded Register (fp_register_sharded: begin -> ONE ncclAllGather on the model's
stream -> finish) at world > 1 on ONE GPU.  The ranks are threads of this process, each with its own model on device 0; the collective is
tests/fake_rccl (a test double of librccl whose communicators share a host barrier and copy device-to-device, stream-ordered).  torch must
NOT be in this process: the library binds an RCCL that is already loaded first, and torch brings its own.

   python tests/sharded_native_worker.py <fake librccl.so.1> <world> <inplane_steps> [bad_rank]
prints one JSON line: {"ok": true, ...} or {"ok": false, "why": ...}."""
import ctypes as C
import json
import os
import sys
import tempfile
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

assert "torch" not in sys.modules
fake = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)      # first: the library's dlopen("librccl.so.1", RTLD_NOLOAD) / phdr scan finds THIS copy
from foundationpose_cpp_amd import FoundationPose, _lib, synthetic as syn, weights as W  # noqa: E402

world, steps = int(sys.argv[2]), int(sys.argv[3])
bad_rank = int(sys.argv[4]) if len(sys.argv) > 4 else -1
fake.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
fake.ncclCommDestroy.argtypes = [C.c_void_p]


def main():
    mesh = syn.make_mesh()
    scene = syn.make_scene(mesh)
    d = tempfile.mkdtemp()
    rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
    cal = W.load_calibration(os.path.join(ROOT, "tests", "golden", "disc_calib_seed9.npz"))     # the discriminating set: a unique winner
    W.pack_synthetic("refiner", rp, 9, cal)
    W.pack_synthetic("scorer", sp, 9, cal)
    ref = FoundationPose(mesh, scene.K, rp, sp)
    ref.set_inplane_steps(steps)
    n_total = ref.num_hypotheses
    ok, ref_pose, ref_idx, ref_scores, ref_refined, _ = ref.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok, ref.last_error
    top3 = [int(i) for i in np.argsort(-ref_scores)[:3]]
    models = [FoundationPose(mesh, scene.K, rp, sp) for _ in range(world)]
    for m in models:
        m.set_inplane_steps(steps)
    comms = (C.c_void_p * world)()
    assert fake.ncclCommInitAll(comms, world, None) == 0
    rgb, depth, mask = (np.ascontiguousarray(x) for x in (scene.rgb, scene.depth, scene.mask))
    zero_mask = np.zeros_like(mask)
    L = models[0]._L
    results = [[] for _ in range(world)]

    def call(r, msk):
        out = np.zeros(16, np.float32)
        idx = C.c_int(-1)
        rc = L.fp_register_sharded(models[r].handle, comms[r], rgb.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                                   msk.ctypes.data_as(C.c_void_p), 0, 480, 640, mesh.name.encode(), 1, out.ctypes.data_as(C.c_void_p), C.byref(idx))
        return rc, syn.from_colmajor(out[None])[0], idx.value, _lib.last_error() if rc else ""    # (thread-local in the library: read on the calling thread)

    def rank_thread(r):
        try:
            for _ in range(3):                                     # eager, capture, replay
                results[r].append(call(r, mask))
            if bad_rank >= 0:                                      # ONE rank gets an all-zero mask: every rank must fail, none may hang ...
                results[r].append(call(r, zero_mask if r == bad_rank else mask))
                results[r].append(call(r, mask))                   # ... and the next Register works again on all of them
        except Exception as e:      # noqa: BLE001
            results[r].append((99, None, -1, repr(e)))

    threads = [threading.Thread(target=rank_thread, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    if any(t.is_alive() for t in threads):
        print(json.dumps({"ok": False, "why": "a rank thread hangs"}), flush=True)
        os._exit(1)
    why = []
    per = -(-n_total // world)
    for r in range(world):
        good = [0, 1, 2] + ([4] if bad_rank >= 0 else [])
        if len(results[r]) != (5 if bad_rank >= 0 else 3):
            why.append(f"rank {r}: {len(results[r])} results: {results[r][-1:]}")
            continue
        for i in good:
            rc, pose, idx, err = results[r][i]
            if rc != 0:
                why.append(f"rank {r} call {i}: rc {rc}: {err}")
                continue
            # every rank and every call: the SAME winner and pose, bit for bit (redundant finish on identical gathered rows) ...
            if idx != results[0][0][2] or not np.array_equal(pose, results[0][0][1]):
                why.append(f"rank {r} call {i}: winner {idx} / pose differ from rank 0's first call ({results[0][0][2]})")
            # ... which is the unsharded Register's up to the schedules a shard of this size takes (tests/test_discriminative_gpu.py::
            # test_sharded_register_1008_over_8_ranks holds the emulated form to the same bar): among its top 3, pose of that hypothesis
            dmm = float(np.linalg.norm(pose[:3, 3] - ref_refined[idx][:3, 3]) * 1e3) if 0 <= idx < n_total else 1e9
            ddeg = float(np.degrees(np.arccos(np.clip((np.trace(pose[:3, :3] @ ref_refined[idx][:3, :3].T) - 1) / 2, -1, 1)))) if 0 <= idx < n_total else 1e9
            if idx not in top3 or dmm > 0.1 or ddeg > 0.1:
                why.append(f"rank {r} call {i}: winner {idx} (unsharded top 3 {top3}), refined pose of it off by {dmm:.4f} mm / {ddeg:.4f} deg")
        if bad_rank >= 0:
            rc, _, _, err = results[r][3]
            want = "Mask is all zero" if r == bad_rank else "scores are not finite"
            if rc == 0 or want not in err:
                why.append(f"rank {r} bad-mask call: rc {rc}, error {err!r} (expected {want!r})")
    for r in range(world):
        fake.ncclCommDestroy(comms[r])
    for m in models + [ref]:
        m.close()
    shards = [min(per, max(0, n_total - r * per)) for r in range(world)]
    print(json.dumps({"ok": not why, "why": why, "world": world, "n_total": n_total, "shards": shards, "winner_unsharded": ref_idx,
                      "winner_sharded": results[0][0][2] if results[0] else -1}), flush=True)


main()
