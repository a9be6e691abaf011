"""Hypothesis sharding over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the orchestration tests).

Every rank owns the same frame, mesh and weights and a contiguous slice of the hypothesis grid.  Render, crop,
refine-net, pose update and the score-net trunk + per-hypothesis self-attention are independent per hypothesis; the
only exchange is ONE all-gather of one row per hypothesis, [pooled score feature 512 | refined pose 16] f32, after which
every rank evaluates the cross-hypothesis attention + Linear + arg-max redundantly and therefore agrees on the winner
without a second collective.

No host stalls on the way (round 2): `shard_begin_packed` only enqueues on the library's stream and writes the rows
straight into a persistent send buffer; the collective is ordered behind it with an event (no synchronize, no copies, no
per-call allocations), and the library's stream waits for the collective the same way before the redundant finish, which
ends in the Register's single synchronisation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ROW = 528  # floats per exchanged row: 512 feature + 16 pose


def shard_range(n_total: int, world: int, rank: int):
    """contiguous slices of ceil(n/world) hypotheses; the last ranks may get fewer (or none)."""
    per = -(-n_total // world)
    b = min(rank * per, n_total)
    return b, min(per, n_total - b)


class HipShardBackend:
    """The packed shard protocol over the C ABI with torch CUDA tensors as the (persistent) exchange buffers."""

    def __init__(self, model, device):
        import torch
        self.m, self.dev, self.torch = model, device, torch
        self._bufs = {}
        self._lib_stream = torch.cuda.ExternalStream(model.stream, device=device)   # the library's non-blocking stream

    def buffers(self, per: int, world: int):
        key = (per, world)
        if key not in self._bufs:
            t = self.torch
            self._bufs[key] = (t.zeros((per, ROW), dtype=t.float32, device=self.dev),
                               t.zeros((world * per, ROW), dtype=t.float32, device=self.dev))
        return self._bufs[key]

    def shard_begin_packed(self, rgb_dev, depth_dev, mask_dev, H, W, name, refine_itr, begin, count, packed, per):
        self.m._must(self.m._L.fp_register_shard_begin_packed(
            self.m.handle, C.c_void_p(rgb_dev.data_ptr()), C.c_void_p(depth_dev.data_ptr()),
            C.c_void_p(mask_dev.data_ptr()), 1, H, W, name.encode(), refine_itr, begin, count,
            C.c_void_p(packed.data_ptr()), per))

    def before_collective(self):
        # the collective (on torch's current stream) starts when the library's stream has produced the rows
        self.torch.cuda.current_stream(self.dev).wait_event(self._lib_stream.record_event())

    def after_collective(self):
        self._lib_stream.wait_event(self.torch.cuda.current_stream(self.dev).record_event())

    def shard_finish_packed(self, gathered, n_total):
        out = np.zeros(16, np.float32)
        idx = C.c_int(-1)
        self.m._must(self.m._L.fp_register_shard_finish_packed(
            self.m.handle, C.c_void_p(gathered.data_ptr()), n_total, out.ctypes.data_as(C.c_void_p), C.byref(idx)))
        return out, idx.value


def sharded_register(backend, dist, n_total, rgb, depth, mask, H, W, name, refine_itr=1):
    """One Register over `n_total` hypotheses sharded across dist.get_world_size() ranks.
    Returns (pose16 column-major, winning global hypothesis index); identical on every rank -- also in failure: a sampler
    failure (bad mask) is found by every rank's own sampler run (ranks with an empty shard included), and a rank whose begin
    raises still joins the collective with NaN rows, which every other rank's finish reports.

    `backend` provides buffers / shard_begin_packed / before_collective / after_collective / shard_finish_packed
    (HipShardBackend on MI355X; tests/test_distributed_cpu.py has a numpy stand-in for the gloo runs)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_total // world)
    begin, count = shard_range(n_total, world, rank)
    packed, gathered = backend.buffers(per, world)          # fixed-size slots: one collective also for a ragged last shard
    failure = None
    try:
        backend.shard_begin_packed(rgb, depth, mask, H, W, name, refine_itr, begin, count, packed, per)
    except Exception as e:      # bad arguments / allocation failure on THIS rank: the others are already heading for the
        failure = e             # collective -- join it with poisoned (NaN) rows so that nobody hangs, raise afterwards
        packed.fill_(float("nan"))
    backend.before_collective()
    if world > 1:
        dist.all_gather_into_tensor(gathered, packed)       # THE collective: [per, 528] f32 per rank
    else:
        gathered.copy_(packed)
    backend.after_collective()
    if failure is not None:
        raise failure
    # contiguous shards of `per` rows: the gathered rows are already in global hypothesis order, padding only at the end.
    # NaN rows of a failed rank make every other rank's finish fail too ("scores are not finite ...", fp_register_shard_finish)
    return backend.shard_finish_packed(gathered, n_total)


# ---------------------------------------------------------------------------------------------------------------------
# The native path (round 4): the library issues the ncclAllGather itself (fp_register_sharded, include/foundationpose_amd.h) on its own
# stream -- no torch tensors, no events across runtimes.  A Python host only has to hand over an RCCL communicator; torch.distributed
# does not expose its ncclComm_t, so one is made here with the SAME librccl the process already holds (torch's bundled copy, which is
# also the one the library binds with dlopen(RTLD_NOLOAD)): rank 0 draws a ncclUniqueId, torch.distributed broadcasts its 128 bytes
# (control plane only), every rank calls ncclCommInitRank.
class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class NativeRcclComm:
    def __init__(self, dist, device):
        import os
        import torch
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        uid = _NcclUniqueId()
        err = None
        try:    # everything that can fail on ONE rank happens before any rank enters ncclCommInitRank (a rendezvous: the others would wait for ever)
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            self._lib = C.CDLL(path if os.path.exists(path) else "librccl.so.1")
            L = self._lib
            L.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
            L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
            L.ncclCommDestroy.argtypes = [C.c_void_p]
            L.ncclGetErrorString.restype = C.c_char_p
            if rank == 0:
                self._check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        except Exception as e:      # noqa: BLE001 -- reported below, on every rank
            err = e
        if world > 1:
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                raise RuntimeError(f"NativeRcclComm: a rank could not prepare its RCCL communicator (this rank: {err!r}); none was created")
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=device)
            dist.broadcast(t, 0)
            C.memmove(C.byref(uid), bytes(t.cpu().tolist()), 128)
        elif err is not None:
            raise err
        torch.cuda.set_device(device)
        self.comm = C.c_void_p()
        self._check(L.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")
        self.world, self.rank = world, rank

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self._lib.ncclGetErrorString(rc).decode()}")

    def close(self):
        if self.comm:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def sharded_register_native(model, comm: NativeRcclComm, rgb, depth, mask, H, W, name, refine_itr=1):
    """One Register sharded over comm.world ranks through fp_register_sharded (device-resident frame tensors).  -> (pose16, index)"""
    out = np.zeros(16, np.float32)
    idx = C.c_int(-1)
    model._must(model._L.fp_register_sharded(model.handle, comm.comm, C.c_void_p(rgb.data_ptr()), C.c_void_p(depth.data_ptr()),
                                             C.c_void_p(mask.data_ptr()), 1, H, W, name.encode(), refine_itr,
                                             out.ctypes.data_as(C.c_void_p), C.byref(idx)))
    return out, idx.value
