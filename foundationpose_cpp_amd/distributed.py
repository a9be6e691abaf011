"""Hypothesis sharding over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the orchestration tests).

Every rank owns the same frame, mesh and weights and a contiguous slice of the hypothesis grid.  Render, crop,
refine-net, pose update and the score-net trunk + per-hypothesis self-attention are independent per hypothesis; the
only exchange is ONE all-gather of the pooled score features [n_local,512] (+ the refined poses [n_local,16] riding
along), after which every rank evaluates the cross-hypothesis attention + Linear + arg-max redundantly and therefore
agrees on the winner without a second collective.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_range(n_total: int, world: int, rank: int):
    """contiguous slices of ceil(n/world) hypotheses; the last ranks may get fewer (or none)."""
    per = -(-n_total // world)
    b = min(rank * per, n_total)
    return b, min(per, n_total - b)


class HipShardBackend:
    """shard_begin / shard_finish over the C ABI with torch CUDA tensors as the exchange buffers."""

    def __init__(self, model, device):
        import torch
        self.m, self.dev, self.torch = model, device, torch
        self._hip = C.CDLL("libamdhip64.so")

    def _d2d(self, dst_ptr, src_ptr, nbytes):
        rc = self._hip.hipMemcpy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes), 3)
        assert rc == 0, f"hipMemcpy D2D failed ({rc})"

    def shard_begin(self, rgb_dev, depth_dev, mask_dev, H, W, name, refine_itr, begin, count):
        torch = self.torch
        feat = torch.zeros((count, 512), dtype=torch.float32, device=self.dev)
        poses = torch.zeros((count, 16), dtype=torch.float32, device=self.dev)
        if count > 0:
            fp_, pp_ = C.c_void_p(), C.c_void_p()
            self.m._must(self.m._L.fp_register_shard_begin(
                self.m.handle, C.c_void_p(rgb_dev.data_ptr()), C.c_void_p(depth_dev.data_ptr()),
                C.c_void_p(mask_dev.data_ptr()), 1, H, W, name.encode(), refine_itr, begin, count,
                C.byref(fp_), C.byref(pp_)))
            self.m.synchronize()
            self._d2d(feat.data_ptr(), fp_.value, count * 512 * 4)
            self._d2d(poses.data_ptr(), pp_.value, count * 16 * 4)
        return feat, poses

    def shard_finish(self, all_feat, all_poses):
        self.torch.cuda.synchronize(self.dev)
        n = all_feat.shape[0]
        out = np.zeros(16, np.float32)
        idx = C.c_int(-1)
        self.m._must(self.m._L.fp_register_shard_finish(
            self.m.handle, C.c_void_p(all_feat.data_ptr()), C.c_void_p(all_poses.data_ptr()), n,
            out.ctypes.data_as(C.c_void_p), C.byref(idx), None))
        return out, idx.value


def sharded_register(backend, dist, n_total, rgb, depth, mask, H, W, name, refine_itr=1):
    """One Register over `n_total` hypotheses sharded across dist.get_world_size() ranks.
    Returns (pose16 column-major, winning global hypothesis index); identical on every rank."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_total // world)
    begin, count = shard_range(n_total, world, rank)
    feat, poses = backend.shard_begin(rgb, depth, mask, H, W, name, refine_itr, begin, count)
    # fixed-size slots so a single all_gather_into_tensor works for ragged last shards
    slot_f = torch.zeros((per, 512), dtype=feat.dtype, device=feat.device)
    slot_p = torch.zeros((per, 16), dtype=poses.dtype, device=poses.device)
    slot_f[:count] = feat
    slot_p[:count] = poses
    packed = torch.cat([slot_f, slot_p], dim=1).contiguous()          # ONE collective: [per, 528]
    gathered = torch.empty((world * per, 528), dtype=packed.dtype, device=packed.device)
    if world > 1:
        dist.all_gather_into_tensor(gathered, packed)
    else:
        gathered.copy_(packed)
    rows = torch.cat([gathered[r * per: r * per + shard_range(n_total, world, r)[1]] for r in range(world)], dim=0)
    all_feat = rows[:, :512].contiguous()
    all_poses = rows[:, 512:].contiguous()
    return backend.shard_finish(all_feat, all_poses)
