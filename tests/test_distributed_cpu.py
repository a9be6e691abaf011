"""world_size-2 gloo test of the N>1 orchestration (foundationpose_cpp_amd/distributed.py): contiguous hypothesis
shards, ONE all-gather, redundant finish on every rank -> every rank agrees with the single-process result.

The HIP library cannot run on CPU, so a numpy stand-in backend supplies deterministic per-hypothesis features; what is
under test is the sharding arithmetic, the packed all-gather (incl. ragged last shard) and the agreement property."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from foundationpose_cpp_amd.distributed import shard_range, sharded_register


class FakeBackend:
    """features / poses are a fixed function of the GLOBAL hypothesis index; finish = arg-max of a cross-row score."""

    def __init__(self, n_total):
        rng = np.random.default_rng(0)
        self.feat = rng.normal(size=(n_total, 512)).astype(np.float32)
        self.poses = rng.normal(size=(n_total, 16)).astype(np.float32)
        self.calls = []

    def shard_begin(self, rgb, depth, mask, H, W, name, itr, begin, count):
        self.calls.append((begin, count))
        return torch.from_numpy(self.feat[begin:begin + count].copy()), torch.from_numpy(self.poses[begin:begin + count].copy())

    def shard_finish(self, all_feat, all_poses):
        f = all_feat.numpy()
        # cross-hypothesis dependence (like att_cross): score depends on the mean over ALL rows
        s = f @ f.mean(0)
        idx = int(np.argmax(s))
        return all_poses.numpy()[idx].copy(), idx


def test_shard_range_partitions():
    for n, w in [(252, 1), (252, 8), (1008, 8), (8, 3), (5, 8), (2016, 8)]:
        spans = [shard_range(n, w, r) for r in range(w)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for b, c in spans:
            assert b == min(pos, n) or c == 0
            pos += c
    assert shard_range(1008, 8, 3) == (378, 126)
    assert shard_range(252, 8, 7) == (224, 28)       # ceil(252/8)=32 per rank, last rank ragged


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = FakeBackend(n_total)
    pose, idx = sharded_register(be, dist, n_total, None, None, None, 480, 640, "obj", 1)
    q.put((rank, idx, pose.tolist(), be.calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [252, 45])
def test_two_rank_gloo_agrees_with_single_process(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    be = FakeBackend(n_total)
    ref_pose, ref_idx = be.shard_finish(torch.from_numpy(be.feat), torch.from_numpy(be.poses))
    for rank, idx, pose, calls in res:
        assert idx == ref_idx
        np.testing.assert_array_equal(np.asarray(pose, np.float32), ref_pose)
        assert calls == [shard_range(n_total, 2, rank)]


def test_single_process_world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        be = FakeBackend(100)
        pose, idx = sharded_register(be, dist, 100, None, None, None, 480, 640, "obj", 1)
        assert idx == be.shard_finish(torch.from_numpy(be.feat), torch.from_numpy(be.poses))[1]
    finally:
        dist.destroy_process_group()
