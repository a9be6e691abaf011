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
    """features / poses are a fixed function of the GLOBAL hypothesis index; finish = arg-max of a cross-row score.
    Implements the packed shard protocol of foundationpose_cpp_amd.distributed (rows [feature 512 | pose 16])."""

    def __init__(self, n_total):
        rng = np.random.default_rng(0)
        self.feat = rng.normal(size=(n_total, 512)).astype(np.float32)
        self.poses = rng.normal(size=(n_total, 16)).astype(np.float32)
        self.calls = []
        self.order = []
        self._bufs = {}

    def buffers(self, per, world):
        if (per, world) not in self._bufs:
            self._bufs[(per, world)] = (torch.full((per, 528), 7.0), torch.full((world * per, 528), 9.0))   # stale garbage on purpose
        return self._bufs[(per, world)]

    def shard_begin_packed(self, rgb, depth, mask, H, W, name, itr, begin, count, packed, per):
        if name == "fail-on-rank-1" and begin > 0:
            raise RuntimeError("allocation failed on this rank")
        self.calls.append((begin, count))
        self.order.append("begin")
        packed.zero_()
        packed[:count, :512] = torch.from_numpy(self.feat[begin:begin + count])
        packed[:count, 512:] = torch.from_numpy(self.poses[begin:begin + count])

    def before_collective(self):
        self.order.append("before")

    def after_collective(self):
        self.order.append("after")

    def shard_finish_packed(self, gathered, n_total):
        self.order.append("finish")
        rows = gathered.numpy()[:n_total]
        f = rows[:, :512]
        if not np.isfinite(f).all():      # what fp_register_shard_finish reports for poisoned rows
            raise RuntimeError("scores are not finite (a rank of a sharded Register reported a failed shard)")
        # cross-hypothesis dependence (like att_cross): score depends on the mean over ALL rows
        s = f @ f.mean(0)
        idx = int(np.argmax(s))
        return rows[idx, 512:].copy(), idx

    def reference(self):
        s = self.feat @ self.feat.mean(0)
        idx = int(np.argmax(s))
        return self.poses[idx].copy(), idx


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
    pose2, idx2 = sharded_register(be, dist, n_total, None, None, None, 480, 640, "obj", 1)   # persistent buffers are reused
    assert idx2 == idx and np.array_equal(pose2, pose) and len(be._bufs) == 1
    assert be.order[:4] == ["begin", "before", "after", "finish"]
    q.put((rank, idx, pose.tolist(), be.calls[:1]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [252, 45, 1])
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
    ref_pose, ref_idx = be.reference()
    for rank, idx, pose, calls in res:
        assert idx == ref_idx
        np.testing.assert_array_equal(np.asarray(pose, np.float32), ref_pose)
        assert calls == [shard_range(n_total, 2, rank)]


def _failing_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = FakeBackend(252)
    try:
        sharded_register(be, dist, 252, None, None, None, 480, 640, "fail-on-rank-1", 1)
        q.put((rank, "returned"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    # the group is still usable: nobody was left behind in the collective
    pose, idx = sharded_register(be, dist, 252, None, None, None, 480, 640, "obj", 1)
    assert idx == be.reference()[1]
    dist.destroy_process_group()


def test_a_failing_rank_does_not_hang_the_others():
    """rank 1's begin raises before the all-gather: it must still join the collective (NaN rows); rank 0's finish then
    fails as well instead of returning a pose computed from garbage, and both raise"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "allocation failed" in res[1] and "not finite" in res[0], res


def test_single_process_world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        be = FakeBackend(100)
        pose, idx = sharded_register(be, dist, 100, None, None, None, 480, 640, "obj", 1)
        assert idx == be.reference()[1]
    finally:
        dist.destroy_process_group()
