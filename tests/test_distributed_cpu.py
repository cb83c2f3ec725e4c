"""CPU, world_size 2, gloo: the multi-rank plumbing of embedding_amd/distributed.py — shard plan, vocabulary count
all-reduce, delta exchange.  The trainer behind export_delta/import_delta is a numpy stand-in here (the real one
is the GPU model, covered by tests/test_gpu_sgns.py::test_sharded_training_and_delta_exchange)."""
import os
import socket

import numpy as np
import pytest


class FakeModel:
    def __init__(self, n, rank):
        self.snap = np.linspace(-1, 1, n).astype(np.float32)
        self.cur = self.snap + np.float32(0.01 * (rank + 1)) * np.arange(n, dtype=np.float32)

    def sync_size(self):
        return len(self.cur)

    def export_delta(self, buf):
        buf.copy_(__import__("torch").from_numpy(self.cur - self.snap))

    def import_delta(self, buf, scale):
        self.cur = self.snap + np.float32(scale) * buf.numpy()
        self.snap = self.cur.copy()


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from embedding_amd.distributed import allreduce_counts, exchange_deltas, shard_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    epoch_walks, L, NV = 1001, 5, 50
    first, n = shard_plan(epoch_walks, world, rank)
    rng = np.random.default_rng(7)
    corpus = rng.integers(0, NV, (epoch_walks, L))            # same on every rank (strided RNG in the real path)
    mine = corpus[first:first + n]
    counts = torch.from_numpy(np.bincount(mine.reshape(-1), minlength=NV).astype(np.int64))
    allreduce_counts(counts)
    m = FakeModel(64, rank)
    buf = torch.empty(m.sync_size(), dtype=torch.float32)
    exchange_deltas(m, buf, world)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), first=first, n=n, counts=counts.numpy(), cur=m.cur)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_plumbing(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(str(tmp_path / ("r%d.npz" % i))) for i in range(2)]
    # shards: contiguous, disjoint, cover 2*(1001//2) walks
    assert int(r[0]["first"]) == 0 and int(r[1]["first"]) == int(r[0]["n"]) == 500 and int(r[1]["n"]) == 500
    # vocabulary counts agree on both ranks and equal the counts of the covered corpus
    rng = np.random.default_rng(7); corpus = rng.integers(0, 50, (1001, 5))
    want = np.bincount(corpus[:1000].reshape(-1), minlength=50)
    assert np.array_equal(r[0]["counts"], want) and np.array_equal(r[1]["counts"], want)
    # delta exchange: both ranks end on snapshot + mean(delta)
    snap = np.linspace(-1, 1, 64).astype(np.float32)
    mean_delta = np.float32(0.5) * (np.float32(0.01) * np.arange(64, dtype=np.float32) + np.float32(0.02) * np.arange(64, dtype=np.float32))
    assert np.allclose(r[0]["cur"], snap + mean_delta, atol=1e-6) and np.array_equal(r[0]["cur"], r[1]["cur"])


def test_shard_plan_edges():
    from embedding_amd.distributed import shard_plan
    assert shard_plan(10, 1, 0) == (0, 10)
    assert [shard_plan(10, 4, r) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    with pytest.raises(ValueError):
        shard_plan(10, 2, 2)
