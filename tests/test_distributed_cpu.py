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


class FakeBlockModel:
    """numpy stand-in for the block schedule: 'training' a block adds 1 to every syn0 row of the context partition and to
    every syn1neg row of the target partition, and records the block — a row touched by two ranks in one episode, or a
    block trained twice, shows up as a count != the expected one."""

    def __init__(self, V, stride, hs=False):
        import types
        self.V, self.stride = V, stride
        self.tab = [np.zeros((V, stride), np.float32) for _ in range(3 if hs else 2)]      # syn0, syn1neg (, syn1: its partitions travel with syn1neg's)
        self.cfg = types.SimpleNamespace(use_hs=int(hs))
        self.part = (1, 0, 0)
        self.blocks = []

    def set_partition(self, n, ctx=0, tgt=0):
        self.part = (n, ctx, tgt)

    def partition_floats(self, n):
        return (self.V + n - 1) // n * self.stride

    def train(self):
        n, ctx, tgt = self.part
        self.blocks.append((ctx, tgt))
        self.tab[0][ctx::n] += 1
        for t in self.tab[1:]:
            t[tgt::n] += 1

    def export_partition(self, table, n, part, buf):
        rows = self.tab[table][part::n]
        out = np.zeros(((self.V + n - 1) // n, self.stride), np.float32); out[:len(rows)] = rows
        buf.copy_(__import__("torch").from_numpy(out.reshape(-1)))

    def import_partition(self, table, n, part, buf):
        rows = len(self.tab[table][part::n])
        self.tab[table][part::n] = buf.numpy().reshape(-1, self.stride)[:rows]


class FakeOrderedBlockModel(FakeBlockModel):
    """... with the stream-ordered entry points of the real model (dge_model_export/import_partition_async): the schedule must use THOSE and nothing that waits."""
    stream_ordered = True

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.calls = []

    def export_partition_async(self, table, n, part, buf, stream):
        self.calls.append(("export_async", table, part)); FakeBlockModel.export_partition(self, table, n, part, buf)

    def import_partition_async(self, table, n, part, buf, stream):
        self.calls.append(("import_async", table, part)); FakeBlockModel.import_partition(self, table, n, part, buf)

    def export_partition(self, *a):
        self.calls.append(("export_BLOCKING",)); super().export_partition(*a)

    def import_partition(self, *a):
        self.calls.append(("import_BLOCKING",)); super().import_partition(*a)

    def train(self):
        self.calls.append(("train",) + tuple(self.part[1:])); super().train()


def _ordered_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from embedding_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = FakeOrderedBlockModel(11, 4)
    tr = D.RingTransport.choose(world, rank, dist, device="cpu")
    waits = D.HOST_WAITS["n"]
    bufs = D.block_schedule_step(m, m.train, world, rank, transport=tr)
    D.block_schedule_step(m, m.train, world, rank, *bufs, transport=tr)
    # inside the episodes: one stream-ordered export and one import per episode, in that order behind the episode's training — never the blocking pair, never a device wait
    want = []
    for _ in range(2):
        for e in range(world):
            want += [("train", rank, (rank + e) % world), ("export_async", 1, (rank + e) % world), ("import_async", 1, (rank + 1 + e) % world)]
    ok = m.calls == want and D.HOST_WAITS["n"] == waits
    open(os.path.join(out_dir, "o%d.txt" % rank), "w").write("ok" if ok else "calls %r waits %d -> %d" % (m.calls, waits, D.HOST_WAITS["n"]))
    dist.barrier()
    dist.destroy_process_group()


def test_block_schedule_episodes_make_no_host_wait(tmp_path):
    """Round-4 verdict: the hand-off was host-synchronous three times per episode.  Now the schedule calls the model's stream-ordered export / import (the real ones order
    libdge's stream and torch's with events: tests/test_gpu_distributed.py counts the library's own blocking waits) and this module waits for no device inside an episode."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ordered_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    got = [open(str(tmp_path / ("o%d.txt" % i))).read() for i in range(3)]
    assert got == ["ok"] * 3, got


def _block_worker(rank, world, port, out_dir, transport=None, hs=False):
    import torch.distributed as dist
    from embedding_amd.distributed import block_schedule_step, gather_table
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = FakeBlockModel(11, 4, hs)                  # 11 rows: partitions of unequal size (padding in the packed buffers)
    bufs = (None, None)
    for _ in range(2):                             # two global batches
        bufs = block_schedule_step(m, m.train, world, rank, *bufs, transport=transport)
    held = m.tab[1][rank::world].copy()            # ring invariant: after a whole batch, partition `rank` of syn1neg is home again
    gather_table(m, 0, world, rank)
    gather_table(m, 1, world, rank)
    if hs:
        gather_table(m, 2, world, rank)
        assert np.array_equal(m.tab[2], m.tab[1])  # the inner-node partitions went round with the syn1neg partitions
    assert np.array_equal(m.tab[1][rank::world], held)
    np.savez(os.path.join(out_dir, "b%d.npz" % rank), syn0=m.tab[0], syn1neg=m.tab[1], blocks=np.array(m.blocks), part=np.array(m.part))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,transport,hs", [(2, None, False), (3, None, False), (3, "allgather", False), (3, None, True)])
def test_block_schedule_gloo(tmp_path, world, transport, hs):
    """Every (context partition, centre partition) block is trained exactly once per batch, by exactly one rank; the blocks
    of one episode are row-disjoint; all ranks end with identical tables (hs: the inner-node table's partitions travel along)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_block_worker, args=(world, port, str(tmp_path), transport, hs), nprocs=world, join=True)
    r = [np.load(str(tmp_path / ("b%d.npz" % i))) for i in range(world)]
    seen = []
    for i in range(world):
        blocks = [tuple(b) for b in r[i]["blocks"]]
        assert len(blocks) == 2 * world and all(b[0] == i for b in blocks)          # a rank only ever trains its own syn0 partition
        seen += blocks[:world]
        assert tuple(r[i]["part"]) == (1, 0, 0)                                      # the filter is switched off afterwards
    assert sorted(seen) == [(a, b) for a in range(world) for b in range(world)]
    for e in range(world):                                                            # one episode: target partitions are all different
        assert sorted(tuple(r[i]["blocks"][e])[1] for i in range(world)) == list(range(world))
    # each syn1neg row was trained `world` times per batch (once per context partition), each syn0 row likewise
    for i in range(world):
        assert np.array_equal(r[i]["syn1neg"], np.full((11, 4), 2.0 * world, np.float32))
        assert np.array_equal(r[i]["syn0"], np.full((11, 4), 2.0 * world, np.float32))


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


class _FaultyDist:
    """torch.distributed with point-to-point transfers refused (NotImplementedError) on the ranks in `bad`."""

    def __init__(self, dist, rank, bad):
        self._d, self._rank, self._bad = dist, rank, bad

    def __getattr__(self, name):
        return getattr(self._d, name)

    def batch_isend_irecv(self, ops):
        if self._rank in self._bad:
            raise NotImplementedError("point-to-point operations are not available (injected)")
        return self._d.batch_isend_irecv(ops)


def _transport_worker(rank, world, port, out_dir, bad, timeout_s):
    import torch.distributed as dist
    from embedding_amd.distributed import RingTransport, block_schedule_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = _FaultyDist(dist, rank, bad)
    outcome = "?"
    try:
        tr = RingTransport.choose(world, rank, d, device="cpu", timeout_s=timeout_s)
        outcome = tr.mode
        m = FakeBlockModel(11, 4)
        block_schedule_step(m, m.train, world, rank, dist_mod=d, transport=tr)
        assert np.array_equal(m.tab[1][rank::world], np.full_like(m.tab[1][rank::world], float(world)))
    except RuntimeError as e:
        outcome = "raised: " + str(e)[:60]
    open(os.path.join(out_dir, "t%d.txt" % rank), "w").write(outcome)
    if not outcome.startswith("raised"):
        dist.barrier()
        dist.destroy_process_group()
    else:
        os._exit(0)        # (a rank whose peer never posted its half of the probe still has that transfer pending: leave without the collective teardown)


@pytest.mark.parametrize("bad,expect", [((), "p2p"), ((0, 1), "allgather"), ((1,), "raised")])
def test_ring_transport_is_chosen_by_all_ranks_together(tmp_path, bad, expect):
    """a rank must never switch transport on its own.  All ranks fine -> p2p; all refused -> all-gather everywhere and the
    schedule still completes; ONE rank refused (its peer's probe can then never complete) -> every rank raises within the time box
    instead of hanging in mismatched collectives."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_transport_worker, args=(2, port, str(tmp_path), tuple(bad), 3.0), nprocs=2, join=True)
    got = [open(str(tmp_path / ("t%d.txt" % i))).read() for i in range(2)]
    assert all(g.startswith(expect) for g in got), got


def test_explicit_transport_is_taken_as_given():
    from embedding_amd.distributed import RingTransport
    assert RingTransport.choose(2, 0, None, requested="allgather").mode == "allgather"
    assert RingTransport.choose(2, 0, None, requested="p2p").mode == "p2p"
    with pytest.raises(ValueError):
        RingTransport("carrier pigeon")
