"""GPU: the multi-rank pipeline of embedding_amd/distributed.py end to end — two processes over gloo sharing this one
GPU (torch.distributed all-reduce / all-gather on device tensors, the C-ABI partition export/import, the block schedule).
Real multi-GPU runs use the same code with backend nccl (= RCCL over xGMI); reference call site: J/DeepWalk.java:79."""
import os
import socket

import numpy as np
import pytest

from helpers import bits, cosine_rows, layered_graph

pytestmark = pytest.mark.gpu

R, T, N_WALKS, DIM = 40, 6, 1200, 32


def _graph(dge):
    src, dst, w, sources = layered_graph(R=R, T=T, deg=5, seed=0)
    g = dge.DeviceGraph(0)
    g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    return g


def _rank(rank, world, port, out_dir, batch_walks, workers, hs=False):
    import torch.distributed as dist
    import embedding_amd as dge
    from embedding_amd.distributed import fit_distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _graph(dge)
    cfg = dge.make_config(DIM, T, R * T, workers=workers, table_size=20011, use_hs=hs)
    from embedding_amd import distributed as D
    m = fit_distributed(g, N_WALKS, T, cfg, world, rank, walk_seed=11, batch_walks=batch_walks)
    # the module waited for the device after the count all-reduce and after each final gather — never inside an episode (the hand-off is stream-ordered)
    assert D.HOST_WAITS["n"] == 1 + (3 if hs else 2), D.HOST_WAITS
    syn0, vid = m.vectors()
    np.savez(os.path.join(out_dir, "d%d.npz" % rank), syn0=syn0, syn1neg=m.syn1neg(), vid=vid, pairs=m.stats()["pairs"], syn1=m.syn1() if hs else np.zeros(0, np.float32))
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, batch_walks, workers, hs=False):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank, args=(2, port, str(tmp_path), batch_walks, workers, hs), nprocs=2, join=True)
    return [np.load(str(tmp_path / ("d%d.npz" % i))) for i in range(2)]


@pytest.mark.parametrize("hs", [False, True], ids=["sgns", "hs"])
def test_two_rank_fit_is_the_oracles_block_run(dge, oracle, tmp_path, hs):
    """One batch = the whole epoch, in-order workers: both ranks end bit-identical to the oracle's sequential run of the 2x2 blocks — with the
    hierarchical-softmax term as well (what DeepWalk's builder default trains, J/DeepWalk.java:73-76): the syn1 partitions travel the ring with syn1neg's."""
    r = _run(tmp_path, N_WALKS, 1, hs)
    walks = _graph(dge).sample_walks(N_WALKS, T, seed=11, rng_mode=1)
    om = oracle.train_sgns(walks, R * T, DIM, T, table_size=20011, arith=1, part_n=2, use_hs=hs)
    for d in r:
        assert np.array_equal(d["vid"], om.vocab_ids)
        assert np.array_equal(bits(d["syn0"]), bits(om.syn0)) and np.array_equal(bits(d["syn1neg"]), bits(om.syn1neg))
        assert not hs or np.array_equal(bits(d["syn1"]), bits(om.syn1))


def test_two_rank_fit_in_batches_matches_one_process(dge, oracle, tmp_path):
    """Several batches per epoch (exact learning-rate bookkeeping across batches): the two processes reproduce, bit for bit,
    the same schedule run with two models inside one process; device-filling workers stay within Hogwild noise of it."""
    import torch
    from helpers import simulate_block_schedule, simulate_gather_syn0
    g = _graph(dge)
    cfg = dge.make_config(DIM, T, R * T, workers=1, table_size=20011)
    corpus = g.sample_walks_device(N_WALKS, T, seed=11, rng_mode=1)
    counts = torch.zeros(R * T, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(R * T, counts)
    ms = [dge.SgnsModel.create(cfg, counts, 0) for _ in range(2)]
    nb, wb = 500, 0
    for b0 in range(0, N_WALKS, nb):
        n = min(nb, N_WALKS - b0)
        for m in ms:
            m.reset_stats()
        simulate_block_schedule(ms, lambda m: m.train(corpus, b0, n, walk_index_base=b0, words_before=wb, total_walks=N_WALKS))
        wb += ms[0].stats()["words"]
    simulate_gather_syn0(ms)
    want0, want1 = ms[0].vectors()[0], ms[0].syn1neg()
    r = _run(tmp_path, nb, 1)
    for d in r:
        assert np.array_equal(bits(d["syn0"]), bits(want0)) and np.array_equal(bits(d["syn1neg"]), bits(want1))
    (tmp_path / "hog").mkdir()
    h = _run(tmp_path / "hog", nb, 64)
    assert np.array_equal(bits(h[0]["syn0"]), bits(h[1]["syn0"]))
    assert float(np.median(cosine_rows(h[0]["syn0"], want0))) > 0.99


def _rank_nccl(rank, world, port, out_dir):
    import ctypes as C
    import torch
    import torch.distributed as dist
    import embedding_amd as dge
    from embedding_amd.distributed import fit_distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:%d" % rank))
    src, dst, w, sources = layered_graph(R=R, T=T, deg=5, seed=0)
    g = dge.DeviceGraph(rank); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    cfg = dge.make_config(DIM, T, R * T, workers=1, table_size=20011)
    m = fit_distributed(g, N_WALKS, T, cfg, world, rank, walk_seed=11, batch_walks=N_WALKS, device=rank)
    syn0, vid = m.vectors()
    np.savez(os.path.join(out_dir, "n%d.npz" % rank), syn0=syn0, syn1neg=m.syn1neg(), vid=vid)
    # the library's own RCCL calls (hosts without torch.distributed): the id travels through torch here, any channel would do
    uid = (C.c_char * 128)()
    if rank == 0:
        dge._native.check(dge.lib.dge_comm_unique_id(uid))
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).cuda(rank)
    dist.broadcast(t, 0)
    uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
    comm = C.c_void_p(0)
    dge._native.check(dge.lib.dge_comm_create(C.byref(comm), uid, rank, world, rank))
    corpus = g.sample_walks_device(N_WALKS, T, seed=11, rng_mode=1)
    counts = torch.zeros(R * T, dtype=torch.int64, device="cuda:%d" % rank); corpus.count_tokens(R * T, counts)
    m2 = dge.SgnsModel.create(cfg, counts, rank)
    for e in range(world):
        m2.set_partition(world, rank, (rank + e) % world)
        m2.train(corpus, total_walks=N_WALKS)
        dge._native.check(dge.lib.dge_model_ring_pass(m2._h, comm, e))
    m2.set_partition(1)
    for table in (0, 1):
        dge._native.check(dge.lib.dge_model_gather_table(m2._h, comm, table))
    np.savez(os.path.join(out_dir, "c%d.npz" % rank), syn0=m2.vectors()[0], syn1neg=m2.syn1neg())
    dge.lib.dge_comm_free(comm)
    dist.barrier()
    dist.destroy_process_group()


def test_two_gpus_over_rccl(dge, oracle, tmp_path):
    """TWO GPUs, backend nccl (= RCCL): fit_distributed over torch.distributed's point-to-point ring, and the library's own
    dge_comm_* / dge_model_ring_pass / dge_model_gather_table, both against the oracle's sequential run of the 2 x 2 blocks, bit for bit.
    Skipped on a one-GPU box (the build boxes): the first place this can run is a multi-GPU node."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank_nccl, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    walks = _graph(dge).sample_walks(N_WALKS, T, seed=11, rng_mode=1)
    om = oracle.train_sgns(walks, R * T, DIM, T, table_size=20011, arith=1, part_n=2)
    for tag in ("n", "c"):
        for r in range(2):
            d = np.load(str(tmp_path / ("%s%d.npz" % (tag, r))))
            assert np.array_equal(bits(d["syn0"]), bits(om.syn0)) and np.array_equal(bits(d["syn1neg"]), bits(om.syn1neg)), (tag, r)


def test_an_episode_of_the_block_schedule_makes_no_blocking_wait_in_the_library(dge):
    """dge_host_sync_count counts every blocking wait the library makes (stream / device / event synchronisations, blocking copies).  One rank's share of a 4-rank
    schedule under the owner-computes kernels (what auto resolves to in a block of a flat vocabulary), the hand-off stream-ordered on the model's stream: a global batch
    costs its item store's two read-backs in episode 0, and the episodes after it none — the host runs ahead of the device through the whole batch."""
    import torch
    rng = np.random.default_rng(5)
    V, L, n, N = 40_000, 12, 60_000, 4
    corpora = [dge.WalkCorpus.from_host(rng.integers(0, V, (n, L)).astype(np.int32), 0) for _ in range(2)]
    counts = torch.zeros(V, dtype=torch.int64, device="cuda:0"); corpora[0].count_tokens(V, counts)
    m = dge.SgnsModel.create(dge.make_config(64, L, V, negative=5, workers=0, epochs=1, seed=3, min_count=1), counts, 0)
    buf = torch.empty(m.partition_floats(N), dtype=torch.float32, device="cuda:0")

    def batch(corpus):
        waits = []
        for e in range(N):
            m.set_partition(N, 0, e)
            c0 = dge.host_sync_count()
            m.train(corpus)
            st = m.stream()
            m.export_partition_async(1, N, e, buf, st); m.import_partition_async(1, N, e, buf, st)
            waits.append(dge.host_sync_count() - c0)
        m.set_partition(1)
        return waits

    batch(corpora[0])                                        # (first use: work buffers)
    w1, w2 = batch(corpora[1]), batch(corpora[0])
    pairs = m.stats()["pairs"]
    assert m.schedule()["update_policy"] == 8 and "one block" in m.kernel()
    assert w1[1:] == [0] * (N - 1) and w2[1:] == [0] * (N - 1), (w1, w2)
    assert 0 < w1[0] <= 3 and 0 < w2[0] <= 3, (w1, w2)
    assert pairs > 0 and np.isfinite(m.vectors()[0][:1000]).all()
