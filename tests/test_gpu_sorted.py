"""GPU: update_policy 8, the owner-computes schedule (embedding_amd/csrc/sgns_sorted.hip) — every (context, target, label) term of a
mini-batch becomes an item, items are sorted by target row and applied by the row's owner, then sorted by context row and summed.
No locks, no atomics; the result is a deterministic function of the batch, so the DEVICE-FILLING run is compared bit for bit with the
oracle's sequential restatement of the same schedule (oracle/dge_oracle.c: train_block_sorted).  Reference call site: w2v.fit(),
J/DeepWalk.java:79 (SGNS half of the oracle: a restatement, parity unpinned — DESIGN.md §3)."""
import numpy as np
import pytest

from helpers import bits, cosine_rows, layered_graph, build_both, simulate_block_schedule, simulate_gather_syn0

pytestmark = pytest.mark.gpu


def _walks(oracle, dge, R=40, T=6, n=1500, seed=0):
    src, dst, w, sources = layered_graph(R=R, T=T, deg=5, seed=seed)
    og, dg = build_both(oracle, dge, src, dst, w, sources)
    return dg.sample_walks(n, T, seed=11, rng_mode=1), R * T


@pytest.mark.parametrize("dim,negative,chunk,per", [(32, 5, 256, 0), (128, 5, 256, 100), (20, 5, 7, 37), (64, 0, 16, 50), (100, 20, 33, 200),
                                                    (256, 3, 5, 0), (130, 17, 64, 300), (512, 1, 9, 64)])
def test_owner_computes_bit_exact_at_full_concurrency(dge, oracle, dim, negative, chunk, per):
    walks, NV = _walks(oracle, dge, n=400 if dim > 256 else 1500)
    kw = dict(negative=negative, min_count=2, epochs=2, seed=5, table_size=20011)
    om = oracle.train_sgns(walks, NV, dim, 6, sorted_chunk=chunk, sorted_walks=per, **kw)
    with dge.tuning(sorted_chunk=chunk, **({"sorted_walks": per} if per else {})):
        dm = dge.SgnsModel.fit(walks, dge.make_config(dim, 6, NV, workers=0, update_policy=8, **kw), 0)
    s0, vid = dm.vectors()
    assert dm.schedule()["update_policy"] == 8
    assert np.array_equal(vid, om.vocab_ids) and dm.stats()["pairs"] == om.pairs and dm.stats()["words"] == 2 * om.total_words
    assert np.array_equal(bits(s0), bits(om.syn0)) and np.array_equal(bits(dm.syn1neg()), bits(om.syn1neg))
    # and it is SGNS: with small mini-batches it tracks the sequential word2vec result like any Hogwild run does
    if per and per <= 100 and dim > 2:
        o0 = oracle.train_sgns(walks, NV, dim, 6, arith=0, **kw)
        assert np.median(cosine_rows(s0, o0.syn0)) > 0.99


def test_owner_computes_ragged_walks_holes_and_tiny_vocabularies(dge, oracle):
    rng = np.random.default_rng(3)
    for NV, L, n, zipf in ((3, 5, 64, False), (1, 4, 10, False), (50, 17, 300, True), (400, 64, 120, False), (9, 3, 500, True)):
        ids = (np.minimum(rng.zipf(1.4, size=(n, L)) - 1, NV - 1) if zipf else rng.integers(0, NV, (n, L))).astype(np.int32)
        lens = rng.integers(0, L + 1, n); ids[np.arange(L)[None, :] >= lens[:, None]] = -1
        ids[rng.random(ids.shape) < 0.1] = -1
        kw = dict(negative=int(rng.choice([0, 2, 5, 16, 30])), min_count=int(rng.choice([1, 2])), epochs=1, seed=int(rng.integers(1, 1 << 30)), table_size=997)
        W = int(rng.integers(1, L + 3)); D = int(rng.choice([8, 33, 64, 200]))
        om = oracle.train_sgns(ids, NV, D, W, sorted_chunk=11, sorted_walks=40, **kw)
        with dge.tuning(sorted_chunk=11, sorted_walks=40):
            dm = dge.SgnsModel.fit(ids, dge.make_config(D, W, NV, workers=0, update_policy=8, **kw), 0)
        s0, vid = dm.vectors()
        assert np.array_equal(vid, om.vocab_ids) and dm.stats()["pairs"] == om.pairs, (NV, L, kw)
        assert np.array_equal(bits(s0), bits(om.syn0)) and np.array_equal(bits(dm.syn1neg()), bits(om.syn1neg)), (NV, L, D, W, kw)


def test_owner_computes_row_numbers_of_17_and_18_bits(dge, oracle):
    """Vocabularies of 2^16 .. 2^18 rows are sorted with 9-bit digits (two passes instead of three): same items, same order, same bits."""
    rng = np.random.default_rng(17)
    for NV in (70_000, 140_000, 262_144):
        ids = rng.integers(0, NV, (3000, 12)).astype(np.int32)
        ids[:, 0] = rng.integers(0, 40, 3000)                      # a few busy rows whose item lists span work units
        kw = dict(negative=5, min_count=1, epochs=1, seed=NV, table_size=100_003)
        om = oracle.train_sgns(ids, NV, 8, 4, sorted_chunk=64, sorted_walks=1000, **kw)
        with dge.tuning(sorted_chunk=64, sorted_walks=1000):
            dm = dge.SgnsModel.fit(ids, dge.make_config(8, 4, NV, workers=0, update_policy=8, **kw), 0)
        s0, vid = dm.vectors()
        assert np.array_equal(vid, om.vocab_ids) and dm.stats()["pairs"] == om.pairs, NV
        assert np.array_equal(bits(s0), bits(om.syn0)) and np.array_equal(bits(dm.syn1neg()), bits(om.syn1neg)), NV


def test_owner_computes_generator_run_form_draws_the_tables_rows(dge, oracle):
    """The item generator takes a negative's row from the unigram table's RUN form in LDS where the vocabulary has at most 510 runs of equal counts
    (round 4), from the rank-block table otherwise: the same rows either way — with the run form switched off (DGE_TUNE_TABLE_RUNS = 0) the tables end
    bit-identical, and both equal the oracle's, which reads word2vec's plain table."""
    rng = np.random.default_rng(23)
    for NV, few_counts in ((3000, True), (2500, False)):
        L, n = 10, 4000
        if few_counts:      # every vertex ~ equally often: a few dozen distinct counts
            ids = rng.integers(0, NV, (n, L)).astype(np.int32)
        else:               # a smooth popularity ramp: thousands of distinct counts (more runs than the generator's LDS search takes)
            p = np.linspace(1.0, 300.0, NV); p /= p.sum()
            ids = rng.choice(NV, size=(n * 30, L), p=p).astype(np.int32)
        kw = dict(negative=5, min_count=1, epochs=1, seed=9, table_size=1_000_003)
        om = oracle.train_sgns(ids, NV, 32, 5, sorted_chunk=64, sorted_walks=500, **kw)
        res = []
        for runs_knob in (None, 0):
            with dge.tuning(sorted_chunk=64, sorted_walks=500, **({} if runs_knob is None else {"table_runs": runs_knob})):
                dm = dge.SgnsModel.fit(ids, dge.make_config(32, 5, NV, workers=0, update_policy=8, **kw), 0)
                res.append((dm.vectors()[0], dm.syn1neg(), dm.table_runs(), dm.stats()["pairs"]))
        (a0, a1, runs_a, pa), (b0, b1, runs_b, pb) = res
        assert runs_b[0] == 0 and pa == pb == om.pairs                   # (with the knob at 0 a model has no run form at all)
        assert (0 < runs_a[0] <= 510) if few_counts else runs_a[0] > 510, runs_a
        assert np.array_equal(bits(a0), bits(b0)) and np.array_equal(bits(a1), bits(b1)), NV
        assert np.array_equal(bits(a0), bits(om.syn0)) and np.array_equal(bits(a1), bits(om.syn1neg)), NV


def test_owner_computes_under_the_block_schedule(dge, oracle):
    """N = 3 ranks on one device, every block a device-filling owner-computes launch: bit-identical to the oracle running the 3 x 3
    blocks one after the other with the same schedule."""
    import torch
    walks, NV = _walks(oracle, dge, n=1200)
    kw = dict(negative=5, min_count=2, epochs=1, seed=9, table_size=20011)
    om = oracle.train_sgns(walks, NV, 64, 6, sorted_chunk=32, sorted_walks=150, part_n=3, **kw)
    corpus = dge.WalkCorpus.from_host(walks, 0)
    counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
    with dge.tuning(sorted_chunk=32, sorted_walks=150):
        ms = [dge.SgnsModel.create(dge.make_config(64, 6, NV, workers=0, update_policy=8, **kw), counts, 0) for _ in range(3)]
        simulate_block_schedule(ms, lambda m: m.train(corpus))
        simulate_gather_syn0(ms)
    assert sum(m.stats()["pairs"] for m in ms) == om.pairs
    assert np.array_equal(bits(ms[0].vectors()[0]), bits(om.syn0))
    # syn1neg: partition p is current on rank p after the batch (the ring's invariant)
    s1 = np.stack([ms[r % 3].syn1neg()[r] for r in range(om.V)])
    assert np.array_equal(bits(s1), bits(om.syn1neg))


def test_owner_computes_epoch_long_launch_beyond_2_31_pairs(dge):
    """One launch over 6 M walks of 24 tokens: 2.3e9 pairs — more than a 32-bit count holds (the pair offsets are summed in 64 bits).
    K = 0, D = 8 keeps it short; the atomics schedule trains the same pairs."""
    import torch
    n, L, NV = 6_000_000, 24, 200_000
    g = torch.Generator(device="cuda:0"); g.manual_seed(4)
    walks = torch.randint(0, NV, (n, L), generator=g, device="cuda:0", dtype=torch.int32)
    corpus = dge.WalkCorpus.from_host(walks.cpu().numpy(), 0)
    del walks
    counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
    pairs = {}
    for pol in (8, 2):
        m = dge.SgnsModel.create(dge.make_config(8, L, NV, negative=0, workers=0, update_policy=pol, epochs=1), counts, 0)
        m.train(corpus)
        st = m.stats()
        pairs[pol] = st["pairs"]
        assert np.isfinite(m.vectors()[0]).all() and st["words"] == n * L
        m.close()
    assert pairs[8] == pairs[2] > 2 ** 31
