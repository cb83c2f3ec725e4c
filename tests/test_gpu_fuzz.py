"""GPU: seeded random sweep of the trainer's parameter space against the oracle — in-order workers, bit-exact.
Each case draws corpus shape (ragged walks, padding, out-of-vocabulary tokens), dimension, window, negatives, min_count,
epochs, table size, and the mode (plain / hierarchical softmax / multi-GPU block schedule).  Reference call site:
J/DeepWalk.java:73-79 (the SGNS half of the oracle is a restatement: parity unpinned)."""
import numpy as np
import pytest

from helpers import bits, simulate_block_schedule, simulate_gather_syn0

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    NV = int(rng.integers(8, 400))
    L = int(rng.choice([2, 3, 5, 8, 16, 17, 24, 40, 64]))
    n = int(rng.integers(20, 300))
    zipf = rng.random() < 0.5
    ids = (np.minimum(rng.zipf(1.4, size=(n, L)) - 1, NV - 1) if zipf else rng.integers(0, NV, (n, L))).astype(np.int32)
    lens = rng.integers(0, L + 1, n)                       # ragged: the tail of every walk is padding (-1)
    ids[np.arange(L)[None, :] >= lens[:, None]] = -1
    if rng.random() < 0.3:
        ids[rng.random(ids.shape) < 0.1] = -1              # holes inside walks too (dropped like out-of-vocabulary tokens)
    cfg = dict(dim=int(rng.choice([2, 8, 20, 33, 64, 65, 100, 128, 192, 200, 256, 384, 500])), window=int(rng.integers(1, L + 3)),
               negative=int(rng.choice([0, 1, 2, 5, 13, 16, 17, 30])), min_count=int(rng.choice([1, 1, 2, 3])),
               epochs=int(rng.choice([1, 1, 2])), table_size=int(rng.choice([64, 997, 20011])), seed=int(rng.integers(1, 1 << 30)))
    mode = str(rng.choice(["plain", "plain", "hs", "blocks"]))
    return ids, NV, cfg, mode


@pytest.mark.parametrize("seed", list(range(40)) + [-s for s in range(1, 17)])
def test_random_configuration_bit_exact(dge, oracle, seed, request):
    import torch
    if seed < 0:                      # the same cases again through the per-segment-descriptor addressing of >= 4 GiB tables
        knobs = dge.tuning(force_segments=1, segment_shift=3)     # 8 rows per descriptor segment: rows of one pair spread over many segments
        knobs.__enter__(); request.addfinalizer(lambda: knobs.__exit__(None, None, None))
        seed = -seed
    ids, NV, cfg, mode = _case(seed)
    if mode == "hs" and cfg["negative"] == 0 and cfg["dim"] > 256:
        cfg["dim"] = 64
    n_ranks = 3 if mode == "blocks" else 0
    om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                           seed=cfg["seed"], table_size=cfg["table_size"], arith=1, use_hs=(mode == "hs"), part_n=n_ranks)
    c = dge.make_config(cfg["dim"], cfg["window"], NV, negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                        workers=1, seed=cfg["seed"], table_size=cfg["table_size"], use_hs=(mode == "hs"))
    if mode != "blocks" or om.V < n_ranks:
        if mode == "blocks":
            om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                                   seed=cfg["seed"], table_size=cfg["table_size"], arith=1)
        dm = dge.SgnsModel.fit(ids, c, 0)
        pairs = dm.stats()["pairs"]
    else:
        corpus = dge.WalkCorpus.from_host(ids, 0)
        counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
        ms = [dge.SgnsModel.create(c, counts, 0) for _ in range(n_ranks)]
        for ep in range(cfg["epochs"]):
            simulate_block_schedule(ms, lambda m: m.train(corpus, epoch=ep))
        simulate_gather_syn0(ms)
        dm = ms[0]
        pairs = sum(m.stats()["pairs"] for m in ms)
    syn0, vid = dm.vectors()
    assert np.array_equal(vid, om.vocab_ids), (seed, mode, cfg)
    assert pairs == om.pairs, (seed, mode, cfg)
    assert np.array_equal(bits(syn0), bits(om.syn0)), (seed, mode, cfg)
    assert np.array_equal(bits(dm.syn1neg()), bits(om.syn1neg)), (seed, mode, cfg)
    if mode == "hs" and om.V > 1:
        assert np.array_equal(bits(dm.syn1()), bits(om.syn1)), (seed, mode, cfg)


def _graph_case(seed):
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.integers(3, 300))
    E = int(rng.integers(V, 12 * V))
    hubby = rng.random() < 0.4
    src = (np.minimum(rng.zipf(1.3, E) - 1, V - 1) if hubby else rng.integers(0, V, E)).astype(np.int32)      # hubs: hundreds of out-edges
    dst = rng.integers(0, V, E).astype(np.int32)                                                             # self loops and duplicates included
    wkind = rng.integers(0, 3)
    w = (rng.integers(1, 60, E).astype(np.float64) if wkind == 0 else rng.random(E) + 1e-3 if wkind == 1 else np.full(E, 2.0))
    top_k = int(rng.integers(1, 4)) if rng.random() < 0.3 else None
    if top_k is not None:                                       # the prune needs every vertex to keep >= k out-edges
        extra = np.repeat(np.arange(V, dtype=np.int32), top_k)
        src = np.concatenate([src, extra]); dst = np.concatenate([dst, rng.integers(0, V, len(extra)).astype(np.int32)])
        w = np.concatenate([w, rng.integers(1, 60, len(extra)).astype(np.float64)])
        p = rng.permutation(len(src)); src, dst, w = src[p], dst[p], w[p]
    elif rng.random() < 0.5:                                    # dead ends: some vertices lose all their out-edges
        dead = rng.random(V) < 0.15
        keep = ~dead[src]
        if keep.sum() >= 2:
            src, dst, w = src[keep], dst[keep], w[keep]
    n_src = int(rng.integers(1, V + 1))
    present = np.unique(np.concatenate([src, dst]))
    sources = rng.choice(present, size=min(n_src, len(present)), replace=False).astype(np.int32)
    return src, dst, w, sources, dict(exact=bool(rng.random() < 0.5), stream_sum=bool(rng.random() < 0.3), L=int(rng.choice([1, 2, 5, 8, 24, 33])),
                                     n=int(rng.integers(1, 3000)), seed=int(rng.integers(-2**40, 2**40)), first=int(rng.integers(0, 10**6)),
                                     top_k=top_k)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_graph_alias_and_walks_bit_exact(dge, oracle, seed):
    """Edge store -> (top-k prune) -> alias tables -> walks in both RNG layouts, on random multigraphs with hubs, self loops,
    duplicate edges, dead ends, arbitrary source sets (J/LayeredGraph.java:54-82,104-116,195-252; J/SpatialGraph.java:29-35)."""
    from helpers import build_both
    src, dst, w, sources, c = _graph_case(seed)
    if c["top_k"] is not None:                                  # the prune needs EVERY vertex to have at least k out-edges (the
        deg = np.bincount(src, minlength=int(max(src.max(), dst.max())) + 1)        # reference's subList(0, k) throws otherwise)
        if (deg < c["top_k"]).any():
            c["top_k"] = None
    og, dg = build_both(oracle, dge, src, dst, w, sources, exact=c["exact"], stream_sum=c["stream_sum"], top_k=c["top_k"])
    assert dg.num_vertices == og.num_vertices and dg.num_edges == og.num_edges
    for v in np.unique(np.concatenate([src[:20], dst[:5], sources[:5]])):
        a, b = og.get_alias(int(v)), dg.get_alias(int(v))
        assert np.array_equal(a["nbr"], b["nbr"]) and np.array_equal(a["alias"], b["alias"]), (seed, v)
        assert np.array_equal(bits(a["prob"]), bits(b["prob"])) and a["out_degree"] == b["out_degree"], (seed, v)
    sa, sb = og.get_source_alias(), dg.get_source_alias()
    assert np.array_equal(sa["alias"], sb["alias"]) and np.array_equal(bits(sa["prob"]), bits(sb["prob"])) and sa["weight_sum"] == sb["weight_sum"]
    for mode in (1, 0):
        first = c["first"] if mode == 1 else 0
        wo, do = og.sample_walks(c["n"], c["L"], seed=c["seed"], rng_mode=mode, first_index=first, return_draws=True)
        wd, dd = dg.sample_walks(c["n"], c["L"], seed=c["seed"], rng_mode=mode, first_index=first, return_draws=True)
        assert np.array_equal(wo, wd), (seed, mode, c)
        if mode == 0:                                           # the java stream continues where this call stopped
            assert do == dd, (seed, c)
        else:                                                   # strided: the call owns n*L draws of the stream
            assert dd == c["n"] * c["L"], (seed, c)


@pytest.mark.parametrize("seed", list(range(100, 124)))
def test_random_configuration_lock_kernels_one_worker(dge, oracle, seed):
    """The same random configurations through the commit-lock kernel (policy 5), its strict form (6) and the mixed policy (7, with a
    random head size), ONE worker each: with nobody else on the tables the lock kernels run the sequential word2vec schedule — positive
    target first, a parked centre delta flushed before its row is read again — so they agree with the oracle to rounding (not bit for bit:
    the centre's delta is summed in LDS and added once, the head rows of policy 7 are updated by float atomics): north_star's 1e-4 cosine
    on EVERY row of both tables, with the exact pair count.  (SGNS half of the oracle: a restatement, parity unpinned — DESIGN.md §3.)"""
    ids, NV, cfg, _ = _case(seed)
    # (three one-worker launches per case: the few cases with long walks, wide windows and 30 negatives are cut to ~1e7 row updates — they took 10 - 23 s each uncut,
    #  6 - 12 s at 1.5e7; the bound on every row does not depend on the corpus' length)
    est = ids.shape[0] * ids.shape[1] * min(2 * cfg["window"], ids.shape[1]) * (cfg["negative"] + 1) * cfg["epochs"]
    if est > 1.0e7:
        ids = ids[:max(20, int(ids.shape[0] * 1.0e7 / est))]
    om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                           seed=cfg["seed"], table_size=cfg["table_size"], arith=0)
    rng = np.random.default_rng(seed)
    for pol in (5, 6, 7):
        c = dge.make_config(cfg["dim"], cfg["window"], NV, negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                            workers=1, seed=cfg["seed"], table_size=cfg["table_size"], update_policy=pol)
        with dge.tuning(**({"hot_rows": int(rng.integers(0, max(om.V, 1) + 1))} if pol == 7 else {})):
            dm = dge.SgnsModel.fit(ids, c, 0)
        syn0, vid = dm.vectors()
        assert np.array_equal(vid, om.vocab_ids) and dm.stats()["pairs"] == om.pairs, (seed, pol, cfg)
        if om.V:
            from helpers import cosine_rows
            assert np.isfinite(syn0).all()
            c0 = cosine_rows(syn0, om.syn0); c1 = cosine_rows(dm.syn1neg() + 1e-30, om.syn1neg + 1e-30)
            assert c0.min() > 1 - 1e-4 and c1.min() > 1 - 1e-4, ("1 - cosine: syn0 %.3g, syn1neg %.3g" % (1 - c0.min(), 1 - c1.min()), seed, pol, cfg)


@pytest.mark.parametrize("seed", list(range(40)) + [-s for s in range(1, 17)])
def test_random_configuration_against_word2vec_order(dge, oracle, seed, request):
    """The 40 + 16 configurations of test_random_configuration_bit_exact once more, against the oracle in word2vec.c's OWN arithmetic order
    (arith=0: sequential dot products, unfused multiply-add) instead of the order written to mirror the kernels' lanes — for the in-order
    schedule (policy 0), memory-side atomics with one worker (2) and the owner-computes policy with one worker (8), plain, with hierarchical
    softmax and under the 3-rank block schedule: north_star's 1e-4 cosine on EVERY row of syn0, syn1neg and syn1, with the exact pair count.
    So no schedule's only witness is an oracle mode shaped after the kernel."""
    import torch
    from helpers import cosine_rows
    if seed < 0:
        knobs = dge.tuning(force_segments=1, segment_shift=3)
        knobs.__enter__(); request.addfinalizer(lambda: knobs.__exit__(None, None, None))
        seed = -seed
    ids, NV, cfg, mode = _case(seed)
    if mode == "hs" and cfg["negative"] == 0 and cfg["dim"] > 256:
        cfg["dim"] = 64
    n_ranks = 3 if mode == "blocks" else 0
    kw = dict(negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"], seed=cfg["seed"], table_size=cfg["table_size"])
    om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], arith=0, use_hs=(mode == "hs"), part_n=n_ranks, **kw)
    if mode == "blocks" and om.V < n_ranks:
        mode = "plain"
        om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], arith=0, **kw)
    for pol in ((0, 2) if mode == "hs" else (0, 2, 8)):          # (hierarchical softmax runs under policies 0 / 2 only)
        c = dge.make_config(cfg["dim"], cfg["window"], NV, workers=1, update_policy=pol, use_hs=(mode == "hs"), **kw)
        if mode != "blocks":
            dm = dge.SgnsModel.fit(ids, c, 0)
            pairs = dm.stats()["pairs"]
            s1 = dm.syn1neg()
        else:
            corpus = dge.WalkCorpus.from_host(ids, 0)
            counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
            ms = [dge.SgnsModel.create(c, counts, 0) for _ in range(n_ranks)]
            for ep in range(cfg["epochs"]):
                simulate_block_schedule(ms, lambda m: m.train(corpus, epoch=ep))
            simulate_gather_syn0(ms)
            dm = ms[0]
            pairs = sum(m.stats()["pairs"] for m in ms)
            s1 = np.stack([ms[r % n_ranks].syn1neg()[r] for r in range(om.V)]) if om.V else dm.syn1neg()      # partition p is current on rank p
        syn0, vid = dm.vectors()
        assert np.array_equal(vid, om.vocab_ids) and pairs == om.pairs, (seed, mode, pol, cfg)
        if om.V == 0:
            continue
        assert np.isfinite(syn0).all()
        c0 = cosine_rows(syn0, om.syn0).min(); c1 = cosine_rows(s1 + 1e-30, om.syn1neg + 1e-30).min()
        assert c0 > 1 - 1e-4 and c1 > 1 - 1e-4, ("1 - cosine: syn0 %.3g, syn1neg %.3g" % (1 - c0, 1 - c1), seed, mode, pol, cfg)
        if mode == "hs" and om.V > 1:
            c2 = cosine_rows(dm.syn1() + 1e-30, om.syn1 + 1e-30).min()
            assert c2 > 1 - 1e-4, ("1 - cosine: syn1 %.3g" % (1 - c2), seed, pol, cfg)
