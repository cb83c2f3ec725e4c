"""CPU: the oracle against every known-answer the reference holds for this path, plus an independent pure-Python
restatement of the parts no reference test covers (RNG streams, pair enumeration, unigram table).

Pinned by the reference: T/LayeredGraphTest.java:12-44 (tests/golden/layered_graph_test.json) and the public
java.util.Random spec (tests/golden/java_random_kats.json).  SGNS: parity unpinned — see oracle/dge_oracle.h.
"""
import json
import math
import os

import numpy as np
import pytest

from helpers import cosine_rows, layered_graph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M64 = (1 << 64) - 1


# ------------------------------------------------------------------------------------------- java.util.Random
def test_java_random_known_answers(oracle):
    for case in json.load(open(os.path.join(GOLD, "java_random_kats.json")))["cases"]:
        r = oracle.JavaRandom(case["seed"])
        if "next_int" in case:
            assert r.next_int() == case["next_int"]
        else:
            assert r.next_double() == case["next_double"]


def test_java_random_jump_equals_stepping(oracle):
    a = oracle.JavaRandom(12345)
    seq = [a.next_double() for _ in range(50)]
    for k in (0, 1, 7, 49):
        b = oracle.JavaRandom(12345)
        b.jump(2 * k)                         # one nextDouble = two LCG steps (next(26), next(27))
        assert b.next_double() == seq[k]
    # spec restated in Python: s' = (s*0x5DEECE66D + 0xB) mod 2^48
    s = (987 ^ 0x5DEECE66D) & ((1 << 48) - 1)
    def nxt(bits):
        nonlocal s
        s = (s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        return s >> (48 - bits)
    r = oracle.JavaRandom(987)
    for _ in range(20):
        assert r.next_double() == ((nxt(26) << 27) + nxt(27)) * 2.0 ** -53


# ------------------------------------------------------------------------------------------- LayeredGraphTest
def test_layered_graph_test_golden_vector(oracle):
    """T/LayeredGraphTest.java:12-44, every assertion."""
    G = json.load(open(os.path.join(GOLD, "layered_graph_test.json")))
    e = np.array(G["edges"])
    g = oracle.Graph()
    g.add_edges(e[:, 0].astype(np.int32), e[:, 1].astype(np.int32), e[:, 2])
    g.set_sources([0])
    g.build_alias(exact=True)
    a = g.get_alias(0)
    assert a["alias"].tolist() == G["alias_table"]
    assert a["prob"].tolist() == G["prob_table"]            # exact doubles, as assertEquals(double,double)
    assert a["out_degree"] == G["out_degree"]
    for x, want in G["draws"]:
        assert g.sample_next(0, x) == want


def test_reference_pairing_order_properties(oracle):
    """The reference pairing (J/LayeredGraph.java:65-81) keeps the sampling distribution: P(edge i) = w_i / sum."""
    rng = np.random.default_rng(0)
    for k in (1, 2, 7, 40):
        w = rng.integers(1, 50, k).astype(float)
        g = oracle.Graph(); g.add_edges(np.zeros(k, np.int32), np.arange(1, k + 1, dtype=np.int32), w); g.set_sources([0])
        for exact in (True, False):
            g.build_alias(exact)
            a = g.get_alias(0)
            p = np.zeros(k)
            for i in range(k):
                p[i] += min(a["prob"][i], 1.0) / k
                if a["alias"][i] >= 0:
                    p[a["alias"][i]] += (1 - a["prob"][i]) / k
                else:
                    p[i] += max(0.0, 1 - a["prob"][i]) / k     # "no alias": stay in the slot
            assert np.allclose(p, w / w.sum(), atol=1e-12)


# ------------------------------------------------------------------------------------------- walks
def test_walk_streams(oracle):
    src, dst, w, sources = layered_graph(R=30, T=5, deg=5, seed=1)
    g = oracle.Graph(); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    a, draws = g.sample_walks(500, 5, seed=7, rng_mode=0, return_draws=True)
    b = g.sample_walks(500, 5, seed=7, rng_mode=1)
    assert draws == 2500 and np.array_equal(a, b)                 # no dead ends: sequential == strided
    assert (a // 30 == np.arange(5)[None, :]).all()              # step j reads slice j (J/CrossTimeGraph.java:36-39)
    # first walk by hand: one draw for the source (J/LayeredGraph.java:234-242), one per step (:104-116)
    r = oracle.JavaRandom(7)
    sa = g.get_source_alias()
    x = r.next_double(); k = len(sa["src"]); i = int(x * k); y = x * k - i
    v = sa["src"][i] if y < sa["prob"][i] else sa["src"][sa["alias"][i]]
    assert v == a[0, 0]
    for j in range(1, 5):
        al = g.get_alias(int(v)); k = len(al["nbr"]); x = r.next_double(); i = int(x * k); y = x * k - i
        v = al["nbr"][i] if y < al["prob"][i] else al["nbr"][al["alias"][i]]
        assert v == a[0, j]


def test_dead_ends_shorten_walks_and_draw_counts(oracle):
    src, dst, w, sources = layered_graph(R=30, T=5, deg=4, seed=3, dead_ends=0.3)
    g = oracle.Graph(); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    a, draws = g.sample_walks(800, 5, seed=9, rng_mode=0, return_draws=True)
    assert (a == -1).any() and draws == int((a >= 0).sum()) < 4000
    lens = (a >= 0).sum(1)
    for row, n in zip(a, lens):                                   # -1 only as right padding
        assert (row[:n] >= 0).all() and (row[n:] == -1).all()
    last = a[np.arange(len(a)), lens - 1]
    short = lens < 5
    assert all(len(g.get_alias(int(v))["nbr"]) == 0 for v in last[short][:50])   # stopped exactly at a sink


def test_transition_frequencies_follow_weights(oracle):
    rng = np.random.default_rng(2)
    k = 12
    w = rng.integers(1, 100, k).astype(float)
    g = oracle.Graph(); g.add_edges(np.zeros(k, np.int32), np.arange(1, k + 1, dtype=np.int32), w); g.set_sources([0]); g.build_alias(True)
    n = 200_000
    walks = g.sample_walks(n, 2, seed=1, rng_mode=1)
    obs = np.bincount(walks[:, 1], minlength=k + 1)[1:]
    e = w / w.sum() * n
    chi2 = float(((obs - e) ** 2 / e).sum())
    assert chi2 < (k - 1) + 5 * math.sqrt(2 * (k - 1))


def test_keep_top_k_is_stable_and_recomputes_degree(oracle):
    g = oracle.Graph()
    w = np.array([0.5, 1.0, 0.5, 0.25, 1.0, 0.5])
    g.add_edges(np.zeros(6, np.int32), np.arange(6, dtype=np.int32), w)
    for v in range(1, 6):
        g.add_edges(np.full(6, v, np.int32), np.arange(6, dtype=np.int32), w)
    g.keep_top_k(4)
    a = g.get_alias(0)
    assert a["nbr"].tolist() == [1, 4, 0, 2]                      # ties keep insertion order (List.sort is stable)
    assert a["out_degree"] == 3.0
    with pytest.raises(RuntimeError):
        g.keep_top_k(5)                                           # subList(0,k) on a shorter list throws


# ------------------------------------------------------------------------------------------- SGNS restatement
def mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def test_vocabulary_table_init_and_pair_enumeration(oracle):
    """Independent Python restatement of everything around the arithmetic: vocabulary order, unigram table
    (word2vec.c InitUnigramTable), InitNet, per-(walk,centre) RNG streams and DL4J's window loop."""
    src, dst, w, sources = layered_graph(R=10, T=4, deg=3, seed=5)
    g = oracle.Graph(); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    walks = g.sample_walks(60, 4, seed=3, rng_mode=1)
    NV, D, W, K, T, seed = 40, 6, 4, 2, 97, 11
    m = oracle.train_sgns(walks, NV, D, W, negative=K, min_count=2, epochs=0, table_size=T, seed=seed)
    cnt = np.bincount(walks[walks >= 0], minlength=NV)
    order = sorted([v for v in range(NV) if cnt[v] >= 2], key=lambda v: (-cnt[v], v))
    assert m.vocab_ids.tolist() == order and m.counts.tolist() == [int(cnt[v]) for v in order]
    # unigram table
    p = np.array([c ** 0.75 for c in m.counts]); p = p / p.sum()
    tab, i, d1 = [], 0, p[0]
    for a in range(T):
        tab.append(i)
        if a / T > d1:
            i += 1; d1 += p[i] if i < len(p) else 0.0
        if i >= len(p):
            i = len(p) - 1
    assert abs(np.array(tab) - m.table(T)).max() <= 1           # python float sum order may differ by one slot at a boundary
    # InitNet
    s = seed
    for r in range(2):
        for b in range(D):
            s = (s * 25214903917 + 11) & M64
            want = np.float32((np.float32(np.float32(s & 0xFFFF) / np.float32(65536)) - np.float32(0.5)) / np.float32(D))
            assert m.syn0[r, b] == want
    assert not m.syn1neg.any()
    # pair count from the window draws (1 epoch)
    m1 = oracle.train_sgns(walks, NV, D, W, negative=K, min_count=2, epochs=1, table_size=T, seed=seed)
    remap = {v: r for r, v in enumerate(order)}
    pairs = 0
    for wi, row in enumerate(walks):
        sen = [remap[t] for t in row if t >= 0 and t in remap]
        for i in range(len(sen)):
            st = mix64((seed + wi * 4 + i) & M64)
            st = (st * 25214903917 + 11) & M64
            b = st % W
            for a in range(b, 2 * W + 1 - b):
                c = i - W + a
                if a != W and 0 <= c < len(sen):
                    pairs += 1
    assert pairs == m1.pairs and m1.total_words == sum(len([t for t in row if t >= 0 and t in remap]) for row in walks)


def test_sigmoid_table_and_lane_order_agreement(oracle):
    e = oracle.exp_table()
    assert e[0] == np.float32(math.exp(-6.0)) / (np.float32(math.exp(-6.0)) + np.float32(1))
    assert abs(e[500] - 0.5) < 1e-6 and (np.diff(e) > 0).all()
    src, dst, w, sources = layered_graph(R=40, T=6, deg=5, seed=0)
    g = oracle.Graph(); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    walks = g.sample_walks(1500, 6, seed=11, rng_mode=1)
    a = oracle.train_sgns(walks, 240, 20, 6, table_size=20011, arith=0)
    b = oracle.train_sgns(walks, 240, 20, 6, table_size=20011, arith=1)
    assert cosine_rows(a.syn0, b.syn0).min() > 1 - 1e-4            # word2vec.c order vs HIP lane order: same vectors
    # training moves positive pairs up and the loss proxy down relative to the initial weights
    init = oracle.train_sgns(walks, 240, 20, 6, table_size=20011, epochs=0)
    assert np.abs(a.syn0 - init.syn0).max() > 1e-3 and a.syn1neg.any()


def test_hogwild_threads_keep_pair_count(oracle):
    src, dst, w, sources = layered_graph(R=40, T=6, deg=5, seed=0)
    g = oracle.Graph(); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
    walks = g.sample_walks(3000, 6, seed=11, rng_mode=1)
    a = oracle.train_sgns(walks, 240, 16, 6, table_size=5003, threads=1)
    b = oracle.train_sgns(walks, 240, 16, 6, table_size=5003, threads=4)
    assert a.pairs == b.pairs and np.isfinite(b.syn0).all()


def test_oracle_regression_fixture(oracle):
    """tests/golden/oracle_small.npz (made by make_golden.py from this oracle): guards against accidental change."""
    G = np.load(os.path.join(GOLD, "oracle_small.npz"))
    g = oracle.Graph(); g.add_edges(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(True)
    assert np.array_equal(g.sample_walks(64, 4, seed=2017, rng_mode=0), G["walks_seq"])
    assert np.array_equal(g.sample_walks(64, 4, seed=2017, rng_mode=1, first_index=5), G["walks_str"])
    a3 = g.get_alias(3)
    assert np.array_equal(a3["alias"], G["alias3"]) and np.array_equal(a3["prob"], G["prob3"])
    for arith in (0, 1):
        m = oracle.train_sgns(G["walks_seq"], 48, 8, 4, negative=3, min_count=2, table_size=257, arith=arith)
        assert np.array_equal(m.vocab_ids, G["vocab"]) and np.array_equal(m.counts, G["counts"])
        assert np.array_equal(m.table(257), G["table"]) and m.pairs == int(G["pairs"][0])
        assert np.array_equal(m.syn0.view(np.int32), G["syn0_arith%d" % arith].view(np.int32))
        assert np.array_equal(m.syn1neg.view(np.int32), G["syn1_arith%d" % arith].view(np.int32))


def test_vec_format_fixture():
    """miscs/taxi_all.txt (LINE-style .vec: header 'V D', then 'id v1..vD'): the reader contract of
    P/embeddingEvaluation_tract.py:113-117 — skip_header=1, first column is the region id."""
    rows = np.genfromtxt(os.path.join(GOLD, "taxi_all_head.vec"), skip_header=1)
    head = open(os.path.join(GOLD, "taxi_all_head.vec")).readline().split()
    assert head == ["77", "8"] and rows.shape == (3, 9) and rows[:, 0].tolist() == [1.0, 2.0, 3.0]
    assert np.allclose(np.linalg.norm(rows[:, 1:5], axis=1), 1.0, atol=1e-4)      # two unit 4-vectors per row


def test_file_formats_roundtrip(tmp_path, oracle):
    """.od -> layered graph -> .seq -> ids, and .vec read-back (formats: embedding_amd/io.py header)."""
    from embedding_amd import io
    rng = np.random.default_rng(0)
    paths = []
    for h in range(3):
        p = tmp_path / ("taxi-h%d.od" % h)
        with open(p, "w") as f:
            for s in (17031, 17045, 17099):
                for d in (17031, 17045, 17099):
                    f.write("%d %d %d\n" % (s, d, int(rng.integers(0, 4))))       # zeros are dropped
        paths.append(str(p))
    G = io.read_od_slices(paths)
    assert G["R"] == 3 and G["T"] == 3 and (G["w"] > 0).all() and (G["src"] // 3 == (G["dst"] // 3 - 1) % 3).all()
    g = oracle.Graph(); g.add_edges(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(True)
    walks = g.sample_walks(50, 3, seed=1, rng_mode=0)
    seq = tmp_path / "taxi-crosstime.seq"
    io.write_seq(str(seq), walks, G["names"])
    first = open(seq).readline().split()
    assert first[0].startswith("0-170") and all(t.count("-") == 1 for t in first)
    back, names = io.read_seq(str(seq))
    assert back.shape[0] == 50 and [names[t] for t in back[0] if t >= 0] == first
    vec = tmp_path / "x.vec"
    with open(vec, "w") as f:
        f.write("0-17031 0.5 -1.25\n1-17045 1e-3 2\n")
    n, v = io.read_vec(str(vec))
    assert n == ["0-17031", "1-17045"] and v.shape == (2, 2) and v[1, 0] == np.float32(1e-3)
    n, v = io.read_vec(os.path.join(GOLD, "taxi_all_head.vec"), header=True)
    assert n == ["1", "2", "3"] and v.shape == (3, 8)


def test_od_sources_include_a_destination_only_layer0_vertex(tmp_path, oracle):
    """J/CrossTimeGraph.java:43-47 adds EVERY "0-<id>" that exists in allVertices as a source — also a region that slice 0 never
    leaves but slice T-1 arrives at (outDegree 0: it sits in the source alias table with weight 0 and is never drawn).  The source
    table's length k enters every draw (i = (int)(x*k)), so leaving such a vertex out would change all seeded walks."""
    from embedding_amd import io
    rows = {0: [(10, 11, 3), (11, 10, 2)], 1: [(10, 11, 1), (11, 12, 4)]}      # region 12: only a destination of the LAST slice
    paths = []
    for h in (0, 1):
        p = tmp_path / ("taxi-h%d.od" % h)
        with open(p, "w") as f:
            for a, b, w in rows[h]:
                f.write("%d %d %d\n" % (a, b, w))
        paths.append(str(p))
    G = io.read_od_slices(paths)
    assert G["regions"].tolist() == [10, 11, 12] and G["names"][2] == "0-12"
    assert G["sources"].tolist() == [0, 1, 2]                         # "0-12" is a source although nothing leaves it
    g = oracle.Graph(); g.add_edges(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(True)
    sa = g.get_source_alias()
    assert len(sa["prob"]) == 3 and sa["prob"][2] == 0.0 and sa["weight_sum"] == 5.0
    walks = g.sample_walks(500, 2, seed=3, rng_mode=0)
    assert (walks[:, 0] != 2).all() and set(walks[:, 0].tolist()) == {0, 1}


def test_quality_metric_restatement(oracle):
    """oracle/quality.py (P/embeddingEvaluation_tract.py:169-196,249-260): KNN by cosine distance + nDCG@k, the host restatement."""
    from oracle import quality as ev
    rng = np.random.default_rng(0)
    f = rng.normal(size=(12, 5)); f[3] = 0.0                        # a zero vector: cosine is NaN -> distance 2
    rids = [100 + i for i in range(12)]
    est, nb = ev.pairwise_estimator(f, rids)
    assert all(len(nb[r]) == 11 and r not in nb[r] for r in rids)
    assert all(np.all(np.diff([d for _, d in est[r]]) >= 0) for r in rids)      # ascending distance
    assert est[100][-1] == (103, 2.0) and all(d == 2.0 for _, d in est[103])
    i, j = 0, rids.index(nb[100][0])
    assert abs(est[100][0][1] - (1 - f[i] @ f[j] / np.linalg.norm(f[i]) / np.linalg.norm(f[j]))) < 1e-12
    assert abs(ev.ndcg_against(f, f, rids, k=5) - 1.0) < 1e-12        # a feature set against itself
    g = f + 0.01 * rng.normal(size=f.shape); g[3] = 0.0
    assert 0.9 < ev.ndcg_against(g, f, rids, k=5) <= 1.0 + 1e-12
    assert ev.ndcg_against(rng.normal(size=f.shape), f, rids, k=5) < ev.ndcg_against(g, f, rids, k=5)
    # statistical parity of two training schedules of the oracle on one slice of a small layered graph
    src, dst, w, sources = layered_graph(R=40, T=4, deg=6, seed=1)
    gph = oracle.Graph(); gph.add_edges(src, dst, w); gph.set_sources(sources); gph.build_alias(True)
    walks = gph.sample_walks(4000, 4, seed=3, rng_mode=1)
    a = oracle.train_sgns(walks, 160, 16, 4, table_size=5003, threads=1)
    b = oracle.train_sgns(walks, 160, 16, 4, table_size=5003, threads=4)
    sl = [r for r, v in enumerate(a.vocab_ids) if 40 <= v < 80]      # the rows of slice 1
    assert ev.ndcg_against(b.syn0[sl], a.syn0[sl], [int(a.vocab_ids[r]) for r in sl], k=5) > 0.8


def test_update_arithmetic_second_reading_in_numpy_float32(oracle):
    """A second, independent reading of the SGNS / hierarchical-softmax UPDATE (w2v.fit(), J/DeepWalk.java:73-79; word2vec.c's skip-gram loop
    with DL4J's pair enumeration): every pair of a tiny corpus trained here in numpy float32 — unfused multiply-add, sequential dot product,
    sigmoid through the 1000-entry table, hierarchical-softmax terms ahead of the negatives, a negative equal to the centre skipped — must
    leave syn0, syn1neg and syn1 BIT-IDENTICAL to the C oracle's word2vec-order run (arith=0).  The oracle's update thus has two readings
    that agree; neither is the reference (DL4J's source is absent: parity unpinned)."""
    f32 = np.float32
    rng = np.random.default_rng(12)
    NV, L, n, D, W, K, T, seed = 9, 5, 14, 7, 3, 3, 53, 77
    walks = rng.integers(0, NV, (n, L)).astype(np.int32)
    walks[rng.random(walks.shape) < 0.15] = -1
    for use_hs in (False, True):
        m0 = oracle.train_sgns(walks, NV, D, W, negative=K, min_count=2, epochs=0, table_size=T, seed=seed, use_hs=use_hs)
        m1 = oracle.train_sgns(walks, NV, D, W, negative=K, min_count=2, epochs=1, table_size=T, seed=seed, use_hs=use_hs, arith=0)
        # (m1: the form the suite runs — a pair's dot products side by side, train_walk_ilp; m1p: the plain word2vec.c-shaped loop, train_walk)
        oracle.set_plain(True)
        try:
            m1p = oracle.train_sgns(walks, NV, D, W, negative=K, min_count=2, epochs=1, table_size=T, seed=seed, use_hs=use_hs, arith=0)
        finally:
            oracle.set_plain(False)
        assert m1p.pairs == m1.pairs and np.array_equal(m1p.syn0.view(np.int32), m1.syn0.view(np.int32)) and np.array_equal(m1p.syn1neg.view(np.int32), m1.syn1neg.view(np.int32))
        V = m0.V
        remap = {int(v): r for r, v in enumerate(m0.vocab_ids)}
        table = m0.table(T)
        syn0 = m0.syn0.astype(f32).copy(); syn1neg = np.zeros((V, D), f32); syn1 = np.zeros((max(V - 1, 1), D), f32)
        paths = [m0.code(r) for r in range(V)] if use_hs else None
        # word2vec.c expTable: exp() in double of a float argument, stored as float; x / (x + 1) in float
        e = np.array([f32(math.exp(float(f32(f32(f32(i) / f32(1000) * f32(2) - f32(1)) * f32(6))))) for i in range(1000)], f32)
        e = (e / (e + f32(1))).astype(f32)
        assert np.array_equal(e, oracle.exp_table())

        def dot(a, b):
            f = f32(0)
            for k in range(D):
                f = f32(f + f32(a[k] * b[k]))
            return f

        def axpy(y, g, x):                                   # y[k] += g * x[k], unfused
            for k in range(D):
                y[k] = f32(y[k] + f32(g * x[k]))

        sens = [[remap[int(t)] for t in row if t >= 0 and int(t) in remap] for row in walks]
        total_words = sum(len(s) for s in sens)
        done, pairs = 0, 0
        for wi, sen in enumerate(sens):
            alpha = f32(float(f32(0.025)) * (1.0 - done / (total_words + 1)))      # (the configured rate is a float; the schedule runs in double)
            if alpha < f32(1e-4):
                alpha = f32(1e-4)
            for i, word in enumerate(sen):
                s = mix64((seed + wi * L + i) & M64)
                s = (s * 25214903917 + 11) & M64
                b = s % W
                for a in range(b, 2 * W + 1 - b):
                    c = i - W + a
                    if a == W or c < 0 or c >= len(sen):
                        continue
                    l1 = syn0[sen[c]]
                    neu = np.zeros(D, f32)
                    if use_hs:
                        pts, cds = paths[word]
                        for node, code in zip(pts, cds):
                            f = dot(l1, syn1[node])
                            if f <= f32(-6) or f >= f32(6):
                                continue
                            g = f32(f32(f32(1) - f32(code) - e[int(f32(f32(f + f32(6)) * f32(1000 // 6 // 2)))]) * alpha)
                            axpy(neu, g, syn1[node]); axpy(syn1[node], g, l1)
                    for d in range(K + 1):
                        if d == 0:
                            target, label = word, f32(1)
                        else:
                            s = (s * 25214903917 + 11) & M64
                            target = int(table[(s >> 16) % T])
                            if target == 0 and V > 1:
                                target = s % (V - 1) + 1
                            if target == word:
                                continue
                            label = f32(0)
                        f = dot(l1, syn1neg[target])
                        if f > f32(6):
                            g = f32(f32(label - f32(1)) * alpha)
                        elif f < f32(-6):
                            g = f32(label * alpha)
                        else:
                            g = f32(f32(label - e[int(f32(f32(f + f32(6)) * f32(1000 // 6 // 2)))]) * alpha)
                        axpy(neu, g, syn1neg[target]); axpy(syn1neg[target], g, l1)
                    for k in range(D):
                        l1[k] = f32(l1[k] + neu[k])
                    pairs += 1
            done += len(sen)
        assert pairs == m1.pairs and pairs > 50
        assert np.array_equal(syn0.view(np.int32), m1.syn0.view(np.int32)), use_hs
        assert np.array_equal(syn1neg.view(np.int32), m1.syn1neg.view(np.int32)), use_hs
        if use_hs and V > 1:
            assert np.array_equal(syn1[:V - 1].view(np.int32), m1.syn1.view(np.int32))


@pytest.mark.parametrize("seed", range(48))
def test_side_by_side_dot_products_leave_the_plain_loops_tables(oracle, seed):
    """oracle/dge_oracle.c has the pair loop twice: train_walk, shaped like word2vec.c's (the definition), and train_walk_ilp, which takes the dot products of
    a pair's pairwise distinct rows side by side and then applies the terms in the same order (what the suite and bench.py's CPU baseline run: ~3x the
    speed).  Same floating-point operations on the same values: the tables must be BIT-IDENTICAL — random widths (tails of both lane orders), 0 .. 30 negatives,
    vocabularies from 3 rows (every pair repeats a row: the term-by-term branch) to thousands, ragged walks, the tree term, blocks of 2 / 3 ranks, both
    arithmetic orders, two epochs."""
    rng = np.random.default_rng(1000 + seed)
    NV = int(rng.choice([3, 5, 12, 60, 400, 3000]))
    L = int(rng.integers(2, 25)); n = int(rng.integers(20, 300))
    D = int(rng.choice([1, 2, 7, 16, 20, 31, 64, 100, 128, 130, 200, 256]))
    K = int(rng.choice([0, 1, 2, 5, 5, 7, 8, 9, 20, 30])); W = int(rng.integers(1, L + 1))
    walks = rng.integers(0, NV, (n, L)).astype(np.int32)
    if seed % 3 == 0:                                                     # a skewed vocabulary: the same rows drawn again and again
        walks = np.minimum(walks, rng.integers(0, NV, (n, L))).astype(np.int32)
    walks[rng.random(walks.shape) < 0.1] = -1
    kw = dict(negative=K, min_count=int(rng.integers(1, 3)), epochs=int(rng.integers(1, 3)), table_size=int(rng.choice([53, 1009, 100_003])), seed=int(rng.integers(1, 1 << 30)),
              arith=seed & 1, use_hs=bool((seed >> 1) & 1), part_n=int(rng.choice([0, 0, 2, 3])))
    fast = oracle.train_sgns(walks, NV, D, W, **kw)
    oracle.set_plain(True)
    try:
        plain = oracle.train_sgns(walks, NV, D, W, **kw)
    finally:
        oracle.set_plain(False)
    assert fast.pairs == plain.pairs and np.array_equal(fast.vocab_ids, plain.vocab_ids), (seed, kw)
    assert np.array_equal(fast.syn0.view(np.int32), plain.syn0.view(np.int32)), (seed, NV, D, K, kw)
    assert np.array_equal(fast.syn1neg.view(np.int32), plain.syn1neg.view(np.int32)), (seed, NV, D, K, kw)
    if kw["use_hs"] and fast.V > 1:
        assert np.array_equal(fast.syn1.view(np.int32), plain.syn1.view(np.int32)), (seed, NV, D, K, kw)


def test_oracle_continues_from_a_given_state(oracle):
    """orc_train_sgns_from (the warm start of tests/test_gpu_quality.py): with the whole corpus' counts, the learning-rate position and the tables a
    run has reached, training the second half of a corpus continues the run bit for bit — sequentially, and under the hierarchical softmax's
    negatives-only tables as well."""
    rng = np.random.default_rng(0)
    w = rng.integers(0, 50, (200, 6)).astype(np.int32)
    w[rng.random(w.shape) < 0.05] = -1
    for arith in (0, 1):
        kw = dict(negative=5, table_size=1009, arith=arith, seed=3)
        whole = oracle.train_sgns(w, 50, 16, 6, **kw)
        cnt = np.bincount(w[w >= 0], minlength=50).astype(np.int64)
        h1 = oracle.train_sgns(w[:100], 50, 16, 6, counts=cnt, total_walks=200, total_words=whole.total_words, **kw)
        assert np.array_equal(h1.vocab_ids, whole.vocab_ids) and np.array_equal(h1.table(1009), whole.table(1009))     # the corpus' vocabulary, not the half's
        words_before = int(np.isin(w[:100], whole.vocab_ids).sum())
        h2 = oracle.train_sgns(w[100:], 50, 16, 6, counts=cnt, total_walks=200, total_words=whole.total_words, walk_index_base=100,
                               words_before=words_before, syn0_init=h1.syn0, syn1neg_init=h1.syn1neg, **kw)
        assert h1.pairs + h2.pairs == whole.pairs
        assert np.array_equal(h2.syn0.view(np.int32), whole.syn0.view(np.int32)) and np.array_equal(h2.syn1neg.view(np.int32), whole.syn1neg.view(np.int32))
    # ... and with the hierarchical softmax: the inner-node table is handed over as well (orc_train_sgns_from_hs)
    kw = dict(negative=3, table_size=1009, arith=0, seed=3, use_hs=True)
    whole = oracle.train_sgns(w, 50, 16, 6, **kw)
    cnt = np.bincount(w[w >= 0], minlength=50).astype(np.int64)
    h1 = oracle.train_sgns(w[:100], 50, 16, 6, counts=cnt, total_walks=200, total_words=whole.total_words, **kw)
    words_before = int(np.isin(w[:100], whole.vocab_ids).sum())
    h2 = oracle.train_sgns(w[100:], 50, 16, 6, counts=cnt, total_walks=200, total_words=whole.total_words, walk_index_base=100,
                           words_before=words_before, syn0_init=h1.syn0, syn1neg_init=h1.syn1neg, syn1_init=h1.syn1, **kw)
    assert h1.pairs + h2.pairs == whole.pairs and np.abs(whole.syn1).max() > 0
    for a, b in ((h2.syn0, whole.syn0), (h2.syn1neg, whole.syn1neg), (h2.syn1, whole.syn1)):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
    with pytest.raises(ValueError):
        oracle.train_sgns(w, 50, 16, 6, counts=np.zeros(7, np.int64))
