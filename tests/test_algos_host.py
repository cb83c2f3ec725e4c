"""CPU: the per-lane scalar logic of the HIP kernels (embedding_amd/csrc/dge_algos.h, host build in tests/native)
against the oracle — the bit-set restatement of the reference's O(k^2) alias pairing must give bit-identical arrays."""
import ctypes as C

import numpy as np

from helpers import bits


def _tables(oracle, w):
    k = len(w)
    g = oracle.Graph(); g.add_edges(np.zeros(k, np.int32), np.arange(1, k + 1, dtype=np.int32), w); g.set_sources([0])
    return g


def test_bitset_structure(algos_harness):
    for k in (1, 5, 64, 65, 4096, 4097, 300_000):
        assert algos_harness.harness_bitset_selftest(k, 123, 20_000) == 0


def test_alias_pairings_bit_identical_to_oracle(algos_harness, oracle):
    rng = np.random.default_rng(1)
    for trial in range(600):
        k = int(rng.integers(1, 60)) if trial < 500 else int(rng.integers(100, 3000))
        mode = trial % 5
        if mode == 0: w = rng.integers(1, 10, k).astype(float)
        elif mode == 1: w = rng.exponential(20, k)
        elif mode == 2: w = np.floor(rng.pareto(1.2, k) * 3) + 1
        elif mode == 3:
            w = np.ones(k); w[rng.integers(0, k)] = k * 2
        else: w = rng.integers(1, 4, k).astype(float)
        g = _tables(oracle, w)
        for exact, fn in ((True, algos_harness.harness_alias_reference), (False, algos_harness.harness_alias_vose)):
            g.build_alias(exact)
            a = g.get_alias(0)
            prob = np.zeros(k); alias = np.zeros(k, np.int32)
            fn(w.ctypes.data, k, a["out_degree"], prob.ctypes.data, alias.ctypes.data)
            assert np.array_equal(bits(prob), bits(a["prob"])) and np.array_equal(alias, a["alias"]), (trial, exact, k)


def test_java_lcg_jump_and_mixers(algos_harness, oracle):
    for seed in (0, 42, -7, 2**40 + 3):
        for n in (0, 1, 2, 1000, 2**33 + 5):
            r = oracle.JavaRandom(seed); r.jump(n)
            assert algos_harness.harness_jr_jump(seed, n) == r.state
    s = C.c_uint64(algos_harness.harness_jr_jump(42, 0))
    assert algos_harness.harness_jr_next_double(C.byref(s)) == 0.7275636800328681
    for x in (0, 1, 2**63, 123456789):
        assert algos_harness.harness_mix64(x) == oracle.mix64(x)
    st = 99
    for _ in range(37):
        st = (st * 25214903917 + 11) & (2**64 - 1)
    assert algos_harness.harness_w2v_jump(99, 37) == st


def test_stream_sum_matches_oracle_top_k_degree(algos_harness, oracle):
    rng = np.random.default_rng(3)
    w = np.exp(-rng.random(40) * 5)
    g = oracle.Graph(); g.add_edges(np.zeros(40, np.int32), np.arange(40, dtype=np.int32), w)
    for v in range(1, 40):
        g.add_edges(np.full(40, v, np.int32), np.arange(40, dtype=np.int32), w)
    g.keep_top_k(10)
    top = np.sort(w)[::-1][:10].copy()
    assert algos_harness.harness_stream_sum(top.ctypes.data, 10) == g.get_alias(0)["out_degree"]


def test_huffman_paths_match_word2vec_tree(algos_harness, oracle):
    """dge_huffman_paths (product, CSR form) == CreateBinaryTree as restated in the oracle, ties included."""
    rng = np.random.default_rng(5)
    cases = [np.array([7], np.int64), np.array([5, 5], np.int64), np.array([9, 4, 4, 4, 1, 1, 1, 1, 1], np.int64),
             np.full(64, 3, np.int64), np.full(37, 1, np.int64),
             np.sort(rng.integers(1, 50, 1000))[::-1].astype(np.int64),
             np.sort(rng.zipf(1.3, 5000).clip(max=10**9))[::-1].astype(np.int64),
             np.sort((1e6 / np.arange(1, 20001) ** 1.1).astype(np.int64) + 1)[::-1].astype(np.int64)]
    for counts in cases:
        counts = np.ascontiguousarray(counts)
        V = len(counts)
        codelen, points, codes = oracle.huffman(counts)
        off = np.zeros(V + 1, np.int64); pts = np.zeros(max(int(codelen.sum()), 1), np.int32); bits = np.zeros(V, np.uint64)
        longest = algos_harness.harness_huffman(counts.ctypes.data, V, off.ctypes.data, pts.ctypes.data, len(pts), bits.ctypes.data)
        assert longest == (codelen.max() if V > 1 else 0) and longest <= 40
        assert np.array_equal(np.diff(off), codelen)
        for r in range(V):
            n = codelen[r]
            assert np.array_equal(pts[off[r]:off[r] + n], points[r, :n])
            assert [(int(bits[r]) >> d) & 1 for d in range(n)] == list(codes[r, :n])
        if V > 1:       # a full binary tree: Kraft equality, every inner node on some path
            assert abs(sum(2.0 ** -int(l) for l in codelen) - 1.0) < 1e-9
            assert set(pts[:off[V]].tolist()) == set(range(V - 1))
