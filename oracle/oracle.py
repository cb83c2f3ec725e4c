"""ctypes view of the CPU oracle (oracle/libdge_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from embedding_amd/ (the product fails loudly without its HIP library).
Walk half pinned by T/LayeredGraphTest.java:12-44; SGNS half: PARITY UNPINNED (see dge_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdge_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("dge_oracle.c", "dge_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libdge_oracle.so"])
    return _SO


class TrainConfig(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("window", C.c_int32), ("negative", C.c_int32), ("min_count", C.c_int32),
        ("epochs", C.c_int32), ("threads", C.c_int32), ("alpha", C.c_float), ("min_alpha", C.c_float),
        ("seed", C.c_uint64), ("table_size", C.c_int64), ("arith", C.c_int32), ("n_vertices", C.c_int32),
        ("walk_index_base", C.c_int64), ("total_walks", C.c_int64), ("total_words", C.c_int64),
        ("words_before", C.c_int64), ("use_hs", C.c_int32), ("part_n", C.c_int32),
        ("sorted_chunk", C.c_int32), ("sorted_walks", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    P = C.POINTER
    L.orc_jrand_seed.argtypes = [P(u64), i64]
    L.orc_jrand_next_int.argtypes = [P(u64)]; L.orc_jrand_next_int.restype = i32
    L.orc_jrand_next_double.argtypes = [P(u64)]; L.orc_jrand_next_double.restype = dbl
    L.orc_jrand_jump.argtypes = [P(u64), u64]
    L.orc_graph_create.restype = vp
    L.orc_graph_free.argtypes = [vp]
    L.orc_graph_add_edges.argtypes = [vp, vp, vp, vp, i64]
    L.orc_graph_set_sources.argtypes = [vp, vp, i64, C.c_int]
    L.orc_graph_keep_top_k.argtypes = [vp, i32]
    L.orc_graph_reserve_vertices.argtypes = [vp, i32]
    L.orc_graph_set_out_degree.argtypes = [vp, vp, i32]
    L.orc_graph_set_source_weight_sum.argtypes = [vp, dbl]
    L.orc_graph_get_csr.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.orc_graph_build_alias.argtypes = [vp, C.c_int]
    L.orc_graph_num_vertices.argtypes = [vp]; L.orc_graph_num_vertices.restype = i32
    L.orc_graph_num_edges.argtypes = [vp]; L.orc_graph_num_edges.restype = i64
    L.orc_graph_get_alias.argtypes = [vp, i32, vp, vp, vp, vp, i32, P(i32), P(dbl)]
    L.orc_graph_get_source_alias.argtypes = [vp, vp, vp, vp, i32, P(i32), P(dbl)]
    L.orc_graph_sample_next.argtypes = [vp, i32, dbl, P(i32)]
    L.orc_sample_walks.argtypes = [vp, i64, i32, i64, C.c_int, i64, vp, P(i64)]
    L.orc_train_sgns.argtypes = [vp, i64, i32, P(TrainConfig), P(vp)]
    L.orc_train_sgns_from.argtypes = [vp, i64, i32, P(TrainConfig), vp, vp, vp, P(vp)]
    L.orc_train_sgns_from_hs.argtypes = [vp, i64, i32, P(TrainConfig), vp, vp, vp, vp, P(vp)]
    for name, rt in (("vocab_size", i64), ("dim", i32), ("syn0", vp), ("syn1neg", vp), ("syn1", vp), ("vocab_ids", vp),
                     ("counts", vp), ("table", vp), ("pairs", i64), ("total_words", i64), ("seconds", dbl)):
        f = getattr(L, "orc_model_" + name); f.argtypes = [vp]; f.restype = rt
    L.orc_model_code.argtypes = [vp, i64, vp, vp, i32]; L.orc_model_code.restype = i32
    L.orc_huffman.argtypes = [vp, i64, vp, vp, vp]
    L.orc_model_free.argtypes = [vp]
    L.orc_exp_table.argtypes = [C.c_int]; L.orc_exp_table.restype = C.c_float
    L.orc_mix64.argtypes = [u64]; L.orc_mix64.restype = u64
    L.orc_set_plain.argtypes = [C.c_int]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class JavaRandom:
    """java.util.Random (public spec) — the RNG at J/LayeredGraph.java:14."""

    def __init__(self, seed):
        self._s = C.c_uint64(0)
        lib().orc_jrand_seed(C.byref(self._s), int(seed))

    def next_int(self):
        return int(lib().orc_jrand_next_int(C.byref(self._s)))

    def next_double(self):
        return float(lib().orc_jrand_next_double(C.byref(self._s)))

    def jump(self, n_lcg_steps):
        lib().orc_jrand_jump(C.byref(self._s), int(n_lcg_steps))

    @property
    def state(self):
        return int(self._s.value)


class Graph:
    """Edge store + alias sampler following J/LayeredGraph.java (ids = insertion ordinals)."""

    def __init__(self):
        self._h = lib().orc_graph_create()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_graph_free(self._h)
            self._h = None

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError("oracle %s failed: rc=%d" % (what, rc))

    def add_edges(self, src, dst, w):
        src = np.ascontiguousarray(src, np.int32); dst = np.ascontiguousarray(dst, np.int32)
        w = np.ascontiguousarray(w, np.float64)
        self._chk(lib().orc_graph_add_edges(self._h, _ptr(src), _ptr(dst), _ptr(w), len(src)), "add_edges")

    def set_sources(self, v, stream_sum=False):
        v = np.ascontiguousarray(v, np.int32)
        self._chk(lib().orc_graph_set_sources(self._h, _ptr(v), len(v), int(stream_sum)), "set_sources")

    def reserve_vertices(self, n):
        self._chk(lib().orc_graph_reserve_vertices(self._h, int(n)), "reserve_vertices")

    def set_out_degree(self, od):
        od = np.ascontiguousarray(od, np.float64)
        self._chk(lib().orc_graph_set_out_degree(self._h, _ptr(od), len(od)), "set_out_degree")

    def set_source_weight_sum(self, s):
        self._chk(lib().orc_graph_set_source_weight_sum(self._h, float(s)), "set_source_weight_sum")

    def get_csr(self, tables=True):
        V = self.num_vertices
        row_ptr = np.zeros(V + 1, np.int64)
        self._chk(lib().orc_graph_get_csr(self._h, _ptr(row_ptr), None, None, None, None, None), "get_csr")
        E = int(row_ptr[V]); n = max(E, 1)
        nbr = np.zeros(n, np.int32); wt = np.zeros(n, np.float64); od = np.zeros(max(V, 1), np.float64)
        prob = np.zeros(n, np.float64) if tables else None; alias = np.zeros(n, np.int32) if tables else None
        self._chk(lib().orc_graph_get_csr(self._h, _ptr(row_ptr), _ptr(nbr), _ptr(wt), _ptr(prob) if tables else None,
                                          _ptr(alias) if tables else None, _ptr(od)), "get_csr")
        out = dict(row_ptr=row_ptr, nbr=nbr[:E], weight=wt[:E], out_degree=od[:V])
        if tables:
            out.update(prob=prob[:E], alias=alias[:E])
        return out

    def keep_top_k(self, k):
        self._chk(lib().orc_graph_keep_top_k(self._h, int(k)), "keep_top_k")

    def build_alias(self, exact=True):
        self._chk(lib().orc_graph_build_alias(self._h, int(bool(exact))), "build_alias")

    @property
    def num_vertices(self):
        return int(lib().orc_graph_num_vertices(self._h))

    @property
    def num_edges(self):
        return int(lib().orc_graph_num_edges(self._h))

    def get_alias(self, v):
        k = C.c_int32(0); od = C.c_double(0)
        lib().orc_graph_get_alias(self._h, v, None, None, None, None, 0, C.byref(k), C.byref(od))
        n = max(k.value, 1)
        prob = np.zeros(n, np.float64); alias = np.zeros(n, np.int32); nbr = np.zeros(n, np.int32)
        wt = np.zeros(n, np.float64)
        self._chk(lib().orc_graph_get_alias(self._h, v, _ptr(prob), _ptr(alias), _ptr(nbr), _ptr(wt), n,
                                            C.byref(k), C.byref(od)), "get_alias")
        kk = k.value
        return dict(prob=prob[:kk], alias=alias[:kk], nbr=nbr[:kk], weight=wt[:kk], out_degree=od.value)

    def get_source_alias(self):
        k = C.c_int32(0); ws = C.c_double(0)
        lib().orc_graph_get_source_alias(self._h, None, None, None, 0, C.byref(k), C.byref(ws))
        n = max(k.value, 1)
        prob = np.zeros(n, np.float64); alias = np.zeros(n, np.int32); src = np.zeros(n, np.int32)
        self._chk(lib().orc_graph_get_source_alias(self._h, _ptr(prob), _ptr(alias), _ptr(src), n,
                                                   C.byref(k), C.byref(ws)), "get_source_alias")
        kk = k.value
        return dict(prob=prob[:kk], alias=alias[:kk], src=src[:kk], weight_sum=ws.value)

    def sample_next(self, v, x):
        nxt = C.c_int32(-1)
        self._chk(lib().orc_graph_sample_next(self._h, int(v), float(x), C.byref(nxt)), "sample_next")
        return nxt.value

    def sample_walks(self, n_walks, max_len, seed, rng_mode=1, first_index=0, return_draws=False):
        out = np.empty((n_walks, max_len), np.int32)
        draws = C.c_int64(0)
        self._chk(lib().orc_sample_walks(self._h, n_walks, max_len, int(seed), int(rng_mode), int(first_index),
                                         _ptr(out), C.byref(draws)), "sample_walks")
        return (out, draws.value) if return_draws else out


class Model:
    def __init__(self, h):
        L = lib()
        self.V = int(L.orc_model_vocab_size(h)); self.dim = int(L.orc_model_dim(h))
        n = self.V * self.dim

        def arr(p, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()

        self.syn0 = arr(L.orc_model_syn0(h), n, np.float32).reshape(self.V, self.dim)
        self.syn1neg = arr(L.orc_model_syn1neg(h), n, np.float32).reshape(self.V, self.dim)
        p1 = L.orc_model_syn1(h)
        self.syn1 = arr(p1, max(self.V - 1, 0) * self.dim, np.float32).reshape(max(self.V - 1, 0), self.dim) if p1 else None
        self.vocab_ids = arr(L.orc_model_vocab_ids(h), self.V, np.int32)
        self.counts = arr(L.orc_model_counts(h), self.V, np.int64)
        self.pairs = int(L.orc_model_pairs(h)); self.total_words = int(L.orc_model_total_words(h))
        self.seconds = float(L.orc_model_seconds(h))
        self._table_ptr = L.orc_model_table(h)
        self._h = h

    def code(self, row):
        """Huffman path of vocabulary row `row`: (points, codes)."""
        pts = np.zeros(40, np.int32); cds = np.zeros(40, np.uint8)
        n = lib().orc_model_code(self._h, int(row), _ptr(pts), _ptr(cds), 40)
        return pts[:n].copy(), cds[:n].copy()

    def table(self, table_size):
        return np.ctypeslib.as_array(C.cast(self._table_ptr, C.POINTER(C.c_int32)), shape=(table_size,)).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_model_free(self._h)
            self._h = None


def train_sgns(walks, n_vertices, dim, window, negative=5, min_count=2, epochs=1, threads=1, alpha=0.025,
               min_alpha=1e-4, seed=1, table_size=100_000_000, arith=0, walk_index_base=0, total_walks=0,
               total_words=0, words_before=0, use_hs=False, part_n=0, sorted_chunk=0, sorted_walks=0, counts=None, syn0_init=None,
               syn1neg_init=None, syn1_init=None):
    """counts / syn0_init / syn1neg_init / syn1_init (the Huffman tree's inner nodes, [V-1 x dim], use_hs): continue from a given state (orc_train_sgns_from[_hs]) — the vocabulary, unigram table and total_words
    from `counts` ([n_vertices], the whole corpus) instead of from `walks`, the tables as a run has left them (rows in vocabulary order)."""
    walks = np.ascontiguousarray(walks, np.int32)
    n, L = walks.shape
    cfg = TrainConfig(dim, window, negative, min_count, epochs, threads, alpha, min_alpha, seed, table_size,
                      arith, n_vertices, walk_index_base, total_walks, total_words, words_before, int(bool(use_hs)), int(part_n),
                      int(sorted_chunk), int(sorted_walks))
    h = C.c_void_p(0)
    if counts is None and syn0_init is None and syn1neg_init is None and syn1_init is None:
        rc = lib().orc_train_sgns(_ptr(walks), n, L, C.byref(cfg), C.byref(h))
    else:
        cn = None if counts is None else np.ascontiguousarray(counts, np.int64)
        s0 = None if syn0_init is None else np.ascontiguousarray(syn0_init, np.float32)
        s1 = None if syn1neg_init is None else np.ascontiguousarray(syn1neg_init, np.float32)
        if cn is not None and len(cn) != n_vertices:
            raise ValueError("counts must have n_vertices entries")
        s2 = None if syn1_init is None else np.ascontiguousarray(syn1_init, np.float32)
        rc = lib().orc_train_sgns_from_hs(_ptr(walks), n, L, C.byref(cfg), None if cn is None else _ptr(cn), None if s0 is None else _ptr(s0),
                                          None if s1 is None else _ptr(s1), None if s2 is None else _ptr(s2), C.byref(h))
    if rc != 0:
        raise RuntimeError("oracle train_sgns failed: rc=%d" % rc)
    return Model(h)


def set_plain(on):
    """True: train_sgns runs the plain word2vec.c-shaped loop (the definition); False (default): the same operations, a pair's dot products side by side."""
    lib().orc_set_plain(int(bool(on)))


def huffman(counts):
    """word2vec.c CreateBinaryTree over counts sorted descending -> (codelen[V], points[V,40], codes[V,40])."""
    counts = np.ascontiguousarray(counts, np.int64)
    V = len(counts)
    codelen = np.zeros(V, np.int32); points = np.zeros((V, 40), np.int32); codes = np.zeros((V, 40), np.uint8)
    if lib().orc_huffman(_ptr(counts), V, _ptr(codelen), _ptr(points), _ptr(codes)) != 0:
        raise RuntimeError("oracle huffman failed")
    return codelen, points, codes


def exp_table():
    return np.array([lib().orc_exp_table(i) for i in range(1000)], np.float32)


def mix64(x):
    return int(lib().orc_mix64(int(x) & 0xFFFFFFFFFFFFFFFF))
