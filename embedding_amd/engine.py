"""Thin object view of the C ABI (include/dge.h).  All compute happens in libdge.so on the GPU."""
import ctypes as C

import numpy as np

from ._native import TrainConfig, TrainStats, check, lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dev_ptr(t):
    """device pointer of a torch tensor (or a raw int)."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class DeviceGraph:
    """Edge store + alias tables in HBM (replaces the LayeredGraph store, J/LayeredGraph.java:142-226)."""

    def __init__(self, device=0):
        h = C.c_void_p(0)
        check(lib.dge_graph_create(C.byref(h), int(device)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib.dge_graph_free(self._h)
            self._h = None

    __del__ = close

    def set_stream(self, stream_ptr):
        check(lib.dge_graph_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def add_edges(self, src, dst, w):
        src = np.ascontiguousarray(src, np.int32); dst = np.ascontiguousarray(dst, np.int32)
        w = np.ascontiguousarray(w, np.float64)
        if not (len(src) == len(dst) == len(w)):
            raise ValueError("src/dst/w lengths differ")
        check(lib.dge_graph_add_edges(self._h, _ptr(src), _ptr(dst), _ptr(w), len(src)))

    def add_edges_device(self, src_t, dst_t, w_t):
        """src/dst int32, w float64 torch tensors on this device."""
        n = int(src_t.numel())
        check(lib.dge_graph_add_edges_device(self._h, _dev_ptr(src_t), _dev_ptr(dst_t), _dev_ptr(w_t), n))

    def set_sources(self, v, stream_sum=False):
        v = np.ascontiguousarray(v, np.int32)
        check(lib.dge_graph_set_sources(self._h, _ptr(v), len(v), int(bool(stream_sum))))

    def reserve_vertices(self, n):
        check(lib.dge_graph_reserve_vertices(self._h, int(n)))

    def set_out_degree(self, out_degree):
        """Vertex.outDegree values as the host holds them (a public field of the reference), one per vertex."""
        od = np.ascontiguousarray(out_degree, np.float64)
        check(lib.dge_graph_set_out_degree(self._h, _ptr(od), len(od)))

    def set_source_weight_sum(self, s):
        check(lib.dge_graph_set_source_weight_sum(self._h, float(s)))

    def get_csr(self, tables=True):
        """The whole store in CSR order: row_ptr, nbr, weight, out_degree (+ prob, alias when the tables are built)."""
        V, E = self.num_vertices, self.num_edges
        row_ptr = np.zeros(V + 1, np.int64); nbr = np.zeros(max(E, 1), np.int32); wt = np.zeros(max(E, 1), np.float64)
        od = np.zeros(max(V, 1), np.float64)
        prob = np.zeros(max(E, 1), np.float64) if tables else None; alias = np.zeros(max(E, 1), np.int32) if tables else None
        check(lib.dge_graph_get_csr(self._h, _ptr(row_ptr), _ptr(nbr), _ptr(wt), _ptr(prob), _ptr(alias), _ptr(od), V, max(E, 1)))
        E = int(row_ptr[V])
        out = dict(row_ptr=row_ptr, nbr=nbr[:E], weight=wt[:E], out_degree=od[:V])
        if tables:
            out.update(prob=prob[:E], alias=alias[:E])
        return out

    def keep_top_k(self, k):
        check(lib.dge_graph_keep_top_k(self._h, int(k)))

    def build_alias(self, exact=True):
        check(lib.dge_graph_build_alias(self._h, int(bool(exact))))

    @property
    def num_vertices(self):
        n = C.c_int32(0); check(lib.dge_graph_num_vertices(self._h, C.byref(n))); return n.value

    @property
    def num_edges(self):
        n = C.c_int64(0); check(lib.dge_graph_num_edges(self._h, C.byref(n))); return n.value

    def get_alias(self, v, tables=True):
        k = C.c_int32(0); od = C.c_double(0)
        check(lib.dge_graph_get_alias(self._h, int(v), None, None, None, None, 0, C.byref(k), C.byref(od)))
        n = max(k.value, 1)
        prob = np.zeros(n, np.float64); alias = np.zeros(n, np.int32); nbr = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
        check(lib.dge_graph_get_alias(self._h, int(v), _ptr(prob) if tables else None, _ptr(alias) if tables else None,
                                      _ptr(nbr), _ptr(wt), n, C.byref(k), C.byref(od)))
        kk = k.value
        return dict(prob=prob[:kk], alias=alias[:kk], nbr=nbr[:kk], weight=wt[:kk], out_degree=od.value)

    def get_source_alias(self):
        k = C.c_int32(0); ws = C.c_double(0)
        check(lib.dge_graph_get_source_alias(self._h, None, None, None, 0, C.byref(k), C.byref(ws)))
        n = max(k.value, 1)
        prob = np.zeros(n, np.float64); alias = np.zeros(n, np.int32); src = np.zeros(n, np.int32)
        check(lib.dge_graph_get_source_alias(self._h, _ptr(prob), _ptr(alias), _ptr(src), n, C.byref(k), C.byref(ws)))
        kk = k.value
        return dict(prob=prob[:kk], alias=alias[:kk], src=src[:kk], weight_sum=ws.value)

    def sample_next(self, v, x):
        nxt = C.c_int32(-1)
        check(lib.dge_graph_sample_next(self._h, int(v), float(x), C.byref(nxt)))
        return nxt.value

    def sample_walks(self, n_walks, max_len, seed, rng_mode=1, first_index=0, return_draws=False):
        out = np.empty((int(n_walks), int(max_len)), np.int32)
        draws = C.c_int64(0)
        check(lib.dge_sample_walks(self._h, int(n_walks), int(max_len), int(seed), int(rng_mode), int(first_index),
                                   _ptr(out), C.byref(draws)))
        return (out, draws.value) if return_draws else out

    def sample_walks_device(self, n_walks, max_len, seed, rng_mode=1, first_index=0):
        h = C.c_void_p(0); draws = C.c_int64(0)
        check(lib.dge_sample_walks_device(self._h, int(n_walks), int(max_len), int(seed), int(rng_mode), int(first_index),
                                          C.byref(h), C.byref(draws)))
        return WalkCorpus(h, self.device)

    def sample_walks_into(self, corpus, row0, n_walks, seed, first_index):
        check(lib.dge_sample_walks_into(self._h, corpus._h, int(row0), int(n_walks), int(seed), int(first_index)))


class WalkCorpus:
    """Walk corpus int32 [n x L] in HBM (replaces the .seq text corpus between J/CrossTimeGraph.java:132-141
    and J/DeepWalk.java:49-56)."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device

    @classmethod
    def from_host(cls, walks, device=0):
        walks = np.ascontiguousarray(walks, np.int32)
        n, L = walks.shape
        h = C.c_void_p(0)
        check(lib.dge_walks_from_host(int(device), _ptr(walks), n, L, C.byref(h)))
        return cls(h, int(device))

    def close(self):
        if getattr(self, "_h", None):
            lib.dge_walks_free(self._h)
            self._h = None

    __del__ = close

    @property
    def shape(self):
        n = C.c_int64(0); L = C.c_int32(0)
        check(lib.dge_walks_info(self._h, C.byref(n), C.byref(L), None))
        return n.value, L.value

    def to_host(self):
        n, L = self.shape
        out = np.empty((n, L), np.int32)
        check(lib.dge_walks_to_host(self._h, _ptr(out), n * L))
        return out

    def add_position_prefix(self, region_count):
        check(lib.dge_walks_add_position_prefix(self._h, int(region_count)))

    def count_tokens(self, n_vertices, d_counts, row0=0, n_rows=None):
        """d_counts: torch int64 tensor [n_vertices] on this device (accumulated into)."""
        if n_rows is None:
            n_rows = self.shape[0] - row0
        check(lib.dge_count_tokens(self._h, int(row0), int(n_rows), int(n_vertices), _dev_ptr(d_counts)))


def make_config(dim, window, n_vertices, negative=5, min_count=2, epochs=1, workers=0, alpha=0.025, min_alpha=1e-4,
                seed=1, table_size=100_000_000, update_policy=0, use_hs=False):
    """struct dge_train_config, field by field (a ctypes view, not a mirror of DeepWalk: use_hs defaults to the plain
    negative-sampling path that BASELINE.json's metric and bench.py are about).  `deepwalk_config` is the reference's setting."""
    return TrainConfig(int(dim), int(window), int(negative), int(min_count), int(epochs), int(workers), float(alpha),
                       float(min_alpha), int(seed), int(table_size), int(n_vertices), int(update_policy), int(bool(use_hs)), 0)


def deepwalk_config(region_level, num_layer, n_vertices, use_hs=True, **kw):
    """What J/DeepWalk.java:62-76 builds: layerSize 20 ("tract") or 2 ("CA"), windowSize = LayeredGraph.numLayer, 5 negatives,
    minWordFrequency 2, one iteration, DL4J's learning rates — and the hierarchical-softmax term ON, because the builder never
    calls useHierarchicSoftmax(false) (java/embedding/DeepWalk.java and the C++ mirror default to the same)."""
    return make_config(2 if region_level == "CA" else 20, num_layer, n_vertices, negative=5, min_count=2, epochs=1, use_hs=use_hs, **kw)


class SgnsModel:
    """Vocabulary + syn0/syn1neg in HBM (replaces the DL4J Word2Vec object of J/DeepWalk.java:73-82)."""

    def __init__(self, handle, device, cfg):
        self._h = handle
        self.device = device
        self.cfg = cfg
        self.torch_device = "cuda:%d" % int(device)

    @classmethod
    def create(cls, cfg, d_counts, device=0):
        h = C.c_void_p(0)
        check(lib.dge_model_create(int(device), C.byref(cfg), _dev_ptr(d_counts), C.byref(h)))
        return cls(h, int(device), cfg)

    @classmethod
    def create_placed(cls, cfg, d_counts, device, probe, trials=3, first=None):
        """Start-up placement trials.  Where the driver puts a model's tables decides its training speed by up to 15 % (two models of one
        process differ reproducibly, a model freed and re-created in the same memory does not: profiles/r02_box_drift.txt), and no cheap probe
        of the memory predicts it — so: create `trials` models side by side (the earlier ones kept, or the next would get their memory
        back), time `probe(model) -> milliseconds` on each (one launch of the caller's own workload), keep the fastest, free the others.
        -> (model, [milliseconds of every trial])."""
        models = [first if first is not None else cls.create(cfg, d_counts, device)]
        ms = [float(probe(models[0]))]
        for _ in range(max(int(trials), 1) - 1):
            models.append(cls.create(cfg, d_counts, device)); ms.append(float(probe(models[-1])))
        best = min(range(len(ms)), key=ms.__getitem__)
        for i, m in enumerate(models):
            if i != best:
                m.close()
        return models[best], ms

    @classmethod
    def fit(cls, walks, cfg, device=0):
        """w2v.fit() one-shot: walks is a host int32 [n x L] array or a WalkCorpus."""
        h = C.c_void_p(0)
        if isinstance(walks, WalkCorpus):
            check(lib.dge_train_sgns_device(walks._h, C.byref(cfg), C.byref(h)))
            device = walks.device
        else:
            walks = np.ascontiguousarray(walks, np.int32)
            n, L = walks.shape
            check(lib.dge_train_sgns(int(device), _ptr(walks), n, L, C.byref(cfg), C.byref(h)))
        return cls(h, int(device), cfg)

    def close(self):
        if getattr(self, "_h", None):
            lib.dge_model_free(self._h)
            self._h = None

    __del__ = close

    def set_stream(self, stream_ptr):
        check(lib.dge_model_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def train(self, corpus, row0=0, n_rows=None, walk_index_base=0, epoch=0, words_before=0, words_scale=1.0, total_walks=0):
        if n_rows is None:
            n_rows = corpus.shape[0] - row0
        check(lib.dge_model_train(self._h, corpus._h, int(row0), int(n_rows), int(walk_index_base), int(epoch),
                                  int(words_before), float(words_scale), int(total_walks)))

    def walk_and_train(self, graph, corpus, row0, n_rows, walk_seed, walk_index_base, epoch=0, words_before=0,
                       words_scale=1.0, total_walks=0):
        check(lib.dge_model_walk_and_train(self._h, graph._h, corpus._h, int(row0), int(n_rows), int(walk_seed),
                                           int(walk_index_base), int(epoch), int(words_before), float(words_scale),
                                           int(total_walks)))

    def vectors(self):
        p = C.c_void_p(0); ids = C.c_void_p(0); V = C.c_int64(0); D = C.c_int32(0)
        check(lib.dge_model_vectors(self._h, C.byref(p), C.byref(ids), C.byref(V), C.byref(D)))
        if V.value == 0:
            return np.zeros((0, D.value), np.float32), np.zeros(0, np.int32)
        syn0 = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(V.value * D.value,)).copy().reshape(V.value, D.value)
        vid = np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_int32)), shape=(V.value,)).copy()
        return syn0, vid

    def syn1neg(self):
        syn0, _ = self.vectors()
        p = C.c_void_p(0)
        check(lib.dge_model_syn1neg(self._h, C.byref(p)))
        if syn0.size == 0:
            return np.zeros_like(syn0)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(syn0.size,)).copy().reshape(syn0.shape)

    def syn1(self):
        """Inner-node table of the hierarchical softmax, [V-1 x dim] (models created with use_hs)."""
        p = C.c_void_p(0); rows = C.c_int64(0)
        check(lib.dge_model_syn1(self._h, C.byref(p), C.byref(rows)))
        n = rows.value * self.cfg.dim
        if n == 0:
            return np.zeros((0, self.cfg.dim), np.float32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).copy().reshape(rows.value, self.cfg.dim)

    def huffman(self):
        """Huffman paths of the vocabulary rows: (offsets[V+1], points, codes) — bit d of codes[r] is the branch at
        points[offsets[r] + d] (word2vec.c CreateBinaryTree)."""
        _, vid = self.vectors()
        V = len(vid)
        po, pp, pc = C.c_void_p(0), C.c_void_p(0), C.c_void_p(0)
        check(lib.dge_model_huffman(self._h, C.byref(po), C.byref(pp), C.byref(pc)))
        if V == 0:
            return np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.uint64)
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_int64)), shape=(V + 1,)).copy()
        n = int(off[-1])
        pts = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_int32)), shape=(max(n, 1),)).copy()[:n]
        codes = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint64)), shape=(V,)).copy()
        return off, pts, codes

    def counts(self):
        _, vid = self.vectors()
        p = C.c_void_p(0)
        check(lib.dge_model_counts(self._h, C.byref(p)))
        if len(vid) == 0:
            return np.zeros(0, np.int64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), shape=(len(vid),)).copy()

    def table(self):
        p = C.c_void_p(0); T = C.c_int64(0)
        check(lib.dge_model_table(self._h, C.byref(p), C.byref(T)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(T.value,)).copy()

    def row_rates(self):
        """(GB/s of random row reads, GB/s of random row read + write-back, lock-word exchanges per second, unigram-table look-ups per
        second) on this model's own memory (diagnostic: include/dge.h, dge_model_row_rates)."""
        v = [C.c_double(0) for _ in range(4)]
        check(lib.dge_model_row_rates(self._h, *[C.byref(x) for x in v])); return tuple(x.value for x in v)

    def table_runs(self):
        """(runs, exceptions) of the negative-sampling table's run form (include/dge.h, dge_model_table_runs); (0, 0): this model has none."""
        n, e = C.c_int32(0), C.c_int32(0)
        check(lib.dge_model_table_runs(self._h, C.byref(n), C.byref(e)))
        return n.value, e.value

    def table_placement(self):
        """What dge_model_create's probe-selected table allocation saw: [(candidates probed, best GB/s = the one kept, worst GB/s)] for syn0, syn1neg
        (and syn1 under hierarchical softmax)."""
        out = []
        for t in range(3 if self.cfg.use_hs else 2):
            n, a, b = C.c_int32(0), C.c_double(0), C.c_double(0)
            check(lib.dge_model_table_placement(self._h, t, C.byref(n), C.byref(a), C.byref(b)))
            out.append((n.value, round(a.value), round(b.value)))
        return out

    def tune_placement(self, corpus, row0=0, n_rows=None, candidates=3):
        """Placement search (include/dge.h: dge_model_tune_placement): -> (probe ms before, probe ms after, arrays moved).  The model's tables,
        counters and statistics are as before the call."""
        if n_rows is None:
            n_rows = corpus.shape[0] - row0
        a, b, n = C.c_double(0), C.c_double(0), C.c_int32(0)
        check(lib.dge_model_tune_placement(self._h, corpus._h, int(row0), int(n_rows), int(candidates), C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def placement_search(self):
        """{"runs", "probe_ms_before", "probe_ms_after", "arrays_moved"} of the placement search on this model (runs == 0: it never ran — the one-shot
        fit skips it where it cannot pay)."""
        n, a, b, mv = C.c_int32(0), C.c_double(0), C.c_double(0), C.c_int32(0)
        check(lib.dge_model_placement_search(self._h, C.byref(n), C.byref(a), C.byref(b), C.byref(mv)))
        return {"runs": n.value, "probe_ms_before": a.value, "probe_ms_after": b.value, "arrays_moved": mv.value}

    def stats(self):
        s = TrainStats()
        check(lib.dge_model_stats(self._h, C.byref(s)))
        return dict(pairs=s.pairs, words=s.words, kernel_ms=s.kernel_ms, walk_kernel_ms=s.walk_kernel_ms, launches=s.launches)

    def reset_stats(self):
        check(lib.dge_model_reset_stats(self._h))

    def write_vec(self, path, names=None, header=False):
        arr = None
        if names is not None:
            arr = (C.c_char_p * len(names))(*[n.encode() if n is not None else None for n in names])
        check(lib.dge_write_vec(self._h, arr, str(path).encode(), int(bool(header))))

    # --- multi-GPU exchange (include/dge.h, last section)
    def schedule(self):
        """What the latest launch resolved update_policy 0 / workers 0 to."""
        pol, w, hot = C.c_int32(0), C.c_int64(0), C.c_int32(0)
        check(lib.dge_model_schedule(self._h, C.byref(pol), C.byref(w), C.byref(hot)))
        return {"update_policy": pol.value, "workers": w.value, "hot_rows": hot.value}

    def lock_stats(self):
        """{"pairs_put_back", "rounds_short", "rounds"} of the block schedule's lock kernels since reset_stats (include/dge.h: dge_model_lock_stats)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib.dge_model_lock_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"pairs_put_back": a.value, "rounds_short": b.value, "rounds": c.value}

    def kernel(self):
        """Name and form of the trainer kernel the latest launch ran (include/dge.h: dge_model_kernel)."""
        buf = C.create_string_buffer(256)
        check(lib.dge_model_kernel(self._h, buf, 256))
        return buf.value.decode()

    # ---- multi-GPU block schedule (include/dge.h: dge_model_set_partition)
    def set_partition(self, n_parts, ctx_part=0, tgt_part=0):
        check(lib.dge_model_set_partition(self._h, int(n_parts), int(ctx_part), int(tgt_part)))

    def partition_floats(self, n_parts):
        n = C.c_int64(0); check(lib.dge_model_partition_floats(self._h, int(n_parts), C.byref(n))); return n.value

    def export_partition(self, table, n_parts, part, d_buf):
        check(lib.dge_model_export_partition(self._h, int(table), int(n_parts), int(part), _dev_ptr(d_buf)))

    def import_partition(self, table, n_parts, part, d_buf):
        check(lib.dge_model_import_partition(self._h, int(table), int(n_parts), int(part), _dev_ptr(d_buf)))

    def export_partition_async(self, table, n_parts, part, d_buf, consumer_stream=0):
        """Stream-ordered export (include/dge.h): the pack kernel runs on the model's stream and `consumer_stream` (a hipStream_t as an integer, 0 = the
        legacy default stream, e.g. torch.cuda.current_stream().cuda_stream) waits for it; the host does not."""
        check(lib.dge_model_export_partition_async(self._h, int(table), int(n_parts), int(part), _dev_ptr(d_buf), C.c_void_p(int(consumer_stream))))

    def import_partition_async(self, table, n_parts, part, d_buf, producer_stream=0):
        """Stream-ordered import: the model's stream waits for what `producer_stream` holds now, then unpacks; the host does not wait."""
        check(lib.dge_model_import_partition_async(self._h, int(table), int(n_parts), int(part), _dev_ptr(d_buf), C.c_void_p(int(producer_stream))))

    def stream(self):
        """The hipStream_t (as an integer) the model's launches are enqueued on."""
        p = C.c_void_p(0); check(lib.dge_model_stream(self._h, C.byref(p))); return p.value or 0

    def sync_size(self):
        n = C.c_int64(0); check(lib.dge_model_sync_size(self._h, C.byref(n))); return n.value

    def snapshot(self):
        check(lib.dge_model_snapshot(self._h))

    def export_delta(self, d_buf):
        check(lib.dge_model_export_delta(self._h, _dev_ptr(d_buf)))

    def import_delta(self, d_buf, scale):
        check(lib.dge_model_import_delta(self._h, _dev_ptr(d_buf), float(scale)))


def host_sync_count():
    """Blocking waits (stream / device / event synchronisations, blocking copies) the library has made in this process so far (include/dge.h:
    dge_host_sync_count): tests assert that an episode of the block schedule adds none."""
    n = C.c_int64(0); check(lib.dge_host_sync_count(C.byref(n))); return n.value


def build_stamp():
    """{"kernels": hash, "sorted": hash} of the trainer kernels' sources the loaded libdge.so was built from (include/dge.h: dge_build_stamp)."""
    return dict(kv.split("=") for kv in lib.dge_build_stamp().decode().split())


TUNING_KNOBS = {"hot_rows": 0, "hs_drain": 1, "force_segments": 2, "segment_shift": 3, "sorted_chunk": 4, "sorted_walks": 5, "workers": 6, "static_walks": 7, "hs_cold": 8, "hs_wave": 9, "acc_rows": 10, "acc_drain": 11, "table_runs": 12, "block_syn0_free": 13, "hs_centre": 14, "hs_hot_kb": 15, "allow_unsafe": 16, "watchdog_ms": 17, "hs_copies": 18, "small_rows": 19}      # include/dge.h: DGE_TUNE_*


class tuning:
    """Context manager around dge_set_tuning (ablation / test knobs of the trainer; process-wide):
    `with tuning(hot_rows=100): model.train(...)`.  Leaving the block puts the knobs back to what they were before it."""

    def __init__(self, **knobs):
        self.knobs = {TUNING_KNOBS[k]: int(v) for k, v in knobs.items()}
        self.before = {}

    def __enter__(self):
        for k, v in self.knobs.items():
            old = C.c_int64(-1)
            check(lib.dge_get_tuning(k, C.byref(old)))
            self.before[k] = old.value
            check(lib.dge_set_tuning(k, v))
        return self

    def __exit__(self, *exc):
        for k in self.knobs:
            check(lib.dge_set_tuning(k, self.before.get(k, -1)))
        return False
