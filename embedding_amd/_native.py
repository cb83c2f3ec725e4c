"""ctypes binding of include/dge.h (libdge.so).  No fallback: if the HIP library is missing the import fails."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdge.so")

DGE_OK, DGE_ERR_ARG, DGE_ERR_RANGE, DGE_ERR_TOPK, DGE_ERR_CAP, DGE_ERR_STATE, DGE_ERR_DEVICE, DGE_ERR_IO = range(8)


class DgeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdge error %d: %s" % (code, msg))
        self.code = code


class TrainConfig(C.Structure):
    """struct dge_train_config (include/dge.h) — the Word2Vec.Builder contract of J/DeepWalk.java:73-76."""
    _fields_ = [
        ("dim", C.c_int32), ("window", C.c_int32), ("negative", C.c_int32), ("min_count", C.c_int32),
        ("epochs", C.c_int32), ("workers", C.c_int32), ("alpha", C.c_float), ("min_alpha", C.c_float),
        ("seed", C.c_uint64), ("table_size", C.c_int64), ("n_vertices", C.c_int32), ("update_policy", C.c_int32),
        ("use_hs", C.c_int32), ("reserved", C.c_int32),
    ]


class TrainStats(C.Structure):
    _fields_ = [("pairs", C.c_int64), ("words", C.c_int64), ("kernel_ms", C.c_double),
                ("walk_kernel_ms", C.c_double), ("launches", C.c_int64)]


# every symbol include/dge.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _dbl, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_int
_P = C.POINTER
SIGNATURES = {
    "dge_last_error": (C.c_char_p, []),
    "dge_version": (_int, []),
    "dge_build_stamp": (C.c_char_p, []),
    "dge_device_count": (_int, [_P(_int)]),
    "dge_graph_create": (_int, [_P(_vp), _int]),
    "dge_graph_free": (None, [_vp]),
    "dge_graph_set_stream": (_int, [_vp, _vp]),
    "dge_graph_add_edges": (_int, [_vp, _vp, _vp, _vp, _i64]),
    "dge_graph_add_edges_device": (_int, [_vp, _vp, _vp, _vp, _i64]),
    "dge_graph_set_sources": (_int, [_vp, _vp, _i64, _int]),
    "dge_graph_reserve_vertices": (_int, [_vp, _i32]),
    "dge_graph_set_out_degree": (_int, [_vp, _vp, _i32]),
    "dge_graph_set_source_weight_sum": (_int, [_vp, _dbl]),
    "dge_graph_get_csr": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64]),
    "dge_graph_keep_top_k": (_int, [_vp, _i32]),
    "dge_graph_build_alias": (_int, [_vp, _int]),
    "dge_graph_num_vertices": (_int, [_vp, _P(_i32)]),
    "dge_graph_num_edges": (_int, [_vp, _P(_i64)]),
    "dge_graph_get_alias": (_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _P(_i32), _P(_dbl)]),
    "dge_graph_get_source_alias": (_int, [_vp, _vp, _vp, _vp, _i32, _P(_i32), _P(_dbl)]),
    "dge_graph_sample_next": (_int, [_vp, _i32, _dbl, _P(_i32)]),
    "dge_sample_walks": (_int, [_vp, _i64, _i32, _i64, _int, _i64, _vp, _P(_i64)]),
    "dge_sample_walks_device": (_int, [_vp, _i64, _i32, _i64, _int, _i64, _P(_vp), _P(_i64)]),
    "dge_sample_walks_into": (_int, [_vp, _vp, _i64, _i64, _i64, _i64]),
    "dge_walks_from_host": (_int, [_int, _vp, _i64, _i32, _P(_vp)]),
    "dge_walks_to_host": (_int, [_vp, _vp, _i64]),
    "dge_walks_info": (_int, [_vp, _P(_i64), _P(_i32), _P(_vp)]),
    "dge_walks_add_position_prefix": (_int, [_vp, _i32]),
    "dge_walks_free": (None, [_vp]),
    "dge_count_tokens": (_int, [_vp, _i64, _i64, _i32, _vp]),
    "dge_model_create": (_int, [_int, _P(TrainConfig), _vp, _P(_vp)]),
    "dge_model_set_stream": (_int, [_vp, _vp]),
    "dge_model_train": (_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i64, _dbl, _i64]),
    "dge_train_sgns": (_int, [_int, _vp, _i64, _i32, _P(TrainConfig), _P(_vp)]),
    "dge_train_sgns_device": (_int, [_vp, _P(TrainConfig), _P(_vp)]),
    "dge_model_walk_and_train": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i64, _dbl, _i64]),
    "dge_model_vectors": (_int, [_vp, _P(_vp), _P(_vp), _P(_i64), _P(_i32)]),
    "dge_model_syn1neg": (_int, [_vp, _P(_vp)]),
    "dge_model_syn1": (_int, [_vp, _P(_vp), _P(_i64)]),
    "dge_model_huffman": (_int, [_vp, _P(_vp), _P(_vp), _P(_vp)]),
    "dge_model_counts": (_int, [_vp, _P(_vp)]),
    "dge_model_table": (_int, [_vp, _P(_vp), _P(_i64)]),
    "dge_model_stats": (_int, [_vp, _P(TrainStats)]),
    "dge_model_row_rates": (_int, [_vp, _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "dge_model_table_placement": (_int, [_vp, _i32, _vp, _vp, _vp]),
    "dge_model_table_runs": (_int, [_vp, _vp, _vp]),
    "dge_model_tune_placement": (_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "dge_model_placement_search": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "dge_model_reset_stats": (_int, [_vp]),
    "dge_model_schedule": (_int, [_vp, _P(_i32), _P(_i64), _P(_i32)]),
    "dge_model_kernel": (_int, [_vp, C.c_char_p, _i32]),
    "dge_model_lock_stats": (_int, [_vp, _P(_i64), _P(_i64), _P(_i64)]),
    "dge_write_vec": (_int, [_vp, _vp, C.c_char_p, _int]),
    "dge_model_free": (None, [_vp]),
    "dge_model_set_partition": (_int, [_vp, _i32, _i32, _i32]),
    "dge_model_partition_floats": (_int, [_vp, _i32, _P(_i64)]),
    "dge_model_export_partition": (_int, [_vp, _int, _i32, _i32, _vp]),
    "dge_model_import_partition": (_int, [_vp, _int, _i32, _i32, _vp]),
    "dge_model_export_partition_async": (_int, [_vp, _int, _i32, _i32, _vp, _vp]),
    "dge_model_import_partition_async": (_int, [_vp, _int, _i32, _i32, _vp, _vp]),
    "dge_model_stream": (_int, [_vp, _P(_vp)]),
    "dge_host_sync_count": (_int, [_P(_i64)]),
    "dge_model_sync_size": (_int, [_vp, _P(_i64)]),
    "dge_model_snapshot": (_int, [_vp]),
    "dge_model_export_delta": (_int, [_vp, _vp]),
    "dge_model_import_delta": (_int, [_vp, _vp, C.c_float]),
    "dge_comm_unique_id": (_int, [_vp]),
    "dge_comm_create": (_int, [_P(_vp), _vp, _int, _int, _int]),
    "dge_comm_free": (None, [_vp]),
    "dge_model_allreduce_deltas": (_int, [_vp, _vp]),
    "dge_model_ring_pass": (_int, [_vp, _vp, _i32]),
    "dge_model_gather_table": (_int, [_vp, _vp, _int]),
    "dge_set_tuning": (_int, [_i32, _i64]),
    "dge_get_tuning": (_int, [_i32, _vp]),
    "dge_ndcg_at_k": (_int, [_int, _vp, _i32, _vp, _i32, _i32, _i32, _P(_dbl), _P(_dbl)]),
    "dge_knn_cosine": (_int, [_int, _vp, _i32, _i32, _i32, _vp, _vp, _P(_dbl)]),
    "dge_selftest_locked_rows": (_int, [_int, _i32, _i64, _i32, C.c_uint64, _i32, _P(_i64), _P(_dbl)]),
    "dge_selftest_atomics_wave": (_int, [_int, _i32, _i32, _i32, _i32, _i32, C.c_uint64, _P(_i64), _P(_dbl)]),
    "dge_selftest_atomics_wave_block": (_int, [_int, _i32, _i32, _i32, _i32, _i32, _i32, C.c_uint64, _P(_i64), _P(_dbl)]),
    "dge_selftest_fmt_g9": (_int, [_i64, C.c_uint64, _P(_i64), _P(_i64)]),
    "dge_selftest_hot_add": (_int, [_int, _i32, _i64, _i32, _i32, C.c_uint64, _P(_i64), _P(_dbl)]),
}


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own libamdhip64.so (SONAME libamdhip64.so.7,
    loaded by file name), libdge.so needs libamdhip64.so.7: if both copies get loaded, the second one finds no
    device.  When torch is installed, load ITS runtime first so libdge.so binds to it by SONAME; torch tensors
    (torch.distributed / RCCL plumbing) and libdge then share streams and memory."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(libdir, name)
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "embedding_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (rt, at) in SIGNATURES.items():
        f = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        f.restype = rt
        f.argtypes = at
    return lib


lib = load()


def check(rc):
    if rc != 0:
        raise DgeError(rc, (lib.dge_last_error() or b"").decode("utf-8", "replace"))
