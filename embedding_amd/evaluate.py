"""Quality metric of the reference's evaluation scripts on the device (P/embeddingEvaluation_tract.py:169-196 pairwiseEstimator, :249-260
ndcg_atK): ctypes views of dge_knn_cosine / dge_ndcg_at_k.  Used as the statistical parity check between training schedules (in-order /
Hogwild / multi-GPU) on one slice.  The float64 host restatement these kernels are checked against is test infrastructure: oracle/quality.py."""
import numpy as np


def knn_cosine_gpu(features, k, device=0):
    """The same KNN lists from libdge.so (dge_knn_cosine: exact-f32 MFMA tiles fused with top-k on the MI355X).
    -> (idx int32 [n x k], dist float32 [n x k], kernel_ms)."""
    import ctypes as C
    from ._native import check, lib
    f = np.ascontiguousarray(features, np.float32)
    n, D = f.shape
    idx = np.empty((n, k), np.int32); dist = np.empty((n, k), np.float32); ms = C.c_double(0)
    check(lib.dge_knn_cosine(int(device), f.ctypes.data_as(C.c_void_p), n, D, int(k), idx.ctypes.data_as(C.c_void_p),
                             dist.ctypes.data_as(C.c_void_p), C.byref(ms)))
    return idx, dist, ms.value


def ndcg_against_gpu(features, gnd_features, k=10, device=0):
    """ndcg_against wholly on the device (dge_ndcg_at_k): both KNN passes on MFMA, the relevance look-ups and the DCG sums in a
    kernel.  -> (nDCG@k, kernel ms of the two KNN passes)."""
    import ctypes as C
    from ._native import check, lib
    f = np.ascontiguousarray(features, np.float32); g = np.ascontiguousarray(gnd_features, np.float32)
    if len(f) != len(g):
        raise ValueError("features and gnd_features describe different numbers of regions")
    out = C.c_double(0); ms = C.c_double(0)
    check(lib.dge_ndcg_at_k(int(device), f.ctypes.data_as(C.c_void_p), f.shape[1], g.ctypes.data_as(C.c_void_p), g.shape[1], len(f), int(k),
                            C.byref(out), C.byref(ms)))
    return out.value, ms.value
