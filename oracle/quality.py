"""TEST INFRASTRUCTURE (like the rest of oracle/): the quality metric of the reference's evaluation scripts restated on the host in float64
numpy — the checker for embedding_amd.evaluate's device kernels (dge_knn_cosine, dge_ndcg_at_k).  Only tests/ and scripts/ import it; the product
never does (tests/test_abi.py).

pairwise_estimator  P/embeddingEvaluation_tract.py:169-196 — per region the other regions sorted by cosine DISTANCE
                    (scipy's cosine = 1 - cos; NaN -> 2), ascending: the KNN lists the reference ranks with.
ndcg_at_k           P/embeddingEvaluation_tract.py:249-260 — relevance of neighbour j of region r = 1 - gnd_dist[r][j],
                    DCG = sum_i relv_i / log2(i+1), normalised by the DCG of the ground truth's own ordering.
"""
import numpy as np


def cosine_distance_matrix(features):
    f = np.asarray(features, np.float64)
    n = np.linalg.norm(f, axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = 1.0 - (f @ f.T) / np.outer(n, n)
    d[~np.isfinite(d)] = 2.0                       # np.isnan(c) -> 2   (:188-189)
    return d


def pairwise_estimator(features, rids):
    """-> (estimates {rid: [(rid2, dist), ...] ascending}, neighbors {rid: [rid2, ...]}) as the reference returns them."""
    rids = list(rids)
    d = cosine_distance_matrix(features)
    estimates, neighbors = {}, {}
    for i, k in enumerate(rids):
        order = [j for j in np.argsort(d[i], kind="stable") if rids[j] != k]
        estimates[k] = [(rids[j], float(d[i, j])) for j in order]
        neighbors[k] = [rids[j] for j in order]
    return estimates, neighbors


def dcg_at_k(k, gnd_est, neighbors):
    relv = [1.0 - gnd_est[neighbors[i]] for i in range(k)]
    return float(np.sum([relv[i - 1] / np.log2(i + 1) for i in range(1, len(relv) + 1)]))


def ndcg_at_k(k, testing, neighbors, gnd_est, dcg_max):
    total = 0.0
    for rid in testing:
        total += dcg_at_k(k, gnd_est[rid], neighbors[rid]) / dcg_max[rid]
    return total / len(testing)


def ndcg_against(features, gnd_features, rids, k=10):
    """nDCG@k of the KNN lists of `features` against the distances of `gnd_features` (same regions, same order)."""
    _, nb = pairwise_estimator(features, rids)
    gest, gnb = pairwise_estimator(gnd_features, rids)
    gnd = {r: dict(v) for r, v in gest.items()}
    dcg_max = {r: dcg_at_k(k, gnd[r], gnb[r]) for r in rids}
    return ndcg_at_k(k, list(rids), nb, gnd, dcg_max)
