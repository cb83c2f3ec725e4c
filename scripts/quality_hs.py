"""Hierarchical softmax: quality of the device-filling schedule vs the sequential oracle (run on the GPU box).
Usage: python scripts/quality_hs.py [R]   (the drain period is set through dge_set_tuning)"""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from embedding_amd import synth
from oracle import oracle as O
from helpers import cosine_rows

R, T, L, D, K = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 8, 8, 64, 5
NV = R * T
G = synth.flow_graph_numpy(R, T, 20, seed=3)
g = E.DeviceGraph(0); g.add_edges(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(False)
n = 10 * NV
walks = g.sample_walks(n, L, seed=5)
test = g.sample_walks(20000, L, seed=77)          # held-out walks


def auc_rank(pos, neg):
    s = np.concatenate([pos, neg]); r = s.argsort().argsort()
    return float((r[:len(pos)].mean() - (len(pos) - 1) / 2) / len(neg))


def metrics(syn0, syn1neg, vid):
    remap = -np.ones(NV, np.int64); remap[vid] = np.arange(len(vid))
    a = remap[test[:, :-1].reshape(-1)]; b = remap[test[:, 1:].reshape(-1)]
    ok = (a >= 0) & (b >= 0); a, b = a[ok], b[ok]
    rb = np.random.default_rng(0).integers(0, len(vid), len(a))
    io = auc_rank((syn0[b] * syn1neg[a]).sum(1), (syn0[rb] * syn1neg[a]).sum(1)) if syn1neg is not None else float("nan")
    # syn0-only: two vertices that follow the same vertex in held-out walks vs a random vertex
    n0 = syn0 / np.maximum(np.linalg.norm(syn0, axis=1, keepdims=True), 1e-12)
    order = np.argsort(a, kind="stable"); a2, b2 = a[order], b[order]
    same = a2[1:] == a2[:-1]
    x, y = b2[1:][same], b2[:-1][same]
    ry = np.random.default_rng(1).integers(0, len(vid), len(x))
    sim = auc_rank((n0[x] * n0[y]).sum(1), (n0[x] * n0[ry]).sum(1))
    return "auc_io %.4f auc_sim %.4f" % (io, sim)


t = time.time(); om = O.train_sgns(walks, NV, D, L, negative=K, threads=1, table_size=10_000_000, arith=0, use_hs=True)
print("oracle HS seq: V", om.V, "pairs", om.pairs, "%.1fs" % (time.time() - t), metrics(om.syn0, om.syn1neg, om.vocab_ids), flush=True)
o8 = O.train_sgns(walks, NV, D, L, negative=K, threads=8, table_size=10_000_000, arith=0, use_hs=True)
print("oracle HS 8thr:", metrics(o8.syn0, o8.syn1neg, o8.vocab_ids), "median cos vs seq %.3f" % float(np.median(cosine_rows(o8.syn0, om.syn0))), flush=True)
ons = O.train_sgns(walks, NV, D, L, negative=K, threads=8, table_size=10_000_000, arith=0)
print("oracle NS-only 8thr:", metrics(ons.syn0, ons.syn1neg, ons.vocab_ids), flush=True)
# (drain period of the LDS accumulators near the root, cold = inner nodes updated by plain read-modify-write instead of atomics: -1 = the
#  library's rule (nodes on < 2e-5 of the paths), 0 = none, large = every node outside the LDS accumulators)
for workers, load, cold in ((0, "64", -1), (0, "64", 0), (0, "64", 1 << 30), (0, "1", -1), (0, "16", -1), (0, "256", -1), (0, "1024", -1), (1024, "64", -1), (64, "64", -1)):
    with E.tuning(hs_drain=int(load), **({"hs_cold": cold} if cold >= 0 else {})):
        cfg = E.make_config(D, L, NV, negative=K, workers=workers, table_size=10_000_000, use_hs=True)
        dm = E.SgnsModel.fit(walks, cfg, 0); syn0, vid = dm.vectors(); st = dm.stats()
    print("gpu HS workers", workers, "drain", load, "cold", cold, "pairs", st["pairs"], "kernel_ms %.1f" % st["kernel_ms"], metrics(syn0, dm.syn1neg(), vid),
          "median cos vs seq %.3f" % float(np.median(cosine_rows(syn0, om.syn0))), flush=True)
