"""Quality of GPU Hogwild vs the sequential oracle on a mid-size layered graph (run on the GPU box)."""
import sys, time, numpy as np
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

def auc(syn0, syn1, vid):
    remap = -np.ones(NV, np.int64); remap[vid] = np.arange(len(vid))
    a = remap[test[:, :-1].reshape(-1)]; b = remap[test[:, 1:].reshape(-1)]
    ok = (a >= 0) & (b >= 0); a, b = a[ok], b[ok]
    pos = (syn0[b] * syn1[a]).sum(1)
    rng = np.random.default_rng(0); rb = rng.integers(0, len(vid), len(a))
    neg = (syn0[rb] * syn1[a]).sum(1)
    # AUC by rank
    s = np.concatenate([pos, neg]); r = s.argsort().argsort()
    return (r[:len(pos)].mean() - (len(pos) - 1) / 2) / len(neg), float(pos.mean()), float(neg.mean())

t = time.time(); om = O.train_sgns(walks, NV, D, L, negative=K, threads=1, table_size=10_000_000, arith=0); t_or = time.time() - t
print("oracle seq: V", om.V, "pairs", om.pairs, "%.1fs" % t_or, "auc/pos/neg", auc(om.syn0, om.syn1neg, om.vocab_ids), flush=True)
o8 = O.train_sgns(walks, NV, D, L, negative=K, threads=8, table_size=10_000_000, arith=0)
print("oracle 8thr: auc", auc(o8.syn0, o8.syn1neg, o8.vocab_ids), "median cos vs seq", float(np.median(cosine_rows(o8.syn0, om.syn0))), flush=True)
for workers, pol in ((0, 3), (0, 1), (0, 2), (0, 0), (1024, 5), (1024, 6), (64, 5)):
    cfg = E.make_config(D, L, NV, negative=K, workers=workers, table_size=10_000_000); cfg.update_policy = pol
    t = time.time(); dm = E.SgnsModel.fit(walks, cfg, 0); syn0, vid = dm.vectors(); dt = time.time() - t
    st = dm.stats()
    print("gpu workers", workers, "pol", pol, "pairs", st["pairs"], "kernel_ms %.1f" % st["kernel_ms"], "auc/pos/neg", auc(syn0, dm.syn1neg(), vid),
          "median cos vs seq", float(np.median(cosine_rows(syn0, om.syn0))), flush=True)
