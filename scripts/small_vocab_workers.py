"""How many concurrent walks may a SMALL vocabulary take?  train_rows caps the workers at half the vocabulary rows (Hogwild's premise: sparse collisions; measured in round 1 as
cosine >= 0.99 to the in-order result on a 2.3 k-row table).  On the reference's own tract configuration — 801 regions x 8 slices = 6 408 rows, D = 20, K = 5, L = W = 8
(J/DeepWalk.java:62-66,89-104) — that is 3 204 workers on a device that holds 16 384, each pair a latency chain (table look-up -> rows -> atomics): 7e8 edges/s, far from any
bandwidth.  This script trains a tract-sized graph WITH structure (communities of 9 regions, 80 % of a vertex's flow stays inside) with 3 204 ... 16 384 workers and reports
edges/s, the link-prediction AUC on held-out walk steps and the held-out loss, next to the sequential oracle's and the CPU's 8 Hogwild threads (the reference's own schedule).
python scripts/small_vocab_workers.py [hs]"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from oracle import oracle as O
HS = "hs" in sys.argv[1:]
R, T, L, D, K = 801, 8, 8, 20, 5
NV = R * T
rng = np.random.default_rng(1)
src, dst, w = [], [], []
for h in range(T):
    for s in range(R):
        k = int(rng.integers(60, 200))
        inside = rng.random(k) < 0.8
        d = np.where(inside, (s // 9) * 9 + rng.integers(0, 9, k), rng.integers(0, R, k)).clip(max=R - 1)
        src += [h * R + s] * k; dst += list(((h + 1) % T) * R + d); w += list(1.0 + np.floor(-20.0 * np.log(rng.random(k).clip(1e-12))))
g = E.DeviceGraph(0); g.add_edges(np.array(src, np.int32), np.array(dst, np.int32), np.array(w)); g.set_sources(np.arange(R, dtype=np.int32)); g.build_alias(False)
n = 1_500_000
walks = g.sample_walks(n, L, seed=5, rng_mode=1)
test = g.sample_walks(100_000, L, seed=99, rng_mode=1)

def score(syn0, syn1, vid):
    remap = -np.ones(NV, np.int64); remap[vid] = np.arange(len(vid))
    a = test[:, :-1].reshape(-1); b = test[:, 1:].reshape(-1)
    r2 = np.random.default_rng(3); rb = (b // R) * R + r2.integers(0, R, len(b))
    a, b, rb = remap[a], remap[b], remap[rb]
    ok = (a >= 0) & (b >= 0) & (rb >= 0); a, b, rb = a[ok], b[ok], rb[ok]
    pos = (syn0[b].astype(np.float64) * syn1[a]).sum(1); neg = (syn0[rb].astype(np.float64) * syn1[a]).sum(1)
    auc = float((pos > neg).mean() + 0.5 * (pos == neg).mean())
    loss = float(np.log1p(np.exp(-pos)).mean() + np.log1p(np.exp(neg)).mean())
    return auc, loss

kw = dict(negative=K, min_count=2, epochs=1, seed=1, table_size=10_000_000, use_hs=HS)
t = time.time(); om = O.train_sgns(walks, NV, D, L, arith=0, **kw); ts = time.time() - t
print("oracle sequential            : %.2e edges/s  AUC %.4f loss %.4f" % ((om.pairs / ts,) + score(om.syn0, om.syn1neg, om.vocab_ids)), flush=True)
t = time.time(); o8 = O.train_sgns(walks, NV, D, L, arith=0, threads=8, **kw); ts = time.time() - t
print("oracle 8 Hogwild threads     : %.2e edges/s  AUC %.4f loss %.4f" % ((o8.pairs / ts,) + score(o8.syn0, o8.syn1neg, o8.vocab_ids)), flush=True)
corpus = E.WalkCorpus.from_host(walks, 0)
import torch
counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
for workers in (0, 1602, 3204, 6408, 9612, 12816, 16384):
    res = []
    for rep in range(2):
        with E.tuning(**({"workers": workers} if workers else {})):
            m = E.SgnsModel.create(E.make_config(D, L, NV, workers=0, **kw), counts, 0)
            m.train(corpus)
            st = m.stats(); sch = m.schedule()
        syn0, vid = m.vectors()
        res.append((st["pairs"] / (st["kernel_ms"] * 1e-3), sch["workers"]) + score(syn0, m.syn1neg(), vid))
        m.close()
    print("gpu %-24s: %s" % ("auto" if not workers else "%d workers" % workers, "  |  ".join("%.2e edges/s (%d workers) AUC %.4f loss %.4f" % r for r in res)), flush=True)
