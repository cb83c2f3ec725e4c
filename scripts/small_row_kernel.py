"""The small-row trainer (k_sgns_train_small: 32 lanes a worker, a row = one request) against k_sgns_train's 16-lane groups on the reference's tract configuration
(6 408 rows, D = 20, K = 5, L = W = 8; the structured graph of scripts/small_vocab_workers.py): edges/s, held-out link AUC and loss, next to the sequential oracle.
python scripts/small_row_kernel.py"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from oracle import oracle as O
R, T, L, K = 801, 8, 8, 5
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
    return float((pos > neg).mean() + 0.5 * (pos == neg).mean()), float(np.log1p(np.exp(-pos)).mean() + np.log1p(np.exp(neg)).mean())

corpus = E.WalkCorpus.from_host(walks, 0)
import torch
counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
for D in (20, 32, 17):
    kw = dict(negative=K, min_count=2, epochs=1, seed=1, table_size=10_000_000)
    t = time.time(); om = O.train_sgns(walks, NV, D, L, arith=0, **kw); ts = time.time() - t
    print("D = %d  oracle sequential           : %.2e edges/s  AUC %.4f loss %.4f" % ((D, om.pairs / ts) + score(om.syn0, om.syn1neg, om.vocab_ids)), flush=True)
    for small in (0, 1):
        for workers in (0, 3204, 4806, 6408, 9612, 16384):
            res = []
            for rep in range(2):
                knobs = {"small_rows": small}
                if workers: knobs["workers"] = workers
                with E.tuning(**knobs):
                    m = E.SgnsModel.create(E.make_config(D, L, NV, workers=0, **kw), counts, 0)
                    m.train(corpus)
                    st = m.stats(); sch = m.schedule(); kn = m.kernel()
                syn0, vid = m.vectors()
                res.append((st["pairs"] / (st["kernel_ms"] * 1e-3), sch["workers"]) + score(syn0, m.syn1neg(), vid))
                m.close()
            print("D = %d  %-46s %-14s: %s" % (D, kn, "auto" if not workers else "%d workers" % workers, "  |  ".join("%.2e edges/s (%d workers) AUC %.4f loss %.4f" % r for r in res)), flush=True)
