"""How many of ONE row's updates may be in flight?  train_rows caps the workers at 48 / (the busiest row's share of the tokens): round 4 saw a rank^-1 vocabulary go to NaN at
1 130 in flight and took 48 without a sweep in between.  A static graph WITH structure and a heavy head: R regions in communities of 64, half of a vertex's flow stays inside its
community, the other half goes to a region drawn with P(rank r) ~ 1 / (r + 1) over ALL regions (the busiest region: ~4 % of all tokens).  Trained under auto's choice (atomics) with
the worker count forced to 48 ... 768 / share and to the device's fill; edges/s, link AUC and loss on held-out steps, next to the sequential oracle and its 8 Hogwild threads.
python scripts/hot_row_inflight.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from oracle import oracle as O
R, L, D, K = 100_000, 12, 128, 5
dev = "cuda:0"
g0 = torch.Generator(device=dev); g0.manual_seed(1)
deg = torch.randint(30, 70, (R,), generator=g0, device=dev)
Etot = int(deg.sum().item())
src = torch.repeat_interleave(torch.arange(R, device=dev, dtype=torch.int32), deg)
inside = torch.rand(Etot, generator=g0, device=dev) < 0.5
local = ((src // 64) * 64 + torch.randint(0, 64, (Etot,), generator=g0, device=dev, dtype=torch.int32)).clamp_(max=R - 1)
ur = torch.rand(Etot, generator=g0, device=dev, dtype=torch.float64)
pop = (torch.exp(ur * float(np.log(R + 1.0))) - 1.0).to(torch.int64).clamp_(0, R - 1).to(torch.int32)
dst = torch.where(inside, local, pop).to(torch.int32)
w = torch.ones(Etot, dtype=torch.float64, device=dev)
g = E.DeviceGraph(0); g.add_edges_device(src.contiguous(), dst.contiguous(), w); g.set_sources(np.arange(R, dtype=np.int32)); g.build_alias(False)
n = 800_000
walks = g.sample_walks(n, L, seed=5, rng_mode=1)
test = g.sample_walks(100_000, L, seed=99, rng_mode=1)
cnt = np.bincount(walks.reshape(-1)[walks.reshape(-1) >= 0], minlength=R)
share = cnt.max() / cnt.sum()
print("busiest row: %.4f of the tokens; 48 in flight = %d workers" % (share, int(48 / share)), flush=True)

def score(syn0, syn1, vid):
    remap = -np.ones(R, np.int64); remap[vid] = np.arange(len(vid))
    a = test[:, :-1].reshape(-1); b = test[:, 1:].reshape(-1)
    rb = np.random.default_rng(3).integers(0, R, len(b))
    a, b, rb = remap[a], remap[b], remap[rb]
    ok = (a >= 0) & (b >= 0) & (rb >= 0); a, b, rb = a[ok], b[ok], rb[ok]
    pos = (syn0[b].astype(np.float64) * syn1[a]).sum(1); neg = (syn0[rb].astype(np.float64) * syn1[a]).sum(1)
    return float((pos > neg).mean() + 0.5 * (pos == neg).mean()), float(np.log1p(np.exp(-pos)).mean() + np.log1p(np.exp(neg)).mean()), float(np.abs(syn0).max())

kw = dict(negative=K, min_count=2, epochs=1, seed=1, table_size=10_000_000)
t = time.time(); om = O.train_sgns(walks, R, D, L, arith=0, **kw); ts = time.time() - t
print("oracle sequential        : %.2e edges/s  AUC %.4f loss %.4f max|syn0| %.2f" % ((om.pairs / ts,) + score(om.syn0, om.syn1neg, om.vocab_ids)), flush=True)
t = time.time(); o8 = O.train_sgns(walks, R, D, L, arith=0, threads=8, **kw); ts = time.time() - t
print("oracle 8 Hogwild threads : %.2e edges/s  AUC %.4f loss %.4f max|syn0| %.2f" % ((o8.pairs / ts,) + score(o8.syn0, o8.syn1neg, o8.vocab_ids)), flush=True)
corpus = E.WalkCorpus.from_host(walks, 0)
counts = torch.zeros(R, dtype=torch.int64, device=dev); corpus.count_tokens(R, counts)
for inflight in (0, 24, 48, 96, 192, 384, 768, 10**9):
    workers = 0 if inflight == 0 else min(16384, max(64, int(inflight / share)))
    with E.tuning(**({"workers": workers} if workers else {})):
        m = E.SgnsModel.create(E.make_config(D, L, R, workers=0, **kw), counts, 0)
        m.train(corpus)
        st = m.stats(); sch = m.schedule()
    syn0, vid = m.vectors()
    print("gpu %-21s: %.2e edges/s (policy %d, %d workers) AUC %.4f loss %.4f max|syn0| %.2f" % (("auto" if not inflight else "%d in flight" % min(inflight, int(16384 * share)),
          st["pairs"] / (st["kernel_ms"] * 1e-3), sch["update_policy"], sch["workers"]) + score(syn0, m.syn1neg(), vid)), flush=True)
    m.close()
