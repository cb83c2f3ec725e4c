"""Is the auto rule of dge_train_config.update_policy (embedding_amd/csrc/sgns.hip: train_rows) the fastest choice AWAY from the four bench graphs it was
fitted on?  Synthetic walk corpora over V vocabulary rows whose popularity follows rank^-s (s = 0: flat, 0.5, 1.0: Zipf), L = W = 24, K = 5, D in {64, 128, 256};
one launch per policy (auto, 2 = atomics, 5 = commit locks, 7 = locks + head by atomics, 8 = owner-computes); a forced policy that a short probe shows to be more
than 4x slower than the best so far is not run at full length (commit locks on a Zipf head spin for minutes).
Prints a table and, per configuration, auto's rate over the best forced policy's.   python scripts/policy_sweep.py [quick]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E

QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
dev = "cuda:0"
L, K = 24, 5
Vs = (50_000, 300_000, 1_000_000) if QUICK else (50_000, 200_000, 300_000, 1_000_000)
Ss = (0.0, 1.0) if QUICK else (0.0, 0.5, 1.0)
Ds = (128,) if QUICK else (64, 128, 256)
N_WALKS, N_PROBE = 200_000, 8_000


def corpus_for(V, s, n, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    if s == 0.0:
        ids = torch.randint(0, V, (n, L), generator=g, device=dev, dtype=torch.int32)
    else:
        p = torch.arange(1, V + 1, device=dev, dtype=torch.float64).pow(-s)
        cdf = torch.cumsum(p / p.sum(), 0)
        u = torch.rand(n * L, generator=g, device=dev, dtype=torch.float64)
        ids = torch.searchsorted(cdf, u).clamp_(max=V - 1).to(torch.int32).view(n, L)
    return E.WalkCorpus.from_host(ids.cpu().numpy(), 0)


def rate(cfg, counts, corpus, n):
    m = E.SgnsModel.create(cfg, counts, 0)
    m.train(corpus, 0, min(n, 2000))              # (first launch: work buffers, lazy allocations)
    m.reset_stats()
    m.train(corpus, 0, n)
    st, sch = m.stats(), m.schedule()
    ok = bool(np.isfinite(m.vectors()[0][:1000]).all())
    m.close()
    return st["pairs"] / (st["kernel_ms"] * 1e-3), sch, ok


worst = 1e9
print("%9s %4s %4s | %-28s | %s" % ("V", "s", "D", "auto ran as", "edges/s: auto, then forced 2 / 5 / 7 / 8 (- = skipped after the probe, x = refused, nan = diverged)"))
for V in Vs:
    for s in Ss:
        corpus = corpus_for(V, s, N_WALKS, 7)
        counts = torch.zeros(V, dtype=torch.int64, device=dev); corpus.count_tokens(V, counts)
        for D in Ds:
            res = {}
            cfg = lambda pol: E.make_config(D, L, V, negative=K, workers=0, epochs=1, seed=1, update_policy=pol, min_count=1)
            res[0], sch0, ok = rate(cfg(0), counts, corpus, N_WALKS)
            assert ok
            best = res[0]
            for pol in (2, 5, 7, 8):
                try:
                    pr, _, _ = rate(cfg(pol), counts, corpus, N_PROBE)
                    if pr * 4 < best:
                        res[pol] = None; continue
                    res[pol], _, ok = rate(cfg(pol), counts, corpus, N_WALKS)
                    if not ok:
                        res[pol] = "nan"; continue          # a FORCED policy outside its regime (owner-computes on a Zipf head: DESIGN.md section 5.7) may diverge
                    best = max(best, res[pol])
                except E.DgeError:
                    res[pol] = "x"
            forced = [v for k, v in res.items() if k and isinstance(v, float)]
            ratio = res[0] / max(forced) if forced else 1.0
            worst = min(worst, ratio)
            fmt = lambda v: "   -    " if v is None else ("   x    " if v == "x" else ("  nan   " if v == "nan" else "%.2e" % v))
            print("%9d %4.1f %4d | %-28s | %s  %s  -> auto / best forced = %.2f" % (V, s, D, "policy %d, %d workers, head %d" % (sch0["update_policy"], sch0["workers"], sch0["hot_rows"]),
                                                                                 fmt(res[0]), " / ".join(fmt(res[p]) for p in (2, 5, 7, 8)), ratio), flush=True)
        corpus.close()
print("worst auto / best forced: %.2f" % worst)
