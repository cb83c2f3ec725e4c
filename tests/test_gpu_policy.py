"""GPU: the auto rule of dge_train_config.update_policy (embedding_amd/csrc/sgns.hip: auto_policy) away from the bench graphs it was fitted on — four points of
scripts/policy_sweep.py's grid (the whole table: profiles/r04_policy_sweep.txt), each a regime in which round 3's rule was wrong or unsafe.  Synthetic corpora, V
vocabulary rows whose popularity follows rank^-s, L = W = 24, K = 5; one launch per policy.  Asserted: auto stays finite and runs at >= 0.8 of the fastest forced
policy that stays finite.  (w2v.fit(), J/DeepWalk.java:79; the schedules are new.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L, K, N_WALKS = 24, 5, 100_000


def _corpus(dge, V, s, seed=7):
    import torch
    g = torch.Generator(device="cuda:0"); g.manual_seed(seed)
    if s == 0.0:
        ids = torch.randint(0, V, (N_WALKS, L), generator=g, device="cuda:0", dtype=torch.int32)
    else:
        p = torch.arange(1, V + 1, device="cuda:0", dtype=torch.float64).pow(-s)
        cdf = torch.cumsum(p / p.sum(), 0)
        u = torch.rand(N_WALKS * L, generator=g, device="cuda:0", dtype=torch.float64)
        ids = torch.searchsorted(cdf, u).clamp_(max=V - 1).to(torch.int32).view(N_WALKS, L)
    return dge.WalkCorpus.from_host(ids.cpu().numpy(), 0)


def _rate(dge, cfg, counts, corpus, n):
    m = dge.SgnsModel.create(cfg, counts, 0)
    m.train(corpus, 0, min(n, 2000))
    m.reset_stats()
    m.train(corpus, 0, n)
    st, sch = m.stats(), m.schedule()
    ok = bool(np.isfinite(m.vectors()[0][:2000]).all() and np.isfinite(m.syn1neg()[:2000]).all())
    m.close()
    return st["pairs"] / (st["kernel_ms"] * 1e-3), sch, ok


@pytest.mark.parametrize("V,s,D,expect", [
    (200_000, 0.0, 128, 5),      # a flat vocabulary below round 3's 262 144-row bar: commit locks beat owner-computes by a fifth
    (300_000, 0.5, 64, 7),       # a head between an eighth and a quarter of the rows: locks on the tail beat owner-computes by a quarter
    (50_000, 0.5, 256, 8),       # owner-computes 8 % below its old 1e6-item bar, on wide rows: 1.4x the atomics it was left with
    (300_000, 1.0, 128, 2),      # one row with 9 % of the tokens: device-filling Hogwild diverged (NaN); now at most 96 of a row's updates in flight (round 5: swept on a graph with structure), by atomics
])
def test_auto_policy_is_fast_and_finite_off_the_bench_graphs(dge, V, s, D, expect):
    import torch
    corpus = _corpus(dge, V, s)
    counts = torch.zeros(V, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(V, counts)
    cfg = lambda pol: dge.make_config(D, L, V, negative=K, workers=0, epochs=1, seed=1, update_policy=pol, min_count=1)
    auto, sch, ok = _rate(dge, cfg(0), counts, corpus, N_WALKS)
    assert ok and sch["update_policy"] == expect, sch
    best = {}
    for pol in (2, 5, 7, 8):
        try:
            r, _, fin = _rate(dge, cfg(pol), counts, corpus, N_WALKS)
        except dge.DgeError as e:                            # a forced schedule outside its regime is refused (round 5), it neither diverges nor spins
            assert e.code == 1 and pol in (5, 8), (pol, str(e))
            continue
        assert fin, pol                                      # ... so whatever runs stays finite
        best[pol] = r
    print("\n[policy V=%d s=%.1f D=%d] auto %.3e (%s) | forced %s" % (V, s, D, auto, sch, {k: "%.3e" % v for k, v in best.items()}), flush=True)
    assert auto >= 0.8 * max(best.values()), (auto, best, sch)


def test_forced_schedules_are_refused_where_they_would_diverge_or_spin(dge):
    """A host that forces a schedule (dge_train_config.update_policy) outside its regime gets DGE_ERR_ARG with the reason, at once: owner-computes on a vocabulary whose busiest
    row would take tens of thousands of terms of one synchronous mini-batch (round 4: NaN with status 0), commit locks on every row where the busiest row's pairs queue
    behind one lock (round 4: a launch of minutes).  Rank^-1 popularity over 300 000 words: 9 % of the tokens on one row."""
    import time
    import torch
    V = 300_000
    corpus = _corpus(dge, V, 1.0)
    counts = torch.zeros(V, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(V, counts)
    for pol, word in ((8, "mini-batch"), (5, "row lock"), (6, "row lock")):
        m = dge.SgnsModel.create(dge.make_config(128, L, V, negative=K, workers=0, epochs=1, seed=1, update_policy=pol, min_count=1), counts, 0)
        t = time.perf_counter()
        with pytest.raises(dge.DgeError) as e:
            m.train(corpus, 0, N_WALKS)
        assert e.value.code == 1 and word in str(e.value) and "update_policy 0" in str(e.value), str(e.value)
        assert time.perf_counter() - t < 1.0
        assert m.stats()["pairs"] == 0                       # nothing was launched
        m.close()
    # the same policies where they belong still run: a flat vocabulary
    flat = _corpus(dge, V, 0.0)
    c2 = torch.zeros(V, dtype=torch.int64, device="cuda:0"); flat.count_tokens(V, c2)
    for pol in (5, 8):
        m = dge.SgnsModel.create(dge.make_config(128, L, V, negative=K, workers=0, epochs=1, seed=1, update_policy=pol, min_count=1), c2, 0)
        m.train(flat, 0, 20_000)
        assert m.stats()["pairs"] > 0 and np.isfinite(m.vectors()[0][:1000]).all()
        m.close()


def test_watchdog_ends_a_launch_that_waits_for_locks(dge):
    """The safety net behind the refusal: with the refusal switched off (DGE_TUNE_ALLOW_UNSAFE) and the lock kernel's watchdog at 300 ms, commit locks on the rank^-1
    vocabulary — a launch that would spin for minutes — ends within seconds, and the first call that looks at the device reports DGE_ERR_STATE (once)."""
    import time
    import torch
    V = 300_000
    corpus = _corpus(dge, V, 1.0)
    counts = torch.zeros(V, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(V, counts)
    m = dge.SgnsModel.create(dge.make_config(128, L, V, negative=K, workers=0, epochs=1, seed=1, update_policy=5, min_count=1), counts, 0)
    with dge.tuning(allow_unsafe=1, watchdog_ms=300):
        t = time.perf_counter()
        m.train(corpus, 0, N_WALKS)
        with pytest.raises(dge.DgeError) as e:
            m.stats()
        took = time.perf_counter() - t
    assert e.value.code == 5 and "watchdog" in str(e.value), str(e.value)
    assert took < 5.0, took
    st = m.stats()                                           # reported once; the model is usable (its tables hold the part of the launch that was trained)
    assert 0 < st["pairs"] < 0.5 * N_WALKS * 380
    m.close()
