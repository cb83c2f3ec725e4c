"""Statistical parity at BASELINE's full size (cfg3 shape: 1 000 008 vertices, ~100 M edges, D=128, K=5, L=W=24):
one epoch (10 M walks, 3.8e9 pairs) under each update policy on a graph WITH structure (regions form communities of 64;
80 % of a vertex's flow stays inside its community), then link prediction on held-out walk steps:
AUC of sigma(syn0[next] . syn1neg[current]) for true next-steps against random vertices of the same slice.
Policy 2 (float atomics) loses no update and is the yardstick; 5/6 are the commit-lock kernels; 1 and 3 are shown for contrast."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E

# "zipf": the same graph, but a vertex's flow that LEAVES its community goes to a region drawn with P(rank r) ~ 1/(r+1)
# (popular regions, as in cfg5) instead of a uniform one -> a skewed vocabulary, where auto selects the mixed policy 7.
ZIPF = "zipf" in sys.argv[1:]          # ("hs zipf": the tree term on the skewed graph)
# "blocks N": the multi-GPU block schedule with N ranks simulated on this GPU (N models), 10 global batches per epoch,
# against the one-GPU run of the same epoch.
BLOCKS = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "blocks" else 0
# "hs": hierarchical softmax + 5 negatives (what DL4J's builder default trains): every inner node under atomics, and with the cold end of
# the tree updated by plain read-modify-write (the library's rule), one epoch each
HS = "hs" in sys.argv[1:]

R, T, L, D, K = 41667, 24, 24, 128, 5
NV = R * T
dev = "cuda:0"
g0 = torch.Generator(device=dev); g0.manual_seed(1)
deg = torch.exp(np.log(100) - 0.5 + torch.randn(NV, generator=g0, device=dev)).to(torch.int64).clamp_(1, R)
Etot = int(deg.sum().item())
src = torch.repeat_interleave(torch.arange(NV, device=dev, dtype=torch.int32), deg)
reg = src % R
inside = torch.rand(Etot, generator=g0, device=dev) < 0.8
local = (reg // 64) * 64 + torch.randint(0, 64, (Etot,), generator=g0, device=dev, dtype=torch.int32)
anyw = torch.randint(0, R, (Etot,), generator=g0, device=dev, dtype=torch.int32)
if ZIPF:
    ur = torch.rand(Etot, generator=g0, device=dev, dtype=torch.float32)
    anyw = (torch.exp(ur * float(np.log(R + 1.0))) - 1.0).to(torch.int64).clamp_(0, R - 1).to(torch.int32)
    del ur
dreg = torch.where(inside, local.clamp_(max=R - 1), anyw)
dst = (((src // R + 1) % T) * R + dreg).to(torch.int32)
w = (1.0 + torch.floor(-20.0 * torch.log(torch.rand(Etot, generator=g0, device=dev, dtype=torch.float64).clamp_(min=1e-12))))
g = E.DeviceGraph(0); g.add_edges_device(src.contiguous(), dst.contiguous(), w.contiguous()); del src, dst, w, reg, inside, local, anyw, dreg
g.set_sources(np.arange(R, dtype=np.int32)); g.build_alias(False)
n = 10 * NV
corpus = g.sample_walks_device(n, L, seed=5)
test = torch.from_numpy(g.sample_walks(200_000, L, seed=99)).to(dev).to(torch.int64)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)

def auc(m):
    syn0, vid = m.vectors(); syn1 = m.syn1neg()
    s0 = torch.from_numpy(syn0).to(dev); s1 = torch.from_numpy(syn1).to(dev)
    remap = -torch.ones(NV, dtype=torch.int64, device=dev); remap[torch.from_numpy(vid.astype(np.int64)).to(dev)] = torch.arange(len(vid), device=dev)
    a = remap[test[:, :-1].reshape(-1)]; b = remap[test[:, 1:].reshape(-1)]
    ok = (a >= 0) & (b >= 0); a, b = a[ok], b[ok]
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    # negatives: a random region in the SAME slice as the true next vertex
    vb = torch.from_numpy(vid.astype(np.int64)).to(dev)[b]
    rb = remap[(vb // R) * R + torch.randint(0, R, (len(b),), generator=gen, device=dev)]
    ok2 = rb >= 0; a, b, rb = a[ok2], b[ok2], rb[ok2]
    pos = (s0[b] * s1[a]).sum(1); neg = (s0[rb] * s1[a]).sum(1)
    return float((pos > neg).float().mean() + 0.5 * (pos == neg).float().mean()), float(pos.mean()), float(neg.mean())

if BLOCKS:
    sys.path.insert(0, 'tests')
    from helpers import simulate_block_schedule, simulate_gather_syn0
    cfg = E.make_config(D, L, NV, negative=K, workers=0)
    m = E.SgnsModel.create(cfg, counts, 0)
    nb = n // 10
    wb = 0
    for b in range(10):
        m.reset_stats(); m.train(corpus, b * nb, nb, walk_index_base=b * nb, words_before=wb, total_walks=n); wb += m.stats()["words"]
    print("one GPU, 10 batches          | AUC %.4f pos %.3f neg %.3f" % auc(m), "| ran as", m.schedule(), flush=True)
    m.close()
    ms = [E.SgnsModel.create(cfg, counts, 0) for _ in range(BLOCKS)]
    wb = 0; t = time.time()
    for b in range(10):
        for mm in ms:
            mm.reset_stats()
        simulate_block_schedule(ms, lambda mm: mm.train(corpus, b * nb, nb, walk_index_base=b * nb, words_before=wb, total_walks=n))
        wb += ms[0].stats()["words"]
    simulate_gather_syn0(ms)
    print("%d ranks, block schedule      | AUC %.4f pos %.3f neg %.3f" % ((BLOCKS,) + auc(ms[0])), "| ran as", ms[0].schedule(), "| %.0f s for all ranks on one GPU" % (time.time() - t), flush=True)
    sys.exit(0)

if HS:
    import os
    variants = [("pair by pair, cold end plain, atomics through the atomics wave (round 3)", {"hs_centre": 0}), ("a wave per centre, negatives by atomics", {"hs_centre": 1}), ("a wave per centre, negatives under commit locks, three waves a workgroup", {"hs_centre": 2}), ("a wave per centre, negatives under commit locks, seven waves a workgroup", {"hs_centre": 3}), ("default", {})]
    if os.environ.get("DGE_HS_DRAINS"):          # e.g. DGE_HS_DRAINS=6,8: only the wave-per-centre kernel with these drain periods of the LDS accumulators
        variants = [("a wave per centre, LDS accumulators drained every %s centre additions" % d, {"hs_drain": int(d)}) for d in os.environ["DGE_HS_DRAINS"].split(",")]
    if os.environ.get("DGE_HS_VARIANTS"):        # e.g. DGE_HS_VARIANTS="hs_centre=3,hs_hot_kb=30,hs_drain=8;hs_centre=2,hs_drain=8": these knob sets only
        variants = [(v, {k: int(x) for k, x in (kv.split("=") for kv in v.split(","))}) for v in os.environ["DGE_HS_VARIANTS"].split(";")]
    for name, knobs in variants:
        with E.tuning(**knobs):
            m = E.SgnsModel.create(E.make_config(D, L, NV, negative=K, workers=0, use_hs=True), counts, 0)
            m.train(corpus); st = m.stats()
        print("HS, %s | pairs %.3e kernel %.1f s -> %.3e edges/s" % (name, st["pairs"], st["kernel_ms"] / 1e3, st["pairs"] / (st["kernel_ms"] / 1e3)),
              "| AUC %.4f pos %.3f neg %.3f" % auc(m), flush=True)
        m.close()
    sys.exit(0)

import os
POLICIES = tuple(int(x) for x in os.environ["DGE_QUALITY_POLICIES"].split(",")) if os.environ.get("DGE_QUALITY_POLICIES") else ((2, 7, 0) if ZIPF else (0, 2, 5, 6, 1, 3))
for kv in filter(None, os.environ.get("DGE_QUALITY_TUNE", "").split(",")):      # e.g. DGE_QUALITY_TUNE=table_runs=0
    k, v = kv.split("="); E._native.check(E.lib.dge_set_tuning(E.engine.TUNING_KNOBS[k], int(v)))
for pol in POLICIES:
    cfg = E.make_config(D, L, NV, negative=K, workers=0, update_policy=pol)
    m = E.SgnsModel.create(cfg, counts, 0)
    t = time.time(); m.train(corpus); st = m.stats(); dt = time.time() - t
    print("policy", pol, "pairs %.3e" % st["pairs"], "kernel %.2fs" % (st["kernel_ms"] / 1e3), "-> %.3e edges/s" % (st["pairs"] / (st["kernel_ms"] / 1e3)),
          "| AUC %.4f pos %.3f neg %.3f" % auc(m), "| ran as", m.schedule(), flush=True)
    m.close()

if ZIPF and len(sys.argv) > 2 and sys.argv[2] == "sweep":       # head size of the mixed policy (dge_set_tuning overrides the rule)
    import os
    for hot in (1000, 3000, 7000, 30000, 100000):
        E.lib.dge_set_tuning(0, hot)                 # DGE_TUNE_HOT_ROWS
        m = E.SgnsModel.create(E.make_config(D, L, NV, negative=K, workers=0, update_policy=7), counts, 0)
        m.train(corpus); st = m.stats()
        print("policy 7 head", hot, "-> %.3e edges/s" % (st["pairs"] / (st["kernel_ms"] / 1e3)), "| AUC %.4f" % auc(m)[0], flush=True)
        m.close()

