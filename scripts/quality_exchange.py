"""Multi-GPU exchange rule, chosen by quality ("averaging; sum-with-LR-scaling is the alternative — choose
by quality").  N ranks are simulated on ONE GPU with N models that start from the same weights; each trains its walk
shard in `steps` slices and after every slice the deltas are summed (what the RCCL all-reduce yields) and applied with
`scale`: 1/N = model averaging, 1 = every rank's updates count in full (what one GPU would do with all the walks,
stale by one slice).  Link-prediction AUC as in scripts/quality_scale.py, one epoch."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E

R, T, L, D, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 12500), 24, 24, 64, 5
NV = R * T
dev = "cuda:0"
g0 = torch.Generator(device=dev); g0.manual_seed(1)
deg = torch.exp(np.log(50) - 0.5 + torch.randn(NV, generator=g0, device=dev)).to(torch.int64).clamp_(1, R)
Etot = int(deg.sum().item())
src = torch.repeat_interleave(torch.arange(NV, device=dev, dtype=torch.int32), deg)
reg = src % R
inside = torch.rand(Etot, generator=g0, device=dev) < 0.8
local = (reg // 64) * 64 + torch.randint(0, 64, (Etot,), generator=g0, device=dev, dtype=torch.int32)
anyw = torch.randint(0, R, (Etot,), generator=g0, device=dev, dtype=torch.int32)
dreg = torch.where(inside, local.clamp_(max=R - 1), anyw)
dst = (((src // R + 1) % T) * R + dreg).to(torch.int32)
w = (1.0 + torch.floor(-20.0 * torch.log(torch.rand(Etot, generator=g0, device=dev, dtype=torch.float64).clamp_(min=1e-12))))
g = E.DeviceGraph(0); g.add_edges_device(src.contiguous(), dst.contiguous(), w.contiguous()); del src, dst, w, reg, inside, local, anyw, dreg
g.set_sources(np.arange(R, dtype=np.int32)); g.build_alias(False)
n = 10 * NV
corpus = g.sample_walks_device(n, L, seed=5)
test = torch.from_numpy(g.sample_walks(100_000, L, seed=99)).to(dev).to(torch.int64)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
print("graph: %d vertices, %d edges, %d walks" % (NV, Etot, n), flush=True)


def auc(m):
    syn0, vid = m.vectors(); syn1 = m.syn1neg()
    s0 = torch.from_numpy(syn0).to(dev); s1 = torch.from_numpy(syn1).to(dev)
    vt = torch.from_numpy(vid.astype(np.int64)).to(dev)
    remap = -torch.ones(NV, dtype=torch.int64, device=dev); remap[vt] = torch.arange(len(vid), device=dev)
    a = remap[test[:, :-1].reshape(-1)]; b = remap[test[:, 1:].reshape(-1)]
    ok = (a >= 0) & (b >= 0); a, b = a[ok], b[ok]
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    rb = remap[(vt[b] // R) * R + torch.randint(0, R, (len(b),), generator=gen, device=dev)]
    ok2 = rb >= 0; a, b, rb = a[ok2], b[ok2], rb[ok2]
    pos = (s0[b] * s1[a]).sum(1); neg = (s0[rb] * s1[a]).sum(1)
    return float((pos > neg).float().mean() + 0.5 * (pos == neg).float().mean())


def run(N, steps, scale):
    cfg = E.make_config(D, L, NV, negative=K, workers=0)
    ms = [E.SgnsModel.create(cfg, counts, 0) for _ in range(N)]
    shard = n // N
    B = shard // steps
    bufs = [torch.empty(ms[0].sync_size(), dtype=torch.float32, device=dev) for _ in range(N)]
    for m in ms:
        m.snapshot()
    for s in range(steps):
        for r, m in enumerate(ms):
            row0 = r * shard + s * B
            m.train(corpus, row0=row0, n_rows=B, walk_index_base=row0, words_before=int(s * B * L), words_scale=float(N), total_walks=n)
        if N > 1:
            for r, m in enumerate(ms):
                m.export_delta(bufs[r])
            total = bufs[0].clone()
            for r in range(1, N):
                total += bufs[r]
            torch.cuda.synchronize()
            for m in ms:
                m.import_delta(total, scale)
    a = auc(ms[0])
    for m in ms:
        m.close()
    return a


def run_blocks(N, steps):
    """the block schedule (embedding_amd/distributed.py: block_schedule_step), ranks simulated on this GPU"""
    sys.path.insert(0, 'tests')
    from helpers import simulate_block_schedule, simulate_gather_syn0
    cfg = E.make_config(D, L, NV, negative=K, workers=0)
    ms = [E.SgnsModel.create(cfg, counts, 0) for _ in range(N)]
    B = n // steps                                  # one global batch
    for s in range(steps):
        simulate_block_schedule(ms, lambda m: m.train(corpus, row0=s * B, n_rows=B, walk_index_base=s * B, words_before=int(s * B * L), total_walks=n))
    simulate_gather_syn0(ms)
    a = auc(ms[0]); pol = ms[0].schedule()
    for m in ms:
        m.close()
    return a, pol


print("N=1 (one GPU, all walks)          AUC %.4f" % run(1, 10, 1.0), flush=True)
for N in (2, 4, 8):
    for steps in (10, 40):
        print("N=%d exchanges/epoch=%-3d  average (1/N) AUC %.4f   sum (1.0) AUC %.4f" % (N, steps, run(N, steps, 1.0 / N), run(N, steps, 1.0)), flush=True)
for N in (2, 4, 8):
    a, pol = run_blocks(N, 10)
    print("N=%d block schedule, 10 global batches/epoch            AUC %.4f   (policy %d, %d workers)" % (N, a, pol["update_policy"], pol["workers"]), flush=True)
