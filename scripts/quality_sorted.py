"""Statistical parity of the owner-computes schedule (update_policy 8) as a function of its mini-batch size: a 100 k-vertex layered
graph with community structure (the graph of scripts/quality_scale.py at 1/10), one epoch of 1 M walks, D = 128, K = 5; link-prediction
AUC on held-out walk steps.  Policy 2 (lossless atomics) and the lock kernel are the yardsticks."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from helpers import link_auc
R, T, L, D, K = 4166, 24, 24, 128, 5
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
dreg = torch.where(inside, local.clamp_(max=R - 1), anyw)
dst = (((src // R + 1) % T) * R + dreg).to(torch.int32)
w = (1.0 + torch.floor(-20.0 * torch.log(torch.rand(Etot, generator=g0, device=dev, dtype=torch.float64).clamp_(min=1e-12))))
g = E.DeviceGraph(0); g.add_edges_device(src.contiguous(), dst.contiguous(), w.contiguous()); del src, dst, w, reg, inside, local, anyw, dreg
g.set_sources(np.arange(R, dtype=np.int32)); g.build_alias(False)
n = 10 * NV
corpus = g.sample_walks_device(n, L, seed=5)
test = g.sample_walks(100_000, L, seed=99)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
runs = [(2, None)] + [(8, int(x)) for x in (sys.argv[1:] or [100_000, 20_000, 5_000, 1_000])]
for pol, per in runs:
    cfg = E.make_config(D, L, NV, negative=K, workers=0, update_policy=pol)
    with E.tuning(**({"sorted_walks": per} if per else {})):
        m = E.SgnsModel.create(cfg, counts, 0)
        t = time.time(); m.train(corpus); st = m.stats()
    s0, vid = m.vectors()
    items_per_row = (per or 0) * 383.0 * (K + 1) / NV
    print("policy %d  mini-batch %8s walks (%7.0f items per row)  kernel %.2f s -> %.3e edges/s | AUC %.4f" %
          (pol, per, items_per_row, st["kernel_ms"] / 1e3, st["pairs"] / (st["kernel_ms"] / 1e3), link_auc(s0, m.syn1neg(), vid, test, R)), flush=True)
    m.close()
