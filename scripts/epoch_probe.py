"""Bisecting aid: one epoch on the community graph of scripts/quality_scale.py under policy 5 (and 3), kernel time only.
Tolerates older libdge.so builds that lack newer entry points."""
import ctypes, sys, time
_orig = ctypes.CDLL.__getattr__
def _tolerant(self, name):
    try:
        return _orig(self, name)
    except AttributeError:
        if name.startswith("dge_"):
            def missing(*a): raise RuntimeError(name + " missing in this build")
            missing.argtypes = None; missing.restype = None
            return missing
        raise
ctypes.CDLL.__getattr__ = _tolerant
import numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E
R, T, L, D, K = 41667, 24, 24, 128, 5
NV = R * T
dev = "cuda:0"
COMM = len(sys.argv) < 2 or sys.argv[1] != "random"
g0 = torch.Generator(device=dev); g0.manual_seed(1)
deg = torch.exp(np.log(100) - 0.5 + torch.randn(NV, generator=g0, device=dev)).to(torch.int64).clamp_(1, R)
Etot = int(deg.sum().item())
src = torch.repeat_interleave(torch.arange(NV, device=dev, dtype=torch.int32), deg)
reg = src % R
inside = torch.rand(Etot, generator=g0, device=dev) < (0.8 if COMM else 0.0)
local = (reg // 64) * 64 + torch.randint(0, 64, (Etot,), generator=g0, device=dev, dtype=torch.int32)
anyw = torch.randint(0, R, (Etot,), generator=g0, device=dev, dtype=torch.int32)
dreg = torch.where(inside, local.clamp_(max=R - 1), anyw)
dst = (((src // R + 1) % T) * R + dreg).to(torch.int32)
w = (1.0 + torch.floor(-20.0 * torch.log(torch.rand(Etot, generator=g0, device=dev, dtype=torch.float64).clamp_(min=1e-12))))
if 'synth' in sys.argv:                 # the bench's own generator instead
    from embedding_amd import synth
    G = synth.flow_graph_torch(R, T, 100, dev); src, dst, w = G["src"], G["dst"], G["w"]; del G
g = E.DeviceGraph(0); g.add_edges_device(src.contiguous(), dst.contiguous(), w.contiguous()); del src, dst, w, reg, inside, local, anyw, dreg
g.set_sources(np.arange(R, dtype=np.int32)); g.build_alias(False)
if 'release' in sys.argv:
    torch.cuda.empty_cache()          # hand torch's cached blocks back to the driver before libdge allocates its tables
n = 10 * NV
corpus = g.sample_walks_device(n, L, seed=5)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
cf = counts.double()
print("tokens %d, vocabulary(>=2) %d, max count %d, mean %.1f, std %.1f, top-10 counts %s" % (int(cf.sum()), int((counts >= 2).sum()), int(cf.max()), float(cf.mean()), float(cf.std()), torch.sort(counts, descending=True)[0][:10].tolist()), flush=True)
for pol in (0, 5, 3):
    cfg = E.make_config(D, L, NV, negative=K, workers=0, update_policy=pol, epochs=(1000 if 'flat' in sys.argv else 1))
    m = E.SgnsModel.create(cfg, counts, 0)
    nb = n // 10; out = []
    for b in range(10):
        m.reset_stats(); m.train(corpus, b * nb, nb, walk_index_base=b * nb, total_walks=n); out.append(m.stats()["kernel_ms"])
    print("community" if COMM else "random", "policy", pol, "pairs/launch %d" % m.stats()["pairs"], (m.schedule() if pol == 0 else ""), "per-launch ms:", " ".join("%.0f" % x for x in out), flush=True)
    m.close()
