"""Investigation aid: does WHERE the driver places a model's tables decide the per-launch time?  Creates models one after the
other in one process (keeping some alive so that later ones land elsewhere) and times one bench-sized launch on each."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E
from embedding_amd import synth
R, T, L, D, K = 41667, 24, 24, 128, 5
NV = R * T
G = synth.flow_graph_torch(R, T, 100, "cuda:0")
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(False); del G
torch.cuda.empty_cache()
n = NV
corpus = g.sample_walks_device(n, L, seed=5)
counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
cfg = E.make_config(D, L, NV, negative=K, workers=0, update_policy=5, epochs=1000)
def one(tag, keep):
    m = E.SgnsModel.create(cfg, counts, 0)
    ts = []
    for r in range(3):
        m.reset_stats(); m.train(corpus, 0, n, walk_index_base=0, total_walks=10 * n); ts.append(m.stats()["kernel_ms"])
    print("%-28s ms: %s" % (tag, " ".join("%.0f" % x for x in ts)), flush=True)
    if keep: return m
    m.close(); return None
held = []
for i in range(4):
    one("fresh, nothing held #%d" % i, False)
for i in range(4):
    held.append(one("fresh, %d models held" % i, True))
for m in held: m.close()
held = []
# ballast of odd sizes in front of the model: shifts the tables' physical placement
for mb in (3, 67, 515, 1031):
    b = torch.empty(mb * (1 << 20) + 4096 * 7, dtype=torch.uint8, device="cuda:0")
    one("after %d MB torch ballast" % mb, False)
    del b; torch.cuda.empty_cache()
