"""Investigation aid: per-launch kernel times of one epoch (10 launches of 1 M walks) under policy 5 on the bench graph, inside one
process — stable to 0.3 % within a process, +-5 % between processes on one box."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E
from embedding_amd import synth
R, T, L, D, K = 41667, 24, 24, 128, 5
NV = R * T
G = synth.flow_graph_torch(R, T, 100, "cuda:0")
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(False); del G
n = 10 * NV
corpus = g.sample_walks_device(n, L, seed=5)
counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
m = E.SgnsModel.create(E.make_config(D, L, NV, negative=K, workers=0, update_policy=5, epochs=1000), counts, 0)
nb = n // 10; out = []
for b in range(10):
    m.reset_stats(); m.train(corpus, b * nb, nb, walk_index_base=b * nb, total_walks=n); out.append(m.stats()["kernel_ms"])
print("per-launch ms:", " ".join("%.0f" % x for x in out), flush=True)
# the same first batch again on the trained model, and on a fresh model
m.reset_stats(); m.train(corpus, 0, nb, walk_index_base=0, total_walks=n); print("first batch again, trained model: %.0f ms" % m.stats()["kernel_ms"])
m2 = E.SgnsModel.create(E.make_config(D, L, NV, negative=K, workers=0, update_policy=5, epochs=1000), counts, 0)
m2.train(corpus, 9 * nb, nb, walk_index_base=9 * nb, total_walks=n); print("last batch, fresh model: %.0f ms" % m2.stats()["kernel_ms"])
