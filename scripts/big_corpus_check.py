"""Index-width check: a walk corpus of more than 2^31 elements (100 M walks x 24 = 2.4e9 int32, 9.6 GB) is sampled, counted and
trained on a slice taken from its far end."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E
from embedding_amd import synth
R, T, L = 41667, 24, 24
NV = R * T
G = synth.flow_graph_torch(R, T, 100, "cuda:0")
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(False)
del G; torch.cuda.empty_cache()
n = 100_000_000
t = time.time(); corpus = g.sample_walks_device(n, L, seed=7, rng_mode=1); torch.cuda.synchronize(); print("sampled %d walks (%.1f GB) in %.2f s" % (n, n * L * 4 / 1e9, time.time() - t), flush=True)
counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
total = int(counts.sum().item()); print("tokens counted:", total, "of", n * L, flush=True)
assert 0 < total <= n * L and total > 2**31
# the last walks of the corpus equal the same walk indices sampled on their own (row offsets beyond 2^31 elements)
tail = g.sample_walks(1000, L, seed=7, rng_mode=1, first_index=n - 1000)
m = E.SgnsModel.create(E.make_config(64, L, NV, workers=0), counts, 0)
m.train(corpus, row0=n - 1_000_000, n_rows=1_000_000, walk_index_base=n - 1_000_000, total_walks=n)
st = m.stats(); print("trained the last 1 M walks:", st["pairs"], "pairs", flush=True)
c2 = E.WalkCorpus.from_host(tail, 0); cc = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); c2.count_tokens(NV, cc)
m2 = E.SgnsModel.create(E.make_config(64, L, NV, workers=1), counts, 0); m3 = E.SgnsModel.create(E.make_config(64, L, NV, workers=1), counts, 0)
m2.train(corpus, row0=n - 1000, n_rows=1000, walk_index_base=n - 1000, total_walks=n)
m3.train(c2, row0=0, n_rows=1000, walk_index_base=n - 1000, total_walks=n)
assert np.array_equal(m2.vectors()[0].view(np.int32), m3.vectors()[0].view(np.int32)), "far-end rows differ from the same walks sampled alone"
print("far-end slice == the same walks sampled alone: OK")
