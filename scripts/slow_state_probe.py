"""Is a "slow" process (profiles/r02_box_drift.txt) slow because of WHERE its tables lie?  One process: the bench-sized corpus once, then models
created one after the other (earlier ones kept or freed), one bench-sized launch timed on each.  Run several times back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import embedding_amd as E
from embedding_amd import synth
R, T, L, D, K = 41667, 24, 24, 128, 5
NV = R * T
G = synth.flow_graph_torch(R, T, 100, "cuda:0")
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); g.build_alias(False); del G
torch.cuda.empty_cache()
corpus = g.sample_walks_device(NV, L, seed=5)
counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
cfg = E.make_config(D, L, NV, negative=K, workers=0, update_policy=5, epochs=1000)
def one(m):
    ts = []
    for r in range(2):
        m.reset_stats(); m.train(corpus, 0, NV, walk_index_base=0, total_walks=10 * NV); ts.append(m.stats()["kernel_ms"])
    r = m.row_rates()
    with E.tuning(static_walks=1):
        st = []
        for r2 in range(2):
            m.reset_stats(); m.train(corpus, 0, NV, walk_index_base=0, total_walks=10 * NV); st.append(m.stats()["kernel_ms"])
    return "%.0f ms, walks w, w+W, ...: %.0f ms (rewrite %.0f GB/s)" % (min(ts), min(st), r[1])
out = []
a = E.SgnsModel.create(cfg, counts, 0); out.append("A %s" % one(a))
b = E.SgnsModel.create(cfg, counts, 0); out.append("B(A held) %s" % one(b))
out.append("A again %s" % one(a))
a.close()
c = E.SgnsModel.create(cfg, counts, 0); out.append("C(A freed) %s" % one(c))
corpus2 = g.sample_walks_device(NV, L, seed=5)
corpus, old = corpus2, corpus
out.append("C, corpus re-sampled %s" % one(c))
print("pid %d: %s" % (os.getpid(), " | ".join(out)), flush=True)
