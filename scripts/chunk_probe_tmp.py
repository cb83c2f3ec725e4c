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
def launch(m, n=2):
    ts = []
    for _ in range(n):
        m.reset_stats(); m.train(corpus, 0, NV, walk_index_base=0, total_walks=10 * NV); ts.append(m.stats()["kernel_ms"])
    return min(ts)
alive = []
modes = [int(x) for x in sys.argv[1].split(",")]
for r in range(int(sys.argv[2])):
    for kb in modes:
        t0 = time.time()
        with E.tuning(alloc_chunk_kb=kb):
            m = E.SgnsModel.create(cfg, counts, 0)
        tc = time.time() - t0
        print("round %d chunk %5d KiB: %.1f ms per launch (create %.1f s) rows read %.0f GB/s" % (r, kb, launch(m), tc, m.row_rates()[0]), flush=True)
        alive.append(m)
        while len(alive) > 2:
            alive.pop(0).close()
