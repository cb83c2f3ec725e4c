"""Round-3 placement experiment (profiles/r03_placement.txt).  One process, the bench-sized cfg3 corpus once; then models created one after
the other under every allocation rule of DGE_TUNE_ALLOC (0 hipMalloc, 1 hipExtMallocWithFlags(contiguous), 2 virtual-memory API with 1 GiB-aligned
ranges), each timed on the same bench-sized launch twice: reading word2vec's flat unigram table (400 MB) and reading its rank-block form (16.7 MB).
python scripts/placement_r3.py [rounds] [modes, e.g. 0,1,2] [keep: how many earlier models stay allocated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import embedding_amd as E
from embedding_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
modes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2").split(",")]
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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
print("pid %d, modes %s" % (os.getpid(), modes), flush=True)
for r in range(rounds):
    for mode in modes:
        try:
            with E.tuning(alloc=mode, full_table=1):
                m = E.SgnsModel.create(cfg, counts, 0)
        except Exception as e:  # an allocation rule this runtime refuses
            print("round %d alloc %d: create failed: %s" % (r, mode, e), flush=True)
            continue
        with E.tuning(full_table=1):
            flat = launch(m)
            rates_flat = m.row_rates()
        compact = launch(m)
        rates = m.row_rates()
        with E.tuning(full_table=1):
            flat2 = launch(m, 1)
        print("round %d alloc %d: flat table %.1f ms (again %.1f) | rank blocks %.1f ms | look-ups/s flat %.3g blocks %.3g | rows read %.0f rewritten %.0f GB/s"
              % (r, mode, flat, flat2, compact, rates_flat[3], rates[3], rates[0], rates[1]), flush=True)
        alive.append(m)
        while len(alive) > keep:
            alive.pop(0).close()
