"""A/B of one tuning knob on ONE model in ONE process (consecutive processes differ by more than most effects: profiles/r02_box_drift.txt):
python scripts/ab_inproc.py <workload> <knob>=<value> [launches] [hs|policy=N] — alternates bench-sized launches without and with the knob."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import embedding_amd as E
from embedding_amd import synth
from bench import WORKLOADS

name, knob = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4] if len(sys.argv) > 4 else ""
k, v = knob.split("="); v = int(v)
wl = WORKLOADS[name]; R, T, L, D, K = wl["R"], wl["T"], wl["L"], wl["dim"], wl["negative"]
NV = R * T; dev = torch.device("cuda:0")
G = synth.powerlaw_flow_graph_torch(R, T, wl["n_edges"], dev) if wl.get("powerlaw") else synth.flow_graph_torch(R, T, wl["mean_degree"], dev, dst=wl.get("dst", "uniform"))
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"])
sources = G["sources"] if T > 1 else np.arange(R, dtype=np.int32)
del G; torch.cuda.empty_cache()
g.set_sources(sources); g.build_alias(exact=False)
n = wl["walks_per_vertex"] * NV if wl["walks_per_vertex"] * NV <= NV else NV
corpus = g.sample_walks_device(n, L, seed=5, rng_mode=1)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
m = E.SgnsModel.create(E.make_config(D, L, NV, negative=K, min_count=2, workers=0, epochs=1000, seed=1, use_hs=extra == "hs",
                                      update_policy=int(extra.split("=")[1]) if extra.startswith("policy=") else 0), counts, 0)
def launch():
    m.reset_stats(); m.train(corpus, 0, n, walk_index_base=0, total_walks=10 * n); return m.stats()["kernel_ms"]
launch()
a, b = [], []
for _ in range(reps):
    a.append(launch())
    with E.tuning(**{k: v}): b.append(launch())
print("%s on one model, ms per launch (schedule %s): default %s | %s %s" % (name, m.schedule(), " ".join("%.1f" % x for x in a), knob, " ".join("%.1f" % x for x in b)), flush=True)
