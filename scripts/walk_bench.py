"""Walk-sampling times on the reference's own graph shapes (CA 77 regions x 24 layers, tract 801 x 8),
for 0.5/1/2/5/10 M walks — the only numbers the reference publishes for this path (P/running_time.py:16-21, which
include Java string formatting and file output; here: device sampling, java-sequential RNG stream, ids to host)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import embedding_amd as E
from embedding_amd import synth

for name, R, T, deg, dead in (("CA 77x24", 77, 24, 40, 0.0), ("tract 801x8", 801, 8, 120, 0.0), ("tract 801x8 with 2% dead ends", 801, 8, 120, 0.02)):
    G = synth.flow_graph_numpy(R, T, deg, seed=1, dead_end_fraction=dead)
    g = E.DeviceGraph(0); g.add_edges(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"])
    t0 = time.perf_counter(); g.build_alias(exact=True); t_alias = time.perf_counter() - t0
    g.sample_walks_device(1000, T, seed=1, rng_mode=0)          # warm-up
    row = []
    for n in (500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000):
        t0 = time.perf_counter(); c = g.sample_walks_device(n, T, seed=2017, rng_mode=0); t_dev = time.perf_counter() - t0
        t0 = time.perf_counter(); w = c.to_host(); t_host = time.perf_counter() - t0
        row.append("%.1fM: %.4fs device + %.3fs to host" % (n / 1e6, t_dev, t_host))
    print(name, "| edges", len(G["src"]), "| alias build (reference order) %.3fs |" % t_alias, " ; ".join(row), flush=True)
