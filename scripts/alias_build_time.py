"""dge_graph_build_alias on the bench graphs (setup, not the hot path): seconds per build — Vose pairing as bench.py uses it and, where the
graph has no hubs, the reference's pairing order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import embedding_amd as E
from embedding_amd import synth
from bench import WORKLOADS

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cfg2", "cfg3", "cfg5"]:
    wl = WORKLOADS[name]; R, T = wl["R"], wl["T"]
    G = synth.powerlaw_flow_graph_torch(R, T, wl["n_edges"], dev) if wl.get("powerlaw") else synth.flow_graph_torch(R, T, wl["mean_degree"], dev, dst=wl.get("dst", "uniform"))
    g = E.DeviceGraph(0)
    g.add_edges_device(G["src"], G["dst"], G["w"]); n_edges = G["n_edges"]
    sources = G["sources"] if T > 1 else np.arange(R, dtype=np.int32)
    del G; torch.cuda.empty_cache()
    g.set_sources(sources)
    for exact in ((False,) if wl.get("powerlaw") else (False, True)):
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g.build_alias(exact=exact)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%-5s %-16s alias build %.3f s   (%d edges, %d sources)" % (name, "reference order:" if exact else "Vose:", best, n_edges, len(sources)), flush=True)
    del g
