"""The two count-derived rules of a block of the multi-GPU schedule, evaluated offline on the bench graphs: the head (rows that take atomics; block_head's bound at several
constants) and the accumulator banks' switch (the busiest row's chain against a pair's row traffic).   python scripts/block_head_rule.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import embedding_amd as E
from embedding_amd import synth
dev = "cuda:0"
for name, R, T, D, K, kw in (("cfg3_zipf", 41667, 24, 128, 5, dict(mean_degree=100, dst="zipf")), ("community_zipf", 41667, 24, 128, 5, dict(mean_degree=100, dst="community_zipf")),
                             ("cfg5", 416667, 24, 256, 20, dict(n_edges=1_000_000_000, powerlaw=True))):
    NV = R * T; L = 24
    if "powerlaw" in kw:
        G = synth.powerlaw_flow_graph_torch(R, T, kw["n_edges"], dev)
    else:
        G = synth.flow_graph_torch(R, T, kw["mean_degree"], dev, dst=kw["dst"])
    if G is None:
        print(name, "no generator"); continue
    g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); del G
    torch.cuda.empty_cache(); g.build_alias(False)
    corpus = g.sample_walks_device(1_000_000, L, seed=20171106)
    counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
    c = np.sort(counts.cpu().numpy())[::-1].astype(np.float64); c = c[c >= 2]
    V = len(c); tw = c.sum(); q = c ** 0.75; q /= q.sum(); p = c / tw
    t = (q * q + p * p)[::-1].cumsum()[::-1]
    W, n = 6144, 8
    heads = []
    for f in (0.1, 0.2, 0.4):
        H = int(np.searchsorted(-t, -f / (5.0 * W * n))); Hs = int((W * n * p > 0.5).sum()); heads.append("%.1f: %d" % (f, max(H, Hs)))
    stride = -(-D // 64) * 64
    chain = n * max(p[0], K * q[0]) * 78e-9; pair = 8.0 * stride * (K + 2) / 3e12
    print("%-15s V %8d  head at bound %s | busiest row: %.2e of the contexts, %.2e of the draws; its chain %.2f ns a pair against %.2f ns of row traffic (ratio %.2f)" %
          (name, V, ", ".join(heads), p[0], q[0], chain * 1e9, pair * 1e9, chain / pair), flush=True)
    del g, corpus
