"""Per-row update rates of the busiest syn1neg rows of a workload's vocabulary (are float atomics on ONE row — 78 ns a 512-byte row, scripts/micro/hot_row_spread.hip —
what the mixed policy's head runs against?): python scripts/head_row_rates.py <workload> <edges per second of its bench line>"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench, embedding_amd as E
from embedding_amd import synth
name, rate = sys.argv[1], float(sys.argv[2])
wl = bench.WORKLOADS[name]; R, T, L, D, K = wl["R"], wl["T"], wl["L"], wl["dim"], wl["negative"]; NV = R * T; dev = "cuda:0"
G = synth.powerlaw_flow_graph_torch(R, T, wl["n_edges"], dev) if wl.get("powerlaw") else synth.flow_graph_torch(R, T, wl["mean_degree"], dev, dst=wl.get("dst", "uniform"))
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); del G; g.build_alias(exact=False)
n = wl["walks_per_vertex"] * NV
corpus = g.sample_walks_device(n, L, seed=20171106, rng_mode=1)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
c = np.sort(counts.cpu().numpy().astype(np.float64))[::-1]; c = c[c >= 2]
q = c ** 0.75; q /= q.sum(); p = c / c.sum()
row_ns = 78.0 * (-(-D // 64) * 64) / 128.0          # a row of `stride` floats: 78 ns per 512 bytes
print("%s: V %d, K %d, D %d; at %.3e pairs/s a negative row r takes pairs/s x K x q_r updates/s, a centre row pairs/s x p_r (gathered once per centre: / ~16)" % (name, len(c), K, D, rate))
for r in (0, 1, 2, 4, 9, 19, 49, 99, 299, 999, 2999, 9999):
    if r < len(c):
        upd = rate * K * q[r] + rate * p[r] / 16.0
        print("row %5d: q %.3e p %.3e -> %.3e updates/s = %5.1f %% of one row's atomic capacity (%.0f ns a row)" % (r, q[r], p[r], upd, 100 * upd * row_ns * 1e-9, row_ns))
cum = np.cumsum(q)
for h in (10, 30, 100, 1000, 10000): print("top %5d rows: %.1f %% of the negative draws" % (h, 100 * cum[min(h, len(c)) - 1]))
