"""GPU parity: edge store, alias tables and walk sampler of libdge.so vs the CPU oracle, through the C ABI.

Integer/index outputs and the double-precision tables must be BIT-EXACT.
Reference behaviour under test: J/LayeredGraph.java:54-82,104-116,157-252, J/SpatialGraph.java:29-35,105-108;
golden vector: T/LayeredGraphTest.java:12-44.
"""
import numpy as np
import pytest

from helpers import bits, build_both, layered_graph

pytestmark = pytest.mark.gpu


def test_layered_graph_test_golden_vector(dge):
    """T/LayeredGraphTest.java:12-44 through the device path."""
    g = dge.DeviceGraph(0)
    g.add_edges([0, 0, 0], [1, 2, 3], [2.0, 10.0, 8.0])
    g.set_sources([0])
    g.build_alias(exact=True)
    a = g.get_alias(0)
    assert a["alias"].tolist() == [1, 2, -1]
    assert a["prob"].tolist() == [0.3, 0.8, 1.0]          # exact double equality, as assertEquals does
    assert a["out_degree"] == 20.0
    assert [g.sample_next(0, x) for x in (0.05, 0.3, 0.4, 0.65, 0.9)] == [1, 2, 2, 3, 3]
    assert g.sample_next(1, 0.5) == -1                     # no out-edges -> null (J/LayeredGraph.java:106-107)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("seed", [0, 1])
def test_alias_tables_bit_exact(dge, oracle, exact, seed):
    src, dst, w, sources = layered_graph(R=60, T=5, deg=9, seed=seed)
    og, dg = build_both(oracle, dge, src, dst, w, sources, exact=exact)
    assert dg.num_vertices == og.num_vertices and dg.num_edges == og.num_edges
    for v in range(og.num_vertices):
        a, b = og.get_alias(v), dg.get_alias(v)
        assert np.array_equal(a["nbr"], b["nbr"]), v                 # insertion order kept per vertex
        assert np.array_equal(bits(a["weight"]), bits(b["weight"])), v
        assert a["out_degree"] == b["out_degree"], v
        assert np.array_equal(a["alias"], b["alias"]), v
        assert np.array_equal(bits(a["prob"]), bits(b["prob"])), v
    sa, sb = og.get_source_alias(), dg.get_source_alias()
    assert sa["weight_sum"] == sb["weight_sum"]
    assert np.array_equal(sa["src"], sb["src"]) and np.array_equal(sa["alias"], sb["alias"])
    assert np.array_equal(bits(sa["prob"]), bits(sb["prob"]))


def test_alias_hub_vertex_exact_order(dge, oracle):
    """A 5000-edge hub: the bit-set restatement of the reference's O(k^2) pairing must give identical arrays."""
    rng = np.random.default_rng(5)
    k = 5000
    w = np.floor(rng.pareto(1.1, k) * 3) + 1
    src = np.zeros(k, np.int32); dst = np.arange(1, k + 1, dtype=np.int32)
    og, dg = build_both(oracle, dge, src, dst, w, [0], exact=True)
    a, b = og.get_alias(0), dg.get_alias(0)
    assert np.array_equal(a["alias"], b["alias"]) and np.array_equal(bits(a["prob"]), bits(b["prob"]))


@pytest.mark.parametrize("shape", ["pareto", "near_uniform", "two_values", "one_giant"])
def test_alias_vose_tables_of_thousands_of_slots(dge, oracle, shape):
    """Vose order on big tables (hub vertices, the source table) is built by a whole wave with register windows over the two stacks
    (alias_vose_wave): the same arrays as the serial loop, bit for bit — hubs of 1023 / 1024 / 1025 / 5000 / 40 000 slots, 3000 sources."""
    rng = np.random.default_rng(len(shape))
    ks = [1023, 1024, 1025, 5000, 40_000]
    src, dst, w = [], [], []
    nxt = len(ks)
    for h, k in enumerate(ks):
        if shape == "pareto": wk = np.floor(rng.pareto(1.1, k) * 3) + 1
        elif shape == "near_uniform": wk = 1000.0 + rng.integers(-1, 2, k)            # slots a hair under, at and over 1: long alternations
        elif shape == "two_values": wk = np.where(rng.random(k) < 0.9, 1.0, 37.0)
        else: wk = np.ones(k); wk[k // 3] = 50.0 * k                               # one large serves nearly every small
        src.append(np.full(k, h, np.int32)); dst.append(np.arange(nxt, nxt + k, dtype=np.int32)); w.append(wk)
    src = np.concatenate(src); dst = np.concatenate(dst); w = np.concatenate(w).astype(np.float64)
    # give 3000 of the leaves an out-edge each so that the source table has 3000 slots of varied weight
    leaves = np.arange(len(ks), len(ks) + 3000, dtype=np.int32)
    src = np.concatenate([src, leaves]); dst = np.concatenate([dst, np.zeros(3000, np.int32)])
    w = np.concatenate([w, np.floor(rng.pareto(1.3, 3000) * 5) + 1])
    og, dg = build_both(oracle, dge, src, dst, w, leaves, exact=False)
    for v in range(len(ks)):
        a, b = og.get_alias(v), dg.get_alias(v)
        assert np.array_equal(a["alias"], b["alias"]) and np.array_equal(bits(a["prob"]), bits(b["prob"])), (shape, v)
    sa, sb = og.get_source_alias(), dg.get_source_alias()
    assert np.array_equal(sa["alias"], sb["alias"]) and np.array_equal(bits(sa["prob"]), bits(sb["prob"])), shape
    # and the walk slots built from them
    n = 4000
    wo = og.sample_walks(n, 3, seed=11, rng_mode=1); wd = dg.sample_walks(n, 3, seed=11, rng_mode=1)
    assert np.array_equal(wo, wd)


@pytest.mark.parametrize("exact", [True, False])
def test_walks_strided_bit_exact(dge, oracle, exact):
    src, dst, w, sources = layered_graph(R=50, T=6, deg=7, seed=3)
    og, dg = build_both(oracle, dge, src, dst, w, sources, exact=exact)
    for first in (0, 1234):
        a = og.sample_walks(3000, 6, seed=42, rng_mode=1, first_index=first)
        b = dg.sample_walks(3000, 6, seed=42, rng_mode=1, first_index=first)
        assert np.array_equal(a, b)
    # shards of one corpus reproduce the corpus
    whole = dg.sample_walks(1000, 6, seed=7, rng_mode=1)
    parts = np.concatenate([dg.sample_walks(250, 6, seed=7, rng_mode=1, first_index=250 * r) for r in range(4)])
    assert np.array_equal(whole, parts)


def test_walks_java_sequential_stream(dge, oracle):
    """rng_mode 0 = what the reference yields after LayeredGraph.rnd = new Random(seed) (J/LayeredGraph.java:14)."""
    src, dst, w, sources = layered_graph(R=50, T=6, deg=7, seed=4)
    og, dg = build_both(oracle, dge, src, dst, w, sources)
    a, da = og.sample_walks(2000, 6, seed=99, rng_mode=0, return_draws=True)
    b, db = dg.sample_walks(2000, 6, seed=99, rng_mode=0, return_draws=True)
    assert np.array_equal(a, b) and da == db == 2000 * 6
    assert np.array_equal(a, dg.sample_walks(2000, 6, seed=99, rng_mode=1))   # no dead ends: both layouts agree
    # continuing the stream: draws already consumed
    c = og.sample_walks(500, 6, seed=99, rng_mode=0, first_index=da)
    d = dg.sample_walks(500, 6, seed=99, rng_mode=0, first_index=db)
    assert np.array_equal(c, d)


def test_walks_dead_ends_are_short_not_errors(dge, oracle):
    """J/LayeredGraph.java:247-248: a vertex without out-edges ends the walk; the draw count becomes data dependent."""
    src, dst, w, sources = layered_graph(R=40, T=6, deg=4, seed=8, dead_ends=0.25)
    og, dg = build_both(oracle, dge, src, dst, w, sources)
    a1 = og.sample_walks(4000, 6, seed=5, rng_mode=1); b1 = dg.sample_walks(4000, 6, seed=5, rng_mode=1)
    assert np.array_equal(a1, b1)
    assert (a1 == -1).any() and (a1[:, 0] >= 0).all()
    a0, da = og.sample_walks(1500, 6, seed=5, rng_mode=0, return_draws=True)
    b0, db = dg.sample_walks(1500, 6, seed=5, rng_mode=0, return_draws=True)
    assert np.array_equal(a0, b0) and da == db and da < 1500 * 6
    assert da == int((a0 >= 0).sum())                    # one draw per node of the walk
    # unaligned continuation of the sequential stream
    c = og.sample_walks(300, 6, seed=5, rng_mode=0, first_index=da)
    d = dg.sample_walks(300, 6, seed=5, rng_mode=0, first_index=db)
    assert np.array_equal(c, d)
    # long corpora go through the parallel resolver of the sequential stream (block exits + chain): still the exact walks
    for n, first in ((2048, 0), (50_000, 0), (30_001, da)):
        a2, d2 = og.sample_walks(n, 6, seed=77, rng_mode=0, first_index=first, return_draws=True)
        b2, e2 = dg.sample_walks(n, 6, seed=77, rng_mode=0, first_index=first, return_draws=True)
        assert np.array_equal(a2, b2) and d2 == e2 and d2 == int((a2 >= 0).sum())


def test_keep_nearest_k_vertices(dge, oracle):
    """J/SpatialGraph.java:29-35 incl. ties (stable sort keeps insertion order) and the self-loop of weight 1."""
    rng = np.random.default_rng(11)
    R = 30
    pts = rng.random((R, 2)) * 0.05
    src, dst, w = [], [], []
    for i in range(R):
        for j in range(R):
            d = float(np.hypot(*(pts[i] - pts[j])))
            src.append(i); dst.append(j); w.append(float(np.round(np.exp(-d * 100), 2)))   # rounding forces ties
    og, dg = build_both(oracle, dge, np.array(src, np.int32), np.array(dst, np.int32), np.array(w), np.arange(R, dtype=np.int32),
                        exact=True, stream_sum=True, top_k=10)
    assert dg.num_edges == R * 10
    for v in range(R):
        a, b = og.get_alias(v), dg.get_alias(v)
        assert np.array_equal(a["nbr"], b["nbr"]) and np.array_equal(bits(a["weight"]), bits(b["weight"]))
        assert a["out_degree"] == b["out_degree"]
        assert b["weight"][0] == 1.0 and (np.diff(b["weight"]) <= 0).all()
        assert np.array_equal(a["alias"], b["alias"]) and np.array_equal(bits(a["prob"]), bits(b["prob"]))
    assert og.get_source_alias()["weight_sum"] == dg.get_source_alias()["weight_sum"]
    wa = og.sample_walks(2000, 8, seed=1, rng_mode=1); wb = dg.sample_walks(2000, 8, seed=1, rng_mode=1)
    assert np.array_equal(wa, wb)
    with pytest.raises(dge.DgeError) as ei:      # subList(0,k) throws when a vertex has fewer than k edges
        dg.keep_top_k(11)
    assert ei.value.code == 3


def test_position_prefix_rule(dge):
    """J/SpatialGraph.java:105-108: token j becomes "j-name" -> id j*R + name."""
    walks = np.array([[3, 1, 2, -1], [0, 0, 0, 0]], np.int32)
    c = dge.WalkCorpus.from_host(walks, 0)
    c.add_position_prefix(5)
    assert c.to_host().tolist() == [[3, 6, 12, -1], [0, 5, 10, 15]]


def test_error_behaviour_and_empty_inputs(dge):
    g = dge.DeviceGraph(0)
    g.add_edges([0, 1], [1, 2], [1.0, 1.0])
    with pytest.raises(dge.DgeError) as ei:
        g.sample_walks(4, 3, seed=1)
    assert ei.value.code == 5                         # alias tables not built yet
    with pytest.raises(dge.DgeError) as ei:
        g.set_sources([7])
    assert ei.value.code == 2
    with pytest.raises(dge.DgeError):
        g.add_edges([-1], [0], [1.0])
    g.set_sources([])
    g.build_alias(True)
    assert (g.sample_walks(5, 3, seed=1) == -1).all()  # no sources: nothing to start from
    g.set_sources([2]); g.build_alias(True)           # source without out-edges: one-node walks
    w = g.sample_walks(5, 3, seed=1)
    assert (w[:, 0] == 2).all() and (w[:, 1:] == -1).all()
    assert g.sample_walks(0, 3, seed=1).shape == (0, 3)
    with pytest.raises(dge.DgeError) as ei:
        dge.DeviceGraph(99)
    assert ei.value.code == 6


def test_full_size_walk_properties(dge):
    """cfg2-sized store (100k vertices, ~5M edges): size-independent properties instead of an oracle replay."""
    import torch
    from embedding_amd import synth
    R, T = 25_000, 4
    G = synth.flow_graph_torch(R, T, 50, "cuda:0")
    g = dge.DeviceGraph(0)
    g.add_edges_device(G["src"], G["dst"], G["w"])
    g.set_sources(G["sources"])
    g.build_alias(exact=False)
    assert g.num_vertices == R * T and g.num_edges == G["n_edges"]
    n = 400_000
    walks = g.sample_walks(n, T, seed=9, rng_mode=1)
    assert (walks >= 0).all()                                   # every vertex has out-edges here
    assert (walks // R == np.arange(T)[None, :]).all()          # step j reads slice j (J/CrossTimeGraph.java:36-39)
    # every step is an edge of the store
    key = G["src"].to(torch.int64) * (R * T) + G["dst"].to(torch.int64)
    key = torch.unique(key)
    wt = torch.from_numpy(walks).to("cuda:0").to(torch.int64)
    step = (wt[:, :-1] * (R * T) + wt[:, 1:]).reshape(-1)
    pos = torch.searchsorted(key, step).clamp_(max=key.numel() - 1)
    assert bool((key[pos] == step).all())
    # transition frequencies out of the most visited source follow the edge weights (chi-square, 5 sigma)
    v0 = int(np.bincount(walks[:, 0]).argmax())
    a = g.get_alias(v0, tables=False)
    nxt = walks[walks[:, 0] == v0, 1]
    exp = {}
    for d, ww in zip(a["nbr"], a["weight"]):
        exp[int(d)] = exp.get(int(d), 0.0) + ww / a["out_degree"]
    obs = np.array([np.sum(nxt == d) for d in exp]); e = np.array(list(exp.values())) * len(nxt)
    chi2 = float(((obs - e) ** 2 / e).sum()); dof = len(e) - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof) + 10, (chi2, dof)
    # source choice follows out-degree (J/LayeredGraph.java:199-204)
    sa = g.get_source_alias()
    assert abs(sa["prob"].mean() - 1.0) < 0.5 and sa["weight_sum"] > 0


@pytest.mark.parametrize("exact", [True, False])
def test_host_held_public_fields_are_honoured(dge, oracle, exact):
    """The reference's Vertex.outDegree and LayeredGraph.sourceWeightSum are fields that callers assign (J/LayeredGraph.java:35,146;
    J/SpatialGraph.java:33,57), and addSourceVertex may create vertices no edge names (:182-183).  A host that keeps those fields
    hands them over (dge_graph_reserve_vertices / set_out_degree / set_source_weight_sum) and reads every table back in one piece
    (dge_graph_get_csr): bit-exact against the oracle fed the same values, walks included."""
    src, dst, w, sources = layered_graph(R=50, T=4, deg=7, seed=11)
    rng = np.random.default_rng(3)
    V = int(max(src.max(), dst.max())) + 1
    og, dg = oracle.Graph(), dge.DeviceGraph(0)
    for g in (og, dg):
        g.add_edges(src, dst, w)
        g.reserve_vertices(V + 3)                                    # three isolated vertices (unregistered sources)
    od = og.get_csr(tables=False)["out_degree"].copy()
    assert len(od) == V + 3 and np.array_equal(bits(od), bits(dg.get_csr(tables=False)["out_degree"]))
    od[:V] *= rng.choice([1.0, 1.0, 1.0 + 2.0 ** -40, 0.5, 3.0], V)  # fields as a caller left them, not the running sums
    srcs = np.concatenate([sources, [V + 1, V]]).astype(np.int32)
    for g in (og, dg):
        g.set_out_degree(od)
        g.set_sources(srcs)
        g.set_source_weight_sum(float(od[srcs].sum()) * 1.25)
        g.build_alias(exact)
    a, b = og.get_csr(), dg.get_csr()
    assert np.array_equal(a["row_ptr"], b["row_ptr"]) and np.array_equal(a["nbr"], b["nbr"]) and np.array_equal(a["alias"], b["alias"])
    for k in ("weight", "prob", "out_degree"):
        assert np.array_equal(bits(a[k]), bits(b[k])), k
    assert np.array_equal(bits(b["out_degree"]), bits(od))
    for v in (0, 7, V - 1, V + 2):                                    # the per-vertex view agrees with the bulk one
        x = dg.get_alias(v); lo, hi = b["row_ptr"][v], b["row_ptr"][v + 1]
        assert np.array_equal(bits(x["prob"]), bits(b["prob"][lo:hi])) and np.array_equal(x["alias"], b["alias"][lo:hi])
    sa, sb = og.get_source_alias(), dg.get_source_alias()
    assert sa["weight_sum"] == sb["weight_sum"] == float(od[srcs].sum()) * 1.25
    assert np.array_equal(sa["alias"], sb["alias"]) and np.array_equal(bits(sa["prob"]), bits(sb["prob"]))
    for mode in (0, 1):
        assert np.array_equal(og.sample_walks(3000, 4, seed=9, rng_mode=mode), dg.sample_walks(3000, 4, seed=9, rng_mode=mode))
    # refreshed source weights when the fields change after set_sources; a later set_sources drops the fixed sum
    od2 = od * 2.0
    for g in (og, dg):
        g.set_sources(srcs); g.set_out_degree(od2); g.build_alias(exact)
    sa, sb = og.get_source_alias(), dg.get_source_alias()
    assert sa["weight_sum"] == sb["weight_sum"] and np.array_equal(bits(sa["prob"]), bits(sb["prob"]))
    with pytest.raises(dge.DgeError):
        dg.set_out_degree(od[:-1])                                    # one value per vertex
