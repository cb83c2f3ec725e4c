"""Synthetic time-sliced region-flow graphs: the stand-in for the taxi flow maps the reference
builds its cross-time graph from (J/CrossTimeGraph.java:25-52), which are not shipped.

Vertex id = h*R + r ("h-regionId" in the reference's vocabulary); an edge of slice h goes
(h, src) -> ((h+1) % T, dst) with an integer trip count as weight (J/CrossTimeGraph.java:36-39); sources are the
layer-0 vertices (:43-47).  T == 1 gives a static graph whose vertices are all sources (cfg2).
"""
import numpy as np

SEED = 20171106


def flow_graph_numpy(R, T, mean_degree, seed=SEED, sigma=1.0, weight_mean=20.0, dead_end_fraction=0.0):
    """Host generator (tests, small configs).  Returns dict(src, dst, w, sources, n_vertices)."""
    rng = np.random.default_rng(seed)
    V = R * T
    mu = np.log(mean_degree) - 0.5 * sigma * sigma
    deg = np.minimum(np.maximum(rng.lognormal(mu, sigma, V).astype(np.int64), 1), R)
    if dead_end_fraction > 0:
        dead = rng.random(V) < dead_end_fraction
        dead[:R] = False                      # keep every source alive
        deg[dead] = 0
    E = int(deg.sum())
    src = np.repeat(np.arange(V, dtype=np.int64), deg)
    layer = src // R
    dst_region = rng.integers(0, R, E)
    dst = (((layer + 1) % T) * R + dst_region) if T > 1 else dst_region
    w = 1.0 + np.floor(rng.exponential(weight_mean, E))
    sources = np.arange(R, dtype=np.int32)
    return dict(src=src.astype(np.int32), dst=dst.astype(np.int32), w=w.astype(np.float64), sources=sources,
                n_vertices=V, R=R, T=T)


def flow_graph_torch(R, T, mean_degree, device, seed=SEED, sigma=1.0, weight_mean=20.0, dst="uniform"):
    """Device generator (bench configs: 5 M .. 100 M edges never touch the host).  Returns torch tensors.
    dst = "uniform": destination regions drawn uniformly (the configurations as SURVEY.md §8d specifies them: a flat vocabulary);
    dst = "zipf": region of rank r drawn with P ~ 1/(r+1) — popular regions, what real trip data looks like: a skewed vocabulary;
    dst = "community" / "community_zipf": regions form communities of 64 and 80 % of a vertex's flow stays inside its own; the flow that
    leaves goes to a uniform / a Zipf-popular region.  A graph WITH structure: what the link-prediction parity checks (held-out walk steps,
    tests/helpers.py: link_auc_device) need — on uniform destinations nothing generalises and every schedule scores 0.5."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    V = R * T
    mu = float(np.log(mean_degree) - 0.5 * sigma * sigma)
    z = torch.randn(V, generator=g, device=device, dtype=torch.float32)
    deg = torch.exp(mu + sigma * z).to(torch.int64).clamp_(1, R)
    E = int(deg.sum().item())
    src = torch.repeat_interleave(torch.arange(V, device=device, dtype=torch.int32), deg)
    if dst == "zipf":
        ur = torch.rand(E, generator=g, device=device, dtype=torch.float32)
        dst_region = (torch.exp(ur * float(np.log(R + 1.0))) - 1.0).to(torch.int64).clamp_(0, R - 1).to(torch.int32)
        del ur
    elif dst == "uniform":
        dst_region = torch.randint(0, R, (E,), generator=g, device=device, dtype=torch.int32)
    elif dst in ("community", "community_zipf"):
        inside = torch.rand(E, generator=g, device=device) < 0.8
        local = (torch.div(src % R, 64, rounding_mode="floor") * 64 + torch.randint(0, 64, (E,), generator=g, device=device, dtype=torch.int32)).clamp_(max=R - 1)
        if dst == "community_zipf":
            ur = torch.rand(E, generator=g, device=device, dtype=torch.float32)
            anyw = (torch.exp(ur * float(np.log(R + 1.0))) - 1.0).to(torch.int64).clamp_(0, R - 1).to(torch.int32)
            del ur
        else:
            anyw = torch.randint(0, R, (E,), generator=g, device=device, dtype=torch.int32)
        dst_region = torch.where(inside, local.to(torch.int32), anyw)
        del inside, local, anyw
    else:
        raise ValueError("dst must be 'uniform', 'zipf', 'community' or 'community_zipf'")
    if T > 1:
        layer = torch.div(src, R, rounding_mode="floor")
        dst = ((layer + 1) % T) * R + dst_region
    else:
        dst = dst_region
    u = torch.rand(E, generator=g, device=device, dtype=torch.float64).clamp_(min=1e-12)
    w = 1.0 + torch.floor(-weight_mean * torch.log(u))
    sources = np.arange(R, dtype=np.int32)
    return dict(src=src.contiguous(), dst=dst.to(torch.int32).contiguous(), w=w.contiguous(), sources=sources,
                n_vertices=V, n_edges=E, R=R, T=T)


def powerlaw_flow_graph_torch(R, T, n_edges, device, seed=SEED, alpha=2.1, max_degree=1_000_000, weight_mean=20.0):
    """cfg5: power-law out-degree (exponent alpha, capped) AND power-law in-popularity of the destination
    regions, T slices.  Generated on the device in chunks of vertices so that 1 B edges need no 1 B-element temporaries
    beyond the three COO arrays.  Returns torch tensors like flow_graph_torch."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    V = R * T
    # out-degree: Pareto with P(d >= x) ~ x^(1-alpha), rescaled to hit n_edges in expectation, clamped to [1, min(R, max)]
    u = torch.rand(V, generator=g, device=device, dtype=torch.float64).clamp_(min=1e-12)
    raw = u.pow(-1.0 / (alpha - 1.0))
    cap = float(min(R, max_degree))
    scale = n_edges / float(raw.clamp(max=cap).sum().item())
    deg = (raw * scale).clamp_(1.0, cap).to(torch.int64)
    E = int(deg.sum().item())
    src = torch.repeat_interleave(torch.arange(V, device=device, dtype=torch.int32), deg)
    del u, raw
    # destination popularity: region rank r drawn with P ~ (r+1)^-1 (Zipf) through an inverse-CDF on a uniform
    ur = torch.rand(E, generator=g, device=device, dtype=torch.float32)
    dst_region = (torch.exp(ur * float(np.log(R + 1.0))) - 1.0).to(torch.int64).clamp_(0, R - 1).to(torch.int32)
    del ur
    if T > 1:
        layer = torch.div(src, R, rounding_mode="floor")
        dst = (((layer + 1) % T) * R + dst_region).to(torch.int32)
        del layer
    else:
        dst = dst_region
    del dst_region
    uw = torch.rand(E, generator=g, device=device, dtype=torch.float32).clamp_(min=1e-7)
    w = (1.0 + torch.floor(-weight_mean * torch.log(uw))).to(torch.float64)
    del uw
    sources = np.arange(R, dtype=np.int32)
    return dict(src=src.contiguous(), dst=dst.contiguous(), w=w.contiguous(), sources=sources, n_vertices=V, n_edges=E, R=R, T=T)
