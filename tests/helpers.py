"""Shared builders for the parity tests: the same seeded inputs go to the oracle and to libdge."""
import numpy as np


def layered_graph(R=40, T=4, deg=6, seed=0, shuffle=True, duplicates=True, dead_ends=0.0):
    """Small cross-time graph in INSERTION order that is not sorted by source (as J/CrossTimeGraph.java:33-41
    would produce it, then shuffled) with duplicate (src,dst) pairs kept (J/LayeredGraph.java:157-174)."""
    rng = np.random.default_rng(seed)
    src, dst, w = [], [], []
    for h in range(T):
        for s in range(R):
            if dead_ends > 0 and h > 0 and rng.random() < dead_ends:
                continue
            k = int(rng.integers(1, deg + 1))
            for d in rng.integers(0, R, k):
                src.append(h * R + s); dst.append(((h + 1) % T) * R + int(d)); w.append(float(rng.integers(1, 60)))
            if duplicates and rng.random() < 0.3:
                src.append(h * R + s); dst.append(dst[-1]); w.append(float(rng.integers(1, 60)))
    src = np.array(src, np.int32); dst = np.array(dst, np.int32); w = np.array(w, np.float64)
    if shuffle:
        p = rng.permutation(len(src))
        src, dst, w = src[p], dst[p], w[p]
    present = np.zeros(R * T, bool); present[src] = True; present[dst] = True
    sources = np.array([v for v in range(R) if present[v]], np.int32)
    return src, dst, w, sources


def build_both(O, E, src, dst, w, sources, exact=True, stream_sum=False, top_k=None):
    og = O.Graph(); og.add_edges(src, dst, w)
    dg = E.DeviceGraph(0); dg.add_edges(src, dst, w)
    if top_k is not None:
        og.keep_top_k(top_k); dg.keep_top_k(top_k)
    og.set_sources(sources, stream_sum); dg.set_sources(sources, stream_sum)
    og.build_alias(exact); dg.build_alias(exact)
    return og, dg


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int64 if a.dtype == np.float64 else np.int32)


def cosine_rows(a, b):
    num = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    den = np.linalg.norm(a.astype(np.float64), axis=1) * np.linalg.norm(b.astype(np.float64), axis=1)
    return num / np.maximum(den, 1e-300)


def simulate_block_schedule(models, train_fn):
    """The multi-GPU block schedule (embedding_amd/distributed.py: block_schedule_step) with all ranks on ONE device:
    models[g] plays rank g; the all-gather between episodes becomes direct export/import between the models."""
    import torch
    N = len(models)
    pf = models[0].partition_floats(N)
    bufs = [torch.empty(pf, dtype=torch.float32, device=models[0].torch_device) for _ in range(N)]
    for e in range(N):
        for g, m in enumerate(models):
            m.set_partition(N, g, (g + e) % N)
            train_fn(m)
        for g, m in enumerate(models):
            m.export_partition(1, N, (g + e) % N, bufs[g])
        for g, m in enumerate(models):
            for r in range(N):
                if r != g:
                    m.import_partition(1, N, (r + e) % N, bufs[r])
    for g, m in enumerate(models):
        m.set_partition(1)


def simulate_gather_syn0(models):
    import torch
    N = len(models)
    pf = models[0].partition_floats(N)
    bufs = [torch.empty(pf, dtype=torch.float32, device=models[0].torch_device) for _ in range(N)]
    for g, m in enumerate(models):
        m.export_partition(0, N, g, bufs[g])
    for g, m in enumerate(models):
        for r in range(N):
            if r != g:
                m.import_partition(0, N, r, bufs[r])


def link_auc(syn0, syn1neg, vocab_ids, test_walks, R, seed=3):
    """Link prediction on held-out walk steps: AUC of syn0[next] . syn1neg[current] for true next vertices against a random region of
    the same slice (the statistical parity measure of scripts/quality_scale.py, on the host)."""
    rng = np.random.default_rng(seed)
    NV = int(max(int(vocab_ids.max()), int(test_walks.max())) + 1)
    remap = -np.ones(NV + R, np.int64); remap[vocab_ids.astype(np.int64)] = np.arange(len(vocab_ids))
    a = test_walks[:, :-1].reshape(-1).astype(np.int64); b = test_walks[:, 1:].reshape(-1).astype(np.int64)
    ok = (a >= 0) & (b >= 0); a, b = a[ok], b[ok]
    rnd = (b // R) * R + rng.integers(0, R, len(b))
    ra, rb, rr = remap[a], remap[b], remap[np.minimum(rnd, NV - 1)]
    ok = (ra >= 0) & (rb >= 0) & (rr >= 0); ra, rb, rr = ra[ok], rb[ok], rr[ok]
    pos = (syn0[rb].astype(np.float64) * syn1neg[ra]).sum(1); neg = (syn0[rr].astype(np.float64) * syn1neg[ra]).sum(1)
    return float((pos > neg).mean() + 0.5 * (pos == neg).mean())
