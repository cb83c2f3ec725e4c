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


def simulate_block_schedule(models, train_fn, serial=False):
    """The multi-GPU block schedule (embedding_amd/distributed.py: block_schedule_step) with all ranks on ONE device:
    models[g] plays rank g; the all-gather between episodes becomes direct export/import between the models."""
    import torch
    N = len(models)
    pf = models[0].partition_floats(N)
    bufs = [torch.empty(pf, dtype=torch.float32, device=models[0].torch_device) for _ in range(N)]
    for e in range(N):
        for g, m in enumerate(models):
            m.set_partition(N, g, (g + e) % N)
            if train_fn.__code__.co_argcount >= 2:
                train_fn(m, e)              # (the episode number: learning-rate position by progress)
            else:
                train_fn(m)
            if serial:
                m.stats()                   # drains this model's stream: the ranks' launches of an episode run one after the other, as they would on N devices of their own
        for table in ((1, 2) if getattr(getattr(models[0], "cfg", None), "use_hs", 0) else (1,)):      # syn1 partitions travel with the syn1neg partitions (use_hs)
            for g, m in enumerate(models):
                m.export_partition(table, N, (g + e) % N, bufs[g])
            for g, m in enumerate(models):
                for r in range(N):
                    if r != g:
                        m.import_partition(table, N, (r + e) % N, bufs[r])
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


# ---- device-side helpers of the full-size statistical tests (tables of a million rows never visit the host)
def device_table(model, table, n_parts=1, part=0):
    """Rows of partition `part` (row % n_parts == part; n_parts = 1: the whole table) of syn0 (0) / syn1neg (1) as a torch tensor
    [rows x stride] on the model's device — through the C ABI's partition export, device to device."""
    import torch
    pf = model.partition_floats(n_parts)
    buf = torch.empty(pf, dtype=torch.float32, device=model.torch_device)
    model.export_partition(table, n_parts, part, buf)
    stride = -(-model.cfg.dim // 64) * 64
    return buf.view(-1, stride)


def link_scores_device(model, vocab_ids, test_walks, R, NV, seed=3):
    """Scores syn0[next] . syn1neg[current] of the held-out walk steps of `test_walks` (torch int64 [n x L] on the device) and of a random
    region of the next vertex's slice instead of it -> (pos, neg) score tensors."""
    import torch
    dev = model.torch_device
    s0 = device_table(model, 0); s1 = device_table(model, 1)
    vid = torch.from_numpy(vocab_ids.astype(np.int64)).to(dev)
    remap = -torch.ones(NV, dtype=torch.int64, device=dev); remap[vid] = torch.arange(len(vid), device=dev)
    a = test_walks[:, :-1].reshape(-1); b = test_walks[:, 1:].reshape(-1)
    ok = (a >= 0) & (b >= 0); a, b = a[ok], b[ok]
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    rnd = torch.div(b, R, rounding_mode="floor") * R + torch.randint(0, R, (len(b),), generator=gen, device=dev)
    ra, rb, rr = remap[a], remap[b], remap[rnd]
    ok = (ra >= 0) & (rb >= 0) & (rr >= 0); ra, rb, rr = ra[ok], rb[ok], rr[ok]
    pos = torch.empty(len(ra), dtype=torch.float32, device=dev); neg = torch.empty_like(pos)
    for i in range(0, len(ra), 1 << 20):                      # in slices: the gathered rows of 4.6 M steps would be gigabytes
        j = slice(i, i + (1 << 20))
        x = s1[ra[j]]
        pos[j] = (s0[rb[j]] * x).sum(1); neg[j] = (s0[rr[j]] * x).sum(1)
    return pos, neg


def link_auc_device(model, vocab_ids, test_walks, R, NV, seed=3):
    """Link-prediction AUC on held-out walk steps (helpers.link_auc, scripts/quality_scale.py) computed on the device, and the mean
    negative-sampling loss of the same (positive, random-region) score pairs."""
    import torch
    pos, neg = link_scores_device(model, vocab_ids, test_walks, R, NV, seed)
    auc = float((pos > neg).float().mean() + 0.5 * (pos == neg).float().mean())
    loss = float(torch.nn.functional.softplus(-pos.double()).mean() + torch.nn.functional.softplus(neg.double()).mean())
    return auc, loss
