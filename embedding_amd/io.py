"""File formats of the reference pipeline (read/written by the host side; no compute here).

.seq  one walk per line, space-joined "h-regionId" tokens        J/CrossTimeGraph.java:136-137, J/SpatialGraph.java:105-110
.vec  "name v1 .. vD" per vocabulary row, no header              J/DeepWalk.java:82, P/embeddingEvaluation_tract.py:298-300
      LINE-style variant with a "V D" first line                 miscs/taxi_all.txt:1, P/embeddingEvaluation_tract.py:113-117
.od   "src dst w" per line, one file per time slice              J/Tracts.java:236-264, J/CommunityAreas.java:171-186
"""
import numpy as np


def read_seq(paths):
    """-> (walks int32 [n x Lmax] padded with -1, names list): tokens interned in order of first appearance."""
    ids, names, rows = {}, [], []
    for p in ([paths] if isinstance(paths, str) else paths):
        with open(p) as f:
            for line in f:
                tok = line.split()
                if not tok:
                    continue
                r = []
                for t in tok:
                    i = ids.get(t)
                    if i is None:
                        i = len(names); ids[t] = i; names.append(t)
                    r.append(i)
                rows.append(r)
    L = max((len(r) for r in rows), default=1)
    walks = -np.ones((len(rows), L), np.int32)
    for k, r in enumerate(rows):
        walks[k, :len(r)] = r
    return walks, names


def write_seq(path, walks, names, position_prefix=False):
    with open(path, "w") as f:
        for row in walks:
            toks = [("%d-%s" % (j, names[t]) if position_prefix else names[t]) for j, t in enumerate(row) if t >= 0]
            f.write(" ".join(toks) + "\n")


def read_vec(path, header=False):
    """-> (names list, vectors float32 [V x D])."""
    names, vecs = [], []
    with open(path) as f:
        if header:
            f.readline()
        for line in f:
            p = line.split()
            if len(p) < 2:
                continue
            names.append(p[0]); vecs.append([float(x) for x in p[1:]])
    return names, np.array(vecs, np.float32)


def read_od_slices(paths):
    """Per-slice OD files -> layered cross-time edge list (J/CrossTimeGraph.java:36-47): vertex id = h*R + index of the
    region id in ascending order; edge (h, src) -> ((h+1) % T, dst) for every positive flow.  Sources, as the reference
    picks them (:43-47): EVERY layer-0 vertex that exists in the store — one that only occurs as a destination of slice
    T-1 and has no out-edge included (it enters the source alias table with weight 0) — in ascending region order (the
    reference: its region map's order; embedding_host.hpp:constructGraphFromOD does the same)."""
    T = len(paths)
    flows = []
    for h, p in enumerate(paths):
        a = np.loadtxt(p, dtype=np.float64, ndmin=2)
        a = a[a[:, 2] > 0]
        flows.append(a)
    regions = np.unique(np.concatenate([np.concatenate([a[:, 0], a[:, 1]]) for a in flows])).astype(np.int64)
    R = len(regions)
    src = np.concatenate([h * R + np.searchsorted(regions, a[:, 0].astype(np.int64)) for h, a in enumerate(flows)])
    dst = np.concatenate([((h + 1) % T) * R + np.searchsorted(regions, a[:, 1].astype(np.int64)) for h, a in enumerate(flows)])
    w = np.concatenate([a[:, 2] for a in flows])
    present0 = np.unique(np.concatenate([src[src < R], dst[dst < R]]))
    return dict(src=src.astype(np.int32), dst=dst.astype(np.int32), w=w, sources=present0.astype(np.int32), regions=regions,
                R=R, T=T, names=["%d-%d" % (h, r) for h in range(T) for r in regions])
