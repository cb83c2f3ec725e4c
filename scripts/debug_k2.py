import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import embedding_amd as E
from oracle import oracle as O
from helpers import layered_graph, build_both, bits
R=40
src, dst, w, sources = layered_graph(R=R, T=6, deg=5, seed=0)
og, dg = build_both(O, E, src, dst, w, sources)
walks_all = dg.sample_walks(600, 6, seed=11, rng_mode=1)
NV=R*6; K=11
def run(walks):
    om = O.train_sgns(walks, NV, 32, 6, negative=K, table_size=20011, arith=1, min_count=1)
    dm = E.SgnsModel.fit(walks, E.make_config(32, 6, NV, negative=K, workers=1, table_size=20011, min_count=1), 0)
    s0,vid = dm.vectors(); s1=dm.syn1neg()
    return om, s0, s1
om,s0,s1 = run(walks_all)
print("full diff rows", (bits(s0)!=bits(om.syn0)).any(1).sum())
for n in (10,20,40,80,160,320,600):
    om,s0,s1 = run(walks_all[:n])
    d0=np.argwhere(bits(s0)!=bits(om.syn0)); d1=np.argwhere(bits(s1)!=bits(om.syn1neg))
    print(n, "V", om.V, "syn0 diffs", len(d0), "syn1 diffs", len(d1), d0[:5].tolist(), d1[:5].tolist(), flush=True)
    if len(d0)+len(d1):
        r,c = (d1[0] if len(d1) else d0[0])
        print(" example", s1[r,c], om.syn1neg[r,c], "row", r, "counts", om.counts[r])
