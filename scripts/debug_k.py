import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import embedding_amd as E
from oracle import oracle as O
from helpers import layered_graph, build_both, bits
for R in (40,):
    src, dst, w, sources = layered_graph(R=R, T=6, deg=5, seed=0)
    og, dg = build_both(O, E, src, dst, w, sources)
    walks = dg.sample_walks(600, 6, seed=11, rng_mode=1)
    NV = R*6
    for K,dbg in ((11,0),(15,0),(20,0),(11,1),(15,1),(20,1),(11,2),(15,2),(20,2)):
        om = O.train_sgns(walks, NV, 32, 6, negative=K, table_size=20011, arith=1)
        cfg = E.make_config(32, 6, NV, negative=K, workers=1, table_size=20011); cfg.reserved = dbg
        dm = E.SgnsModel.fit(walks, cfg, 0)
        s0, vid = dm.vectors(); s1 = dm.syn1neg()
        d0 = (bits(s0) != bits(om.syn0)).any(1).sum(); d1 = (bits(s1) != bits(om.syn1neg)).any(1).sum()
        print("dbg", dbg, "K", K, "V", om.V, "rows differing syn0", d0, "syn1neg", d1, "maxabs", np.abs(s0-om.syn0).max(), flush=True)
