import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import embedding_amd as E
from oracle import oracle as O
from helpers import layered_graph, build_both, cosine_rows
src, dst, w, sources = layered_graph(R=40, T=6, deg=5, seed=0)
og, dg = build_both(O, E, src, dst, w, sources)
walks = dg.sample_walks(1500, 6, seed=11, rng_mode=1); NV = 240
for D in (64, 128, 32):
    om = O.train_sgns(walks, NV, D, 6, table_size=20011, arith=0)
    for pol in (4, 5):
        for workers in (1, 2, 16):
            dm = E.SgnsModel.fit(walks, E.make_config(D, 6, NV, workers=workers, table_size=20011, update_policy=pol), 0)
            s0, vid = dm.vectors(); s1 = dm.syn1neg()
            print("D", D, "pol", pol, "workers", workers, "pairs", dm.stats()["pairs"], om.pairs, "min cos syn0 %.6f syn1 %.6f" % (cosine_rows(s0, om.syn0).min(), cosine_rows(s1, om.syn1neg).min()),
                  "maxabs %.2e" % np.abs(s0 - om.syn0).max(), flush=True)
