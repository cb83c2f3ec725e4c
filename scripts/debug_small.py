import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import embedding_amd as E
from oracle import oracle as O
from helpers import layered_graph, build_both, cosine_rows
src, dst, w, sources = layered_graph(R=400, T=6, deg=5, seed=0)
og, dg = build_both(O, E, src, dst, w, sources)
walks = dg.sample_walks(30000, 6, seed=11, rng_mode=1); NV = 2400
om = O.train_sgns(walks, NV, 32, 6, table_size=20011, arith=1)
o8 = O.train_sgns(walks, NV, 32, 6, table_size=20011, arith=1, threads=8)
print("oracle 8 threads vs seq: median cos", np.median(cosine_rows(o8.syn0, om.syn0)))
for pol in (2, 1):
    for workers in (16, 64, 256, 1024, 4096, 0):
        dm = E.SgnsModel.fit(walks, E.make_config(32, 6, NV, workers=workers, table_size=20011, update_policy=pol), 0)
        s0, vid = dm.vectors(); st = dm.stats()
        print("pol", pol, "workers", workers, "median cos vs seq %.4f" % np.median(cosine_rows(s0, om.syn0)), "kernel_ms %.1f" % st["kernel_ms"], flush=True)
