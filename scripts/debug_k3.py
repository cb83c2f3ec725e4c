import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import embedding_amd as E
from oracle import oracle as O
from helpers import layered_graph, build_both, bits
for (R,seed,K,D,n) in ((40,0,11,32,600),(40,1,11,32,600),(40,0,12,32,600),(40,0,15,32,600),(40,0,15,64,600),(20,0,5,32,3000),(20,0,10,32,3000),(40,0,10,32,3000)):
    src, dst, w, sources = layered_graph(R=R, T=6, deg=5, seed=seed)
    og, dg = build_both(O, E, src, dst, w, sources)
    walks = dg.sample_walks(n, 6, seed=11, rng_mode=1)
    NV=R*6
    om = O.train_sgns(walks, NV, D, 6, negative=K, table_size=20011, arith=1)
    dm = E.SgnsModel.fit(walks, E.make_config(D, 6, NV, negative=K, workers=1, table_size=20011), 0)
    s0,vid = dm.vectors(); s1=dm.syn1neg()
    d0=np.argwhere(bits(s0)!=bits(om.syn0)); d1=np.argwhere(bits(s1)!=bits(om.syn1neg))
    print("R",R,"seed",seed,"K",K,"D",D,"n",n,"V",om.V,"pairs",om.pairs,"| syn0 diff elems",len(d0),"rows",len(set(d0[:,0].tolist())),"| syn1 diff elems",len(d1),"rows",len(set(d1[:,0].tolist())))
    for (r,c) in d1[:6]:
        print("    syn1",r,c, s1[r,c], om.syn1neg[r,c], "ulps", int(bits(s1)[r,c])-int(bits(om.syn1neg)[r,c]))
    for (r,c) in d0[:4]:
        print("    syn0",r,c, s0[r,c], om.syn0[r,c], "ulps", int(bits(s0)[r,c])-int(bits(om.syn0)[r,c]))
