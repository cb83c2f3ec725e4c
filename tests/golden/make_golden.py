"""Regenerates tests/golden/oracle_small.npz from the CPU oracle (run from the repo root).

These vectors are produced by oracle/dge_oracle.c itself (there is no runnable reference: Java + DL4J jars are absent),
so they pin the ORACLE against accidental change; the reference-derived vectors are layered_graph_test.json,
java_random_kats.json and taxi_all_head.vec.  SGNS values: parity unpinned."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import layered_graph  # noqa: E402
from oracle import oracle as O  # noqa: E402

src, dst, w, sources = layered_graph(R=12, T=4, deg=4, seed=7)
g = O.Graph(); g.add_edges(src, dst, w); g.set_sources(sources); g.build_alias(True)
walks_seq = g.sample_walks(64, 4, seed=2017, rng_mode=0)
walks_str = g.sample_walks(64, 4, seed=2017, rng_mode=1, first_index=5)
a3 = g.get_alias(3)
m0 = O.train_sgns(walks_seq, 48, 8, 4, negative=3, min_count=2, table_size=257, arith=0)
m1 = O.train_sgns(walks_seq, 48, 8, 4, negative=3, min_count=2, table_size=257, arith=1)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz"),
                    src=src, dst=dst, w=w, sources=sources, walks_seq=walks_seq, walks_str=walks_str,
                    alias3=a3["alias"], prob3=a3["prob"], vocab=m0.vocab_ids, counts=m0.counts, table=m0.table(257),
                    syn0_arith0=m0.syn0, syn1_arith0=m0.syn1neg, syn0_arith1=m1.syn0, syn1_arith1=m1.syn1neg,
                    pairs=np.array([m0.pairs]))
print("wrote oracle_small.npz: pairs", m0.pairs, "V", m0.V)
