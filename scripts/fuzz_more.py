"""One-off: the seeded sweeps of tests/test_gpu_fuzz.py over many more seeds (run on the GPU box)."""
import os, sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as dge
from oracle import oracle
import test_gpu_fuzz as F


class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k): os.environ.pop(k, None)


lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    for big in (False, True):
        try:
            os.environ.pop("DGE_FORCE_BIG", None); os.environ.pop("DGE_BIG_SEG_SHIFT", None)
            F.test_random_configuration_bit_exact(dge, oracle, -seed if big else seed, MP())
        except Exception:
            bad += 1; print("TRAINER seed", seed, "big", big, "FAILED"); traceback.print_exc(limit=2)
    os.environ.pop("DGE_FORCE_BIG", None); os.environ.pop("DGE_BIG_SEG_SHIFT", None)
    try:
        F.test_random_graph_alias_and_walks_bit_exact(dge, oracle, seed)
    except Exception:
        bad += 1; print("GRAPH seed", seed, "FAILED"); traceback.print_exc(limit=2)
print("seeds", lo, "..", hi - 1, "failures:", bad)
