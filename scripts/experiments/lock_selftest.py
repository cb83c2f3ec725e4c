"""The commit-lock protocols in isolation (dge_selftest_locked_rows): lost increments and time, lock array (0 relaxed, 1 strict) against
row-embedded locks (3 relaxed, 4 strict) — needs scripts/experiments/embedded_locks.patch applied.   python scripts/experiments/lock_selftest.py"""
import ctypes as C, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from embedding_amd import _native as dge

def run(n_rows, workers, iters, commit):
    total = C.c_int64(0); err = C.c_double(0)
    t0 = time.time()
    rc = dge.lib.dge_selftest_locked_rows(0, n_rows, workers, iters, 7, commit, C.byref(total), C.byref(err))
    return rc, total.value, err.value, time.time() - t0

run(1024, 1024, 2, 0)
for n_rows in (256, 1024, 65536, 1000000):
    for commit in (0, 3, 1, 4):
        rc, total, err, dt = run(n_rows, 12288, 400, commit)
        print("rows %8d  commit %d  rc %d  increments %9d  worst row off by %g  %.3f s" % (n_rows, commit, rc, total, err, dt), flush=True)
