"""Conservation matrix of the commit-lock protocol (dge_selftest_locked_rows): commit 0 = relaxed (policy 5), 1 = strict
(policy 6), 2 = agent-scope release fence.  Run on the GPU box; the numbers quoted in DESIGN.md §5.1 come from here."""
import ctypes as C, sys
sys.path.insert(0, '.')
import embedding_amd as E
for commit in (0, 1, 2):
    for n_rows, workers, iters in ((64, 4096, 20), (256, 12288, 20), (1024, 12288, 40), (16384, 12288, 40), (65536, 12288, 40), (1048576, 12288, 40)):
        total = C.c_int64(0); err = C.c_double(-1)
        rc = E.lib.dge_selftest_locked_rows(0, n_rows, workers, iters, 7, commit, C.byref(total), C.byref(err))
        print("commit", commit, "rows", n_rows, "workers", workers, "rc", rc, "increments", total.value, "max_err", err.value, flush=True)
