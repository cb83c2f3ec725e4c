import ctypes as C, os, sys, subprocess
sys.path.insert(0, '.')
mode = os.environ.get("DGE_SELFTEST_MODE", "0")
import embedding_amd as E
for n_rows, workers, iters in ((64, 4096, 20), (256, 12288, 20), (1024, 12288, 40), (16384, 12288, 40), (65536, 12288, 40), (1048576, 12288, 40)):
    total = C.c_int64(0); err = C.c_double(-1)
    rc = E.lib.dge_selftest_locked_rows(0, n_rows, workers, iters, 7, C.byref(total), C.byref(err))
    print("mode", mode, "rows", n_rows, "workers", workers, "rc", rc, "increments", total.value, "max_err", err.value, flush=True)
