"""The vector file's number formatter (embedding_amd/csrc/fmt_g9.h, used by dge_write_vec = WordVectorSerializer.writeWordVectors, J/DeepWalk.java:82): host code,
checked here without a GPU against the C library's printf."""
import ctypes as C

import numpy as np


def test_fmt_g9_matches_printf_on_random_floats():
    from embedding_amd import _native
    lib = _native.lib
    fast = C.c_int64(0); bad = C.c_int64(-1)
    assert lib.dge_selftest_fmt_g9(4_000_000, 7, C.byref(fast), C.byref(bad)) == 0
    assert bad.value == 0
    assert fast.value > 2_500_000           # every value of an embedding's range and a fifth of the random bit patterns take the fast path
