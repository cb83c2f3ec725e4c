import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/dge_oracle.c."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def algos_harness():
    """Host build of embedding_amd/csrc/dge_algos.h (the per-lane scalar logic of the HIP kernels)."""
    import ctypes as C
    d = os.path.join(ROOT, "tests", "native")
    so = os.path.join(d, "libalgos_harness.so")
    srcs = [os.path.join(d, "algos_harness.cpp"), os.path.join(ROOT, "embedding_amd", "csrc", "dge_algos.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-o", so, srcs[0]])
    H = C.CDLL(so)
    H.harness_alias_reference.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
    H.harness_alias_vose.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
    H.harness_jr_jump.argtypes = [C.c_int64, C.c_uint64]; H.harness_jr_jump.restype = C.c_uint64
    H.harness_jr_next_double.argtypes = [C.POINTER(C.c_uint64)]; H.harness_jr_next_double.restype = C.c_double
    H.harness_mix64.argtypes = [C.c_uint64]; H.harness_mix64.restype = C.c_uint64
    H.harness_w2v_jump.argtypes = [C.c_uint64, C.c_uint64]; H.harness_w2v_jump.restype = C.c_uint64
    H.harness_stream_sum.argtypes = [C.c_void_p, C.c_int64]; H.harness_stream_sum.restype = C.c_double
    H.harness_bitset_selftest.argtypes = [C.c_int64, C.c_uint64, C.c_int64]; H.harness_bitset_selftest.restype = C.c_int64
    H.harness_huffman.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return H


@pytest.fixture(scope="session")
def dge():
    """The product package; on a GPU box the HIP library must be the thing that runs."""
    import embedding_amd
    return embedding_amd
