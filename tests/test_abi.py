"""CPU: the C-ABI library loads and exports every symbol include/dge.h declares; without a GPU every entry that
needs the device fails loudly (there is no CPU path in the product)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "dge.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(dge_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol(dge):
    names = _declared()
    assert len(names) >= 40
    raw = C.CDLL(dge.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libdge.so does not export %s" % n
    from embedding_amd._native import SIGNATURES
    assert sorted(SIGNATURES) == names, set(names) ^ set(SIGNATURES)     # the ctypes view binds exactly the header


def test_struct_layouts(dge):
    assert C.sizeof(dge.TrainConfig) == 64 and C.sizeof(dge.TrainStats) == 40
    assert dge.lib.dge_version() == 101


def test_no_device_means_loud_failure(dge):
    n = C.c_int(0)
    dge.lib.dge_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible: the no-device behaviour cannot be observed here")
    with pytest.raises(dge.DgeError) as ei:
        dge.DeviceGraph(0)
    assert ei.value.code == 6 and "no CPU path" in str(ei.value)
    with pytest.raises(dge.DgeError):
        dge.WalkCorpus.from_host(np.zeros((2, 3), np.int32), 0)
    with pytest.raises(dge.DgeError):
        dge.SgnsModel.fit(np.zeros((2, 3), np.int32), dge.make_config(8, 3, 4), 0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under embedding_amd/ may reference it."""
    for d, _, files in os.walk(os.path.join(ROOT, "embedding_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "dge_oracle" not in txt.replace("oracle/dge_oracle.c", ""), f


def test_cpp_host_mirror_compiles_against_the_abi(tmp_path, dge):
    """embedding_amd/host/embedding_host.hpp (the C++ mirror of LayeredGraph/CrossTimeGraph/SpatialGraph/DeepWalk) and its
    test program build and link against libdge.so with plain g++; running it needs a GPU (tests/test_gpu_host_mirror.py)."""
    import subprocess
    libdir = os.path.join(ROOT, "embedding_amd")
    exe = str(tmp_path / "host_mirror_test")
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "native", "host_mirror_test.cpp"), "-o", exe,
                           "-L" + libdir, "-l:libdge.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)


def test_bench_workloads_and_traffic_table():
    """bench.py's workload table names BASELINE.json's configurations, and profiles/traffic.json answers for the schedules the
    default runs resolve to (the roofline.traffic field of the bench line)."""
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert {"cfg1", "cfg2", "cfg3", "cfg5"} <= set(bench.WORKLOADS)
    w = bench.WORKLOADS["cfg3"]
    assert w["R"] * w["T"] == 1000008 and w["dim"] == 128 and w["negative"] == 5 and w["L"] == 24      # the configuration the metric is quoted on
    assert bench.measured_traffic("cfg3", "policy5", 1000.0) == 7319.0 * 1000.0
    assert bench.measured_traffic("cfg5", "policy7", 1.0) > 45056 and bench.measured_traffic("cfg3", "hs", 1.0) > 0
    assert bench.measured_traffic("cfg3", "policy99", 1.0) is None
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "edges" in base["metric"].lower() or "edges" in json.dumps(base).lower()
