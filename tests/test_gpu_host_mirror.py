"""GPU: the C++ host mirror of LayeredGraph / CrossTimeGraph / SpatialGraph / DeepWalk (embedding_amd/host) — a port of
T/LayeredGraphTest.java plus the .seq -> .vec plumbing, compiled with g++ against libdge.so and run as a program."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(tmp_path, dge):
    exe = str(tmp_path / "host_mirror_test")
    libdir = os.path.join(ROOT, "embedding_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "native", "host_mirror_test.cpp"), "-o", exe,
                           "-L" + libdir, "-l:libdge.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "HOST MIRROR OK" in out.stdout, out.stdout + out.stderr
