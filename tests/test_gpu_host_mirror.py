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


def test_jni_shim_runs_against_the_library_through_a_jvm_less_jnienv(tmp_path, dge):
    """java/jni/dge_jni.cpp — the functions a JVM would call for NativeEngine's `static native` methods — compiled as shipped and EXECUTED against libdge.so on the
    GPU: tests/native/jni_shim_test.cpp gives the declaration-only JNIEnv of tests/native/jni_stub a body (arrays handed out as COPIES, copied back on release unless
    JNI_ABORT, a pending-exception slot) and drives the entry points as java/embedding/LayeredGraph.java and DeepWalk.java do: T/LayeredGraphTest.java:12-44's golden
    vector, walks, w2v.fit(), the vectors and the .vec file — each compared with the same call on the C ABI — and two errors that must arrive as
    java.lang.RuntimeException(dge_last_error()).  No JVM is involved (none exists in this image or on the GPU boxes: INTEGRATION.md)."""
    exe = str(tmp_path / "jni_shim_test")
    libdir = os.path.join(ROOT, "embedding_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "tests", "native", "jni_stub"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "jni_shim_test.cpp"), "-o", exe, "-L" + libdir, "-l:libdge.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "JNI SHIM OK" in out.stdout and "keepTopK(2) threw java/lang/RuntimeException" in out.stdout, out.stdout + out.stderr
