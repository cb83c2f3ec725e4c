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
    assert dge.lib.dge_version() == 106


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
    # ... nor may a CHECKER live inside the product: host restatements that tests compare the kernels with belong to oracle/ (or tests/)
    ev = open(os.path.join(ROOT, "embedding_amd", "evaluate.py")).read()
    for name in ("pairwise_estimator", "cosine_distance_matrix", "def dcg_at_k", "def ndcg_at_k", "def ndcg_against("):
        assert name not in ev, "embedding_amd/evaluate.py holds the host restatement %s: it belongs to oracle/quality.py" % name
    # ... and the timed region of bench.py may only meet the oracle inside cpu_baseline
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    where = []
    for fn in ast.walk(tree):
        if isinstance(fn, (ast.FunctionDef, ast.Module)):
            for node in ast.iter_child_nodes(fn):
                for sub in ([node] if isinstance(fn, ast.Module) else ast.walk(node)):
                    if isinstance(sub, (ast.Import, ast.ImportFrom)) and "oracle" in (getattr(sub, "module", None) or "") + " ".join(a.name for a in sub.names):
                        where.append(getattr(fn, "name", "<module>"))
    assert where and set(where) <= {"cpu_baseline", "cpu_epoch_baseline"}, where      # only the reported CPU legs, after the timed regions


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
    assert {"cfg1", "cfg2", "cfg3", "cfg3_zipf", "cfg5"} <= set(bench.WORKLOADS)
    w = bench.WORKLOADS["cfg3"]
    assert w["R"] * w["T"] == 1000008 and w["dim"] == 128 and w["negative"] == 5 and w["L"] == 24      # the configuration the metric is quoted on
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    e = table["cfg3/policy5"]
    assert 6000 < e["bytes_per_pair"] < 7600                                              # the rows of a pair: 7168 algorithmic
    assert bench.measured_traffic("cfg3", "policy5", 1000.0) == e["bytes_per_pair"] * 1000.0
    assert bench.measured_traffic("cfg5", "policy7", 1.0) > 45056 and bench.measured_traffic("cfg3", "hs_centre", 1.0) > 0
    assert bench.measured_traffic("cfg3", "policy99", 1.0) is None
    # requests at the L2's memory side x the pair rate; None where no profile says
    assert abs(bench.measured_requests("cfg3", "policy5", 3.8e8, 400.0) - e["requests_per_pair"] * 3.8e8 / 0.4) < 1e3
    assert bench.measured_requests("cfg3", "policy5", 1.0, 0.0) is None
    # a profile is only quoted for the kernels it was collected with: the entry's stamp against the loaded library's (dge_build_stamp)
    import embedding_amd as E
    stamp = E.engine.build_stamp()
    assert set(stamp) == {"kernels", "sorted"} and all(re.fullmatch(r"[0-9a-f]{12}", v) for v in stamp.values()), stamp
    assert bench.measured_traffic("cfg3", "policy5", 1.0, {"kernels": "0" * 12, "sorted": "0" * 12}) is None
    for key, ent in table.items():
        if isinstance(ent, dict) and ent.get("stamp") == stamp["sorted" if key.endswith("policy8") else "kernels"]:
            wl_, pol_ = key.split("/")
            assert bench.measured_traffic(wl_, pol_, 2.0, stamp) == 2.0 * ent["bytes_per_pair"]
    for name, wl in bench.WORKLOADS.items():                            # a default run of a named workload must find its counters
        if "expect_policy" in wl:
            assert "%s/policy%d" % (name, wl["expect_policy"]) in table, "profiles/traffic.json lacks %s/policy%d" % (name, wl["expect_policy"])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "edges" in base["metric"].lower() or "edges" in json.dumps(base).lower()


# ---------------------------------------------------------------------------------------------- the Java / JNI form (source only: no JDK here)
JAVA = os.path.join(ROOT, "java", "embedding")


def _java(name):
    txt = open(os.path.join(JAVA, name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"//[^\n]*", "", txt)


def test_jni_shim_matches_native_engine_and_the_header():
    """java/jni/dge_jni.cpp: one JNI function per `static native` of NativeEngine.java, every C entry point it calls is declared in
    include/dge.h, and its C++ passes g++ -fsyntax-only against a declaration-only stand-in for jni.h (tests/native/jni_stub — not
    the JDK's header; a real build needs $JAVA_HOME)."""
    import subprocess
    natives = set(re.findall(r"static native [\w\[\]]+ (\w+)\(", _java("NativeEngine.java")))
    shim = open(os.path.join(ROOT, "java", "jni", "dge_jni.cpp")).read()
    assert natives == set(re.findall(r"\bJ\((\w+)\)\(", shim)) and len(natives) >= 18
    called = set(re.findall(r"\b(dge_[a-z0-9_]+)\s*\(", re.sub(r"//[^\n]*", "", shim)))
    assert called <= set(_declared()), called - set(_declared())
    used = set()
    for f in os.listdir(JAVA):
        used |= set(re.findall(r"NativeEngine\.(\w+)\(", _java(f)))
    assert used <= natives, used - natives                          # every native the Java classes call exists
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(ROOT, "tests", "native", "jni_stub"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "java", "jni", "dge_jni.cpp")])


def test_jni_shim_turns_a_missing_device_into_the_runtime_exception(tmp_path):
    """tests/native/jni_shim_test.cpp (the shim's own functions executed through a JNIEnv with a body instead of a JVM; the full run is a GPU test,
    tests/test_gpu_host_mirror.py) builds and links against libdge.so here too; without a gfx950 device its first call, NativeEngine.graphCreate, must come back
    with a null handle and a pending java.lang.RuntimeException carrying dge_last_error() — the product has no CPU path, and a Java caller is told so."""
    import subprocess
    exe = str(tmp_path / "jni_shim_test")
    libdir = os.path.join(ROOT, "embedding_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "tests", "native", "jni_stub"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "jni_shim_test.cpp"), "-o", exe, "-L" + libdir, "-l:libdge.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    if out.returncode == 0:                                         # a box with a GPU: the whole scenario ran
        assert "JNI SHIM OK" in out.stdout, out.stdout + out.stderr
    else:
        assert out.returncode == 3 and "graphCreate threw java/lang/RuntimeException" in out.stdout and "no CPU path" in out.stdout, out.stdout + out.stderr


def test_java_surface_lists_every_member_of_the_scope_table():
    """the public Java member set a drop-in must keep (signatures from J/LayeredGraph.java, J/CrossTimeGraph.java,
    J/SpatialGraph.java, J/DeepWalk.java).  Text check only — javac is not in this image."""
    lg = _java("LayeredGraph.java")
    for decl in ("public static Random rnd", "public static int numLayer", "static public class Edge", "public Vertex from", "public Vertex to",
                 "public double weight", "public Edge(Vertex f, Vertex t, double w)", "static public class Vertex", "public String name",
                 "public int id", "public List<Edge> edgesOut", "public double outDegree", "int[] aliasTable", "double[] probTable",
                 "public Vertex(String n, int i)", "public void addOutEdge(Edge e)", "public void initiateAliasTable()",
                 "public Vertex sampleNextVertex()", "public Vertex sampleNextVertex(double x)", "public List<Edge> allEdges",
                 "public Map<String, Vertex> allVertices", "public List<Vertex> sourceVertices", "protected double sourceWeightSum",
                 "protected double[] probTable", "protected int[] aliasTable", "public LayeredGraph()",
                 "public void addEdge(String fn, String tn, double weight)", "public void addSourceVertex(String vn)",
                 "public void initiateAliasTables()", "public List<String> sampleVertexSequence()"):
        assert decl in lg, decl
    ct = _java("CrossTimeGraph.java")
    for decl in ("public class CrossTimeGraph extends LayeredGraph", "public static int numSamples", "public static int numLayer",
                 "public static CrossTimeGraph constructGraph_tract()", "public static CrossTimeGraph constructGraph_CA()",
                 "public static CrossTimeGraph constructGraph_CA(int[] timeIntervals)",
                 "public static void outputSampleSequence(String regionLevel, int[] timeIntervals)",
                 "public static void outputSampleSequence(String regionLevel)",
                 "public static void sampleSequenceHelper(CrossTimeGraph g, String regionLevel)", "public static void main(String[] argv)"):
        assert decl in ct, decl
    sg = _java("SpatialGraph.java")
    for decl in ("public class SpatialGraph extends LayeredGraph", "public static int numSamples", "public static int numLayer",
                 "public void keepNearestKVertices(int k)", "public static SpatialGraph constructGraph_tract()",
                 "public static SpatialGraph constructGraph_CA()", "public static void outputSampleSequence(String regionLevel)",
                 "public static void main(String[] argv)", "NativeEngine.graphKeepTopK"):
        assert decl in sg, decl
    dw = _java("DeepWalk.java")
    for decl in ("public static int Year", "public static void learnEmbedding() throws Exception",
                 "public static void learnEmbedding(String regionLevel, String spatialGF) throws Exception",
                 "public static void checkInputFile(String regionLevel, String spatialGF)", "public static void main(String[] argv)"):
        assert decl in dw, decl
    # the reference's own test (T/LayeredGraphTest.java:13-43) uses exactly these expressions on the class above
    for use in ("new LayeredGraph.Vertex(", "new LayeredGraph.Edge(", ".addOutEdge(", ".initiateAliasTable()", ".aliasTable[", ".probTable[",
                ".outDegree", ".sampleNextVertex("):
        assert use.strip(".(").split(".")[-1].split("(")[0].split("[")[0] in lg
    for f in os.listdir(JAVA):                                      # balanced braces / parentheses: the cheapest syntax check there is
        t = _java(f)
        t = re.sub(r'"(\\.|[^"\\])*"', '""', t); t = re.sub(r"'(\\.|[^'\\])'", "' '", t)
        assert t.count("{") == t.count("}") and t.count("(") == t.count(")") and t.count("[") == t.count("]"), f


def test_tuning_knobs_of_the_ctypes_view_are_the_headers():
    """dge_set_tuning's knob numbers: include/dge.h's enum (DGE_TUNE_<NAME> = n) and embedding_amd.engine.TUNING_KNOBS (name -> n, what tests and bench.py --tune use)
    must be the same table — a knob added to one side only would set another knob."""
    import embedding_amd as E
    hdr = open(os.path.join(ROOT, "include", "dge.h")).read()
    enum = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"\bDGE_TUNE_([A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    count = enum.pop("count")
    assert enum == E.engine.TUNING_KNOBS, (sorted(set(enum.items()) ^ set(E.engine.TUNING_KNOBS.items())))
    assert sorted(enum.values()) == list(range(count))
