// The JNI shim (java/jni/dge_jni.cpp) EXECUTED without a JVM: this file gives the declaration-only JNIEnv of tests/native/jni_stub/jni.h a body — arrays are heap
// objects, Get<T>ArrayElements hands out a COPY (as a JVM may: isCopy), Release copies it back unless JNI_ABORT, ThrowNew parks a pending exception — includes the
// shim's source, and drives its Java_embedding_NativeEngine_* entry points the way java/embedding/LayeredGraph.java and DeepWalk.java do: the golden vector of
// T/LayeredGraphTest.java:12-44, walks, w2v.fit(), the vectors, the .vec file, an error turned into a RuntimeException.  Every result is compared with the same call
// made directly on the C ABI.  What this proves: the shim's marshalling (which arrays are copied back, which are not; null arrays; lengths; handles; error
// mapping) on the real library and a real GPU.  What it does not: anything about a JVM (no JDK in this image or on the GPU boxes).
// Built and run by tests/test_gpu_host_mirror.py (GPU) — on a box without a device the first call must come back as the exception (tests/test_abi.py runs that).
#include <jni.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------- the fake JVM side
struct FakeObj : _jobject {
    enum Kind { INT, LONG, DOUBLE, FLOAT, OBJ, STR, CLS } kind;
    std::vector<jint> i; std::vector<jlong> l; std::vector<jdouble> d; std::vector<jfloat> f; std::vector<jobject> o; std::string s;
    explicit FakeObj(Kind k) : kind(k) {}
};
static FakeObj* fo(jobject x) { return static_cast<FakeObj*>(x); }
static std::string g_pending;            // message of the pending exception ("" = none)
static std::string g_pending_class;
static long g_outstanding = 0;           // element buffers handed out and not yet released
static long g_copied_back = 0, g_aborted = 0;

jclass JNIEnv::FindClass(const char* name) { FakeObj* c = new FakeObj(FakeObj::CLS); c->s = name; return c; }
jint JNIEnv::ThrowNew(jclass c, const char* msg) { g_pending = msg ? msg : "(null)"; g_pending_class = fo(c)->s; if (g_pending.empty()) g_pending = "(empty)"; return 0; }
jsize JNIEnv::GetArrayLength(jarray a) {
    FakeObj* x = fo(a);
    switch (x->kind) {
        case FakeObj::INT: return (jsize)x->i.size();
        case FakeObj::LONG: return (jsize)x->l.size();
        case FakeObj::DOUBLE: return (jsize)x->d.size();
        case FakeObj::FLOAT: return (jsize)x->f.size();
        case FakeObj::OBJ: return (jsize)x->o.size();
        default: return 0;
    }
}
template <class T> static T* hand_out(const std::vector<T>& v, jboolean* is_copy) {
    if (is_copy) *is_copy = 1;
    T* p = new T[v.size() + 1];
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    g_outstanding++;
    return p;
}
template <class T> static void take_back(std::vector<T>& v, T* p, jint mode) {
    if (mode != JNI_ABORT) { if (!v.empty()) std::memcpy(v.data(), p, v.size() * sizeof(T)); g_copied_back++; } else g_aborted++;
    delete[] p;
    g_outstanding--;
}
jint* JNIEnv::GetIntArrayElements(jintArray a, jboolean* c) { return hand_out(fo(a)->i, c); }
void JNIEnv::ReleaseIntArrayElements(jintArray a, jint* p, jint mode) { take_back(fo(a)->i, p, mode); }
jlong* JNIEnv::GetLongArrayElements(jlongArray a, jboolean* c) { return hand_out(fo(a)->l, c); }
void JNIEnv::ReleaseLongArrayElements(jlongArray a, jlong* p, jint mode) { take_back(fo(a)->l, p, mode); }
jdouble* JNIEnv::GetDoubleArrayElements(jdoubleArray a, jboolean* c) { return hand_out(fo(a)->d, c); }
void JNIEnv::ReleaseDoubleArrayElements(jdoubleArray a, jdouble* p, jint mode) { take_back(fo(a)->d, p, mode); }
jdoubleArray JNIEnv::NewDoubleArray(jsize n) { FakeObj* x = new FakeObj(FakeObj::DOUBLE); x->d.assign((size_t)n, 0.0); return x; }
void JNIEnv::SetDoubleArrayRegion(jdoubleArray a, jsize at, jsize n, const jdouble* src) { std::memcpy(fo(a)->d.data() + at, src, (size_t)n * sizeof(jdouble)); }
jfloatArray JNIEnv::NewFloatArray(jsize n) { FakeObj* x = new FakeObj(FakeObj::FLOAT); x->f.assign((size_t)n, 0.0f); return x; }
void JNIEnv::SetFloatArrayRegion(jfloatArray a, jsize at, jsize n, const jfloat* src) { std::memcpy(fo(a)->f.data() + at, src, (size_t)n * sizeof(jfloat)); }
void JNIEnv::SetIntArrayRegion(jintArray a, jsize at, jsize n, const jint* src) { std::memcpy(fo(a)->i.data() + at, src, (size_t)n * sizeof(jint)); }
jobject JNIEnv::GetObjectArrayElement(jobjectArray a, jsize k) { return fo(a)->o[(size_t)k]; }
const char* JNIEnv::GetStringUTFChars(jstring s, jboolean* c) { if (c) *c = 0; g_outstanding++; return fo(s)->s.c_str(); }
void JNIEnv::ReleaseStringUTFChars(jstring, const char*) { g_outstanding--; }

// ---------------------------------------------------------------------------------------------- the shim itself, as shipped
#include "../../java/jni/dge_jni.cpp"

static jintArray ints(const std::vector<jint>& v) { FakeObj* x = new FakeObj(FakeObj::INT); x->i = v; return x; }
static jlongArray longs(size_t n) { FakeObj* x = new FakeObj(FakeObj::LONG); x->l.assign(n, 0); return x; }
static jdoubleArray doubles(const std::vector<jdouble>& v) { FakeObj* x = new FakeObj(FakeObj::DOUBLE); x->d = v; return x; }
static jstring str(const std::string& s) { FakeObj* x = new FakeObj(FakeObj::STR); x->s = s; return x; }

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s   (pending exception: %s)\n", __FILE__, __LINE__, #c, g_pending.c_str()); return 1; } } while (0)
#define NO_EXCEPTION() CHECK(g_pending.empty())

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    JNIEnv env; JNIEnv* e = &env; jclass cls = nullptr;

    jlong g = J(graphCreate)(e, cls, 0);
    if (!g_pending.empty()) {            // no gfx950 device: the status became the exception NativeEngine's callers see (and the handle stayed null)
        std::printf("JNI SHIM: graphCreate threw %s: %s\n", g_pending_class.c_str(), g_pending.c_str());
        return g == 0 && g_pending_class == "java/lang/RuntimeException" ? 3 : 1;
    }
    CHECK(g != 0);
    {   // T/LayeredGraphTest.java:12-44 through the entry points LayeredGraph.upload() / Vertex.initiateAliasTable() call: start -> d1 (2), d2 (10), d3 (8)
        J(graphAddEdges)(e, cls, g, ints({0, 0, 0}), ints({1, 2, 3}), doubles({2, 10, 8}), 3); NO_EXCEPTION();
        J(graphSetSources)(e, cls, g, ints({0}), 1, 0); NO_EXCEPTION();
        J(graphBuildAlias)(e, cls, g, 1); NO_EXCEPTION();
        jdoubleArray prob = doubles({0, 0, 0, 0}); jintArray alias = ints({7, 7, 7, 7}), nbr = ints({7, 7, 7, 7});
        jdoubleArray kd = J(graphGetAlias)(e, cls, g, 0, prob, alias, nbr); NO_EXCEPTION();
        CHECK(kd && fo(kd)->d.size() == 2 && fo(kd)->d[0] == 3.0 && fo(kd)->d[1] == 20.0);                       // degree, outDegree
        CHECK(fo(prob)->d[0] == 0.3 && fo(prob)->d[1] == 0.8 && fo(prob)->d[2] == 1.0);                            // copied BACK into the Java arrays (mode 0)
        CHECK(fo(alias)->i[0] == 1 && fo(alias)->i[1] == 2 && fo(alias)->i[2] == -1 && fo(alias)->i[3] == 7);
        CHECK(fo(nbr)->i[0] == 1 && fo(nbr)->i[1] == 2 && fo(nbr)->i[2] == 3);
        const double xs[5] = {0.05, 0.3, 0.4, 0.65, 0.9}; const int want[5] = {1, 2, 2, 3, 3};
        for (int k = 0; k < 5; k++) { CHECK(J(graphSampleNext)(e, cls, g, 0, xs[k]) == want[k]); NO_EXCEPTION(); }
        CHECK(J(graphSampleNext)(e, cls, g, 1, 0.5) == -1); NO_EXCEPTION();                                        // dead end
        jdoubleArray sp = doubles({0}); jintArray sa = ints({7});
        J(graphGetSourceAlias)(e, cls, g, sp, sa); NO_EXCEPTION();
        CHECK(fo(sp)->d[0] == 1.0 && fo(sa)->i[0] == -1);
        // graphGetCsr with some arrays null (LayeredGraph reads back only what it mirrors)
        jlongArray rp = longs(5); jintArray nb = ints({7, 7, 7});
        J(graphGetCsr)(e, cls, g, rp, nb, nullptr, nullptr, nullptr, nullptr); NO_EXCEPTION();
        CHECK(fo(rp)->l[0] == 0 && fo(rp)->l[1] == 3 && fo(rp)->l[4] == 3 && fo(nb)->i[0] == 1 && fo(nb)->i[2] == 3);
        // an error becomes java.lang.RuntimeException(dge_last_error()): keepNearestKVertices(2) on a graph whose d1 has no out-edge (J/SpatialGraph.java:29-35 would throw too)
        J(graphKeepTopK)(e, cls, g, 2);
        CHECK(!g_pending.empty() && g_pending_class == "java/lang/RuntimeException" && g_pending == dge_last_error());
        std::printf("JNI SHIM: keepTopK(2) threw %s: %s\n", g_pending_class.c_str(), g_pending.c_str());
        g_pending.clear();
        J(graphFree)(e, cls, g);
    }
    {   // a 4-slice layered graph: sampleVertexSequences -> w2v.fit() -> vectors -> .vec, each against the same call on the C ABI
        const int R = 40, T = 4, L = 4, NV = R * T, D = 20;
        std::vector<jint> src, dst; std::vector<jdouble> w;
        uint64_t s = 12345;
        for (int h = 0; h < T; h++)
            for (int a = 0; a < R; a++)
                for (int k = 0; k < 5; k++) {
                    s = s * 6364136223846793005ull + 1442695040888963407ull;
                    src.push_back(h * R + a); dst.push_back(((h + 1) % T) * R + (int)((s >> 33) % R)); w.push_back(1.0 + (double)((s >> 20) % 40));
                }
        std::vector<jint> sources(R); for (int a = 0; a < R; a++) sources[(size_t)a] = a;
        jlong g2 = J(graphCreate)(e, cls, 0); NO_EXCEPTION();
        J(graphAddEdges)(e, cls, g2, ints(src), ints(dst), doubles(w), (jint)src.size()); NO_EXCEPTION();
        J(graphSetSources)(e, cls, g2, ints(sources), R, 0); NO_EXCEPTION();
        // walks before the alias tables: call order violated -> exception, and the out array comes back untouched
        const long n = 3000;
        jintArray out = ints(std::vector<jint>((size_t)n * L, -7));
        J(sampleWalks)(e, cls, g2, n, L, 7, 1, 0, out);
        CHECK(!g_pending.empty()); g_pending.clear();
        J(graphBuildAlias)(e, cls, g2, 0); NO_EXCEPTION();
        jlong draws = J(sampleWalks)(e, cls, g2, n, L, 7, 1, 0, out); NO_EXCEPTION();
        std::vector<int32_t> direct((size_t)n * L); int64_t ddraws = 0;
        CHECK(dge_sample_walks((const dge_graph*)g2, n, L, 7, 1, 0, direct.data(), &ddraws) == DGE_OK);
        CHECK(draws == ddraws && std::memcmp(direct.data(), fo(out)->i.data(), direct.size() * 4) == 0);              // copied back, identical
        CHECK(fo(out)->i[0] >= 0 && fo(out)->i[0] < R && fo(out)->i[1] >= R && fo(out)->i[1] < 2 * R);                // layer 0 -> layer 1
        // DeepWalk.learnEmbedding: J/DeepWalk.java:62-79 (one in-order worker here so that two runs can be compared bit for bit)
        jlong m = J(trainSgns)(e, cls, 0, out, n, L, D, T, 5, 2, 1, 1, 0.025f, 1e-4f, 42, NV, 0); NO_EXCEPTION();
        CHECK(m != 0);
        jintArray ids = ints(std::vector<jint>((size_t)NV, -1));
        jfloatArray vec = J(modelVectors)(e, cls, m, ids); NO_EXCEPTION();
        dge_train_config c{}; c.dim = D; c.window = T; c.negative = 5; c.min_count = 2; c.epochs = 1; c.workers = 1; c.alpha = 0.025f; c.min_alpha = 1e-4f; c.seed = 42; c.n_vertices = NV;
        dge_model* dm = nullptr;
        CHECK(dge_train_sgns(0, direct.data(), n, L, &c, &dm) == DGE_OK);
        const float* syn0; const int32_t* vid; int64_t V; int32_t dim;
        CHECK(dge_model_vectors(dm, &syn0, &vid, &V, &dim) == DGE_OK);
        CHECK(V > 100 && dim == D && (int64_t)fo(vec)->f.size() == V * D);
        CHECK(std::memcmp(fo(vec)->f.data(), syn0, (size_t)(V * D) * 4) == 0 && std::memcmp(fo(ids)->i.data(), vid, (size_t)V * 4) == 0);
        CHECK(fo(ids)->i[(size_t)V] == -1 || V == NV);
        bool moved = false; for (int64_t k = 0; k < V * D; k++) if (std::fabs(syn0[k]) > 0.5f / D) moved = true;      // trained, not InitNet's +-0.5/D
        CHECK(moved);
        // WordVectorSerializer.writeWordVectors (J/DeepWalk.java:82): names by vertex id
        FakeObj* names = new FakeObj(FakeObj::OBJ);
        for (int v = 0; v < NV; v++) names->o.push_back(str(std::to_string(v / R) + "-" + std::to_string(v % R)));
        const std::string path = tmp + "/jni_shim.vec";
        J(writeVec)(e, cls, m, names, str(path), 0); NO_EXCEPTION();
        std::ifstream f(path); std::string first; std::getline(f, first);
        const std::string name0 = std::to_string(vid[0] / R) + "-" + std::to_string(vid[0] % R);
        CHECK(first.compare(0, name0.size() + 1, name0 + " ") == 0);
        long lines = 1; std::string ln; while (std::getline(f, ln)) lines++;
        CHECK(lines == V);
        dge_model_free(dm);
        J(modelFree)(e, cls, m); J(graphFree)(e, cls, g2);
    }
    CHECK(g_outstanding == 0);                       // every Get...Elements / GetStringUTFChars met its Release
    std::printf("JNI SHIM OK (%ld element buffers copied back, %ld released with JNI_ABORT)\n", g_copied_back, g_aborted);
    return 0;
}
