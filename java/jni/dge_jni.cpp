// dge_jni.cpp — JNI shim between java/embedding/NativeEngine.java and libdge.so (include/dge.h).
// Optional target: needs a JDK (jni.h); the build image has none, so this file is shipped as source only.
//   g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include dge_jni.cpp -L../../embedding_amd -l:libdge.so -o libdge_jni.so
// Executed without a JVM by tests/native/jni_shim_test.cpp (a JNIEnv whose arrays are heap objects handed out as copies): marshalling, handles and the
// status -> RuntimeException mapping run against the real library on the GPU box (tests/test_gpu_host_mirror.py).
#include <jni.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "dge.h"

static bool fail(JNIEnv* e, int rc) {
    if (rc == DGE_OK) return false;
    e->ThrowNew(e->FindClass("java/lang/RuntimeException"), dge_last_error());
    return true;
}
#define J(name) Java_embedding_NativeEngine_##name

extern "C" {
JNIEXPORT jlong JNICALL J(graphCreate)(JNIEnv* e, jclass, jint device) { dge_graph* g = nullptr; fail(e, dge_graph_create(&g, device)); return (jlong)g; }
JNIEXPORT void JNICALL J(graphFree)(JNIEnv*, jclass, jlong g) { dge_graph_free((dge_graph*)g); }
JNIEXPORT void JNICALL J(graphAddEdges)(JNIEnv* e, jclass, jlong g, jintArray s, jintArray d, jdoubleArray w, jint n) {
    jint* ps = e->GetIntArrayElements(s, nullptr); jint* pd = e->GetIntArrayElements(d, nullptr); jdouble* pw = e->GetDoubleArrayElements(w, nullptr);
    int rc = dge_graph_add_edges((dge_graph*)g, (const int32_t*)ps, (const int32_t*)pd, pw, n);
    e->ReleaseIntArrayElements(s, ps, JNI_ABORT); e->ReleaseIntArrayElements(d, pd, JNI_ABORT); e->ReleaseDoubleArrayElements(w, pw, JNI_ABORT);
    fail(e, rc);
}
JNIEXPORT void JNICALL J(graphSetSources)(JNIEnv* e, jclass, jlong g, jintArray v, jint n, jboolean ss) {
    jint* p = e->GetIntArrayElements(v, nullptr);
    int rc = dge_graph_set_sources((dge_graph*)g, (const int32_t*)p, n, ss ? 1 : 0);
    e->ReleaseIntArrayElements(v, p, JNI_ABORT); fail(e, rc);
}
JNIEXPORT void JNICALL J(graphReserveVertices)(JNIEnv* e, jclass, jlong g, jint n) { fail(e, dge_graph_reserve_vertices((dge_graph*)g, n)); }
JNIEXPORT void JNICALL J(graphSetOutDegree)(JNIEnv* e, jclass, jlong g, jdoubleArray od, jint n) {
    jdouble* p = e->GetDoubleArrayElements(od, nullptr);
    int rc = dge_graph_set_out_degree((dge_graph*)g, p, n);
    e->ReleaseDoubleArrayElements(od, p, JNI_ABORT); fail(e, rc);
}
JNIEXPORT void JNICALL J(graphSetSourceWeightSum)(JNIEnv* e, jclass, jlong g, jdouble s) { fail(e, dge_graph_set_source_weight_sum((dge_graph*)g, s)); }
JNIEXPORT void JNICALL J(graphKeepTopK)(JNIEnv* e, jclass, jlong g, jint k) { fail(e, dge_graph_keep_top_k((dge_graph*)g, k)); }
JNIEXPORT void JNICALL J(graphBuildAlias)(JNIEnv* e, jclass, jlong g, jboolean exact) { fail(e, dge_graph_build_alias((dge_graph*)g, exact ? 1 : 0)); }
JNIEXPORT jdoubleArray JNICALL J(graphGetAlias)(JNIEnv* e, jclass, jlong g, jint v, jdoubleArray prob, jintArray alias, jintArray nbr) {
    jint cap = e->GetArrayLength(prob); int32_t k = 0; double od = 0;
    jdouble* pp = e->GetDoubleArrayElements(prob, nullptr); jint* pa = e->GetIntArrayElements(alias, nullptr); jint* pn = e->GetIntArrayElements(nbr, nullptr);
    int rc = dge_graph_get_alias((dge_graph*)g, v, pp, (int32_t*)pa, (int32_t*)pn, nullptr, cap, &k, &od);
    e->ReleaseDoubleArrayElements(prob, pp, 0); e->ReleaseIntArrayElements(alias, pa, 0); e->ReleaseIntArrayElements(nbr, pn, 0);
    if (fail(e, rc)) return nullptr;
    jdouble r[2] = {(jdouble)k, od}; jdoubleArray out = e->NewDoubleArray(2); e->SetDoubleArrayRegion(out, 0, 2, r); return out;
}
// any array may be null; lengths are taken from the arrays themselves
JNIEXPORT void JNICALL J(graphGetCsr)(JNIEnv* e, jclass, jlong g, jlongArray rowPtr, jintArray nbr, jdoubleArray w, jdoubleArray prob, jintArray alias,
                                      jdoubleArray outDegree) {
    jlong* prp = rowPtr ? e->GetLongArrayElements(rowPtr, nullptr) : nullptr;
    jint* pn = nbr ? e->GetIntArrayElements(nbr, nullptr) : nullptr;
    jdouble* pw = w ? e->GetDoubleArrayElements(w, nullptr) : nullptr;
    jdouble* pp = prob ? e->GetDoubleArrayElements(prob, nullptr) : nullptr;
    jint* pa = alias ? e->GetIntArrayElements(alias, nullptr) : nullptr;
    jdouble* po = outDegree ? e->GetDoubleArrayElements(outDegree, nullptr) : nullptr;
    jsize capV = 0x7fffffff; int64_t capE = INT64_MAX;
    if (rowPtr) capV = e->GetArrayLength(rowPtr) - 1;
    if (outDegree) capV = std::min<jsize>(capV, e->GetArrayLength(outDegree));
    if (nbr) capE = std::min<int64_t>(capE, e->GetArrayLength(nbr));
    if (w) capE = std::min<int64_t>(capE, e->GetArrayLength(w));
    if (prob) capE = std::min<int64_t>(capE, e->GetArrayLength(prob));
    if (alias) capE = std::min<int64_t>(capE, e->GetArrayLength(alias));
    static_assert(sizeof(jlong) == sizeof(int64_t), "jlong is 64 bits");
    int rc = dge_graph_get_csr((const dge_graph*)g, (int64_t*)prp, (int32_t*)pn, pw, pp, (int32_t*)pa, po, capV, capE);
    if (rowPtr) e->ReleaseLongArrayElements(rowPtr, prp, 0);
    if (nbr) e->ReleaseIntArrayElements(nbr, pn, 0);
    if (w) e->ReleaseDoubleArrayElements(w, pw, 0);
    if (prob) e->ReleaseDoubleArrayElements(prob, pp, 0);
    if (alias) e->ReleaseIntArrayElements(alias, pa, 0);
    if (outDegree) e->ReleaseDoubleArrayElements(outDegree, po, 0);
    fail(e, rc);
}
JNIEXPORT void JNICALL J(graphGetSourceAlias)(JNIEnv* e, jclass, jlong g, jdoubleArray prob, jintArray alias) {
    jint cap = e->GetArrayLength(prob); int32_t k = 0;
    jdouble* pp = e->GetDoubleArrayElements(prob, nullptr); jint* pa = e->GetIntArrayElements(alias, nullptr);
    int rc = dge_graph_get_source_alias((const dge_graph*)g, pp, (int32_t*)pa, nullptr, cap, &k, nullptr);
    e->ReleaseDoubleArrayElements(prob, pp, 0); e->ReleaseIntArrayElements(alias, pa, 0);
    fail(e, rc);
}
JNIEXPORT jint JNICALL J(graphSampleNext)(JNIEnv* e, jclass, jlong g, jint v, jdouble x) { int32_t n = -1; fail(e, dge_graph_sample_next((dge_graph*)g, v, x, &n)); return n; }
JNIEXPORT jlong JNICALL J(sampleWalks)(JNIEnv* e, jclass, jlong g, jlong n, jint L, jlong seed, jint mode, jlong first, jintArray out) {
    jint* p = e->GetIntArrayElements(out, nullptr); int64_t draws = 0;
    int rc = dge_sample_walks((dge_graph*)g, n, L, seed, mode, first, (int32_t*)p, &draws);
    e->ReleaseIntArrayElements(out, p, 0); fail(e, rc); return draws;
}
JNIEXPORT jlong JNICALL J(trainSgns)(JNIEnv* e, jclass, jint device, jintArray walks, jlong n, jint L, jint dim, jint window, jint negative,
                                     jint minCount, jint epochs, jint workers, jfloat alpha, jfloat minAlpha, jlong seed, jint nVertices,
                                     jboolean useHierarchicSoftmax) {
    dge_train_config c{}; c.dim = dim; c.window = window; c.negative = negative; c.min_count = minCount; c.epochs = epochs; c.workers = workers;
    c.alpha = alpha; c.min_alpha = minAlpha; c.seed = (uint64_t)seed; c.n_vertices = nVertices; c.use_hs = useHierarchicSoftmax ? 1 : 0;
    jint* p = e->GetIntArrayElements(walks, nullptr); dge_model* m = nullptr;
    int rc = dge_train_sgns(device, (const int32_t*)p, n, L, &c, &m);
    e->ReleaseIntArrayElements(walks, p, JNI_ABORT); fail(e, rc); return (jlong)m;
}
JNIEXPORT void JNICALL J(writeVec)(JNIEnv* e, jclass, jlong m, jobjectArray names, jstring path, jboolean header) {
    jsize n = e->GetArrayLength(names); std::vector<const char*> c(n); std::vector<jstring> js(n);
    for (jsize i = 0; i < n; i++) { js[i] = (jstring)e->GetObjectArrayElement(names, i); c[i] = e->GetStringUTFChars(js[i], nullptr); }
    const char* p = e->GetStringUTFChars(path, nullptr);
    int rc = dge_write_vec((dge_model*)m, c.data(), p, header ? 1 : 0);
    e->ReleaseStringUTFChars(path, p);
    for (jsize i = 0; i < n; i++) e->ReleaseStringUTFChars(js[i], c[i]);
    fail(e, rc);
}
JNIEXPORT jfloatArray JNICALL J(modelVectors)(JNIEnv* e, jclass, jlong m, jintArray idsOut) {
    const float* syn0; const int32_t* ids; int64_t V; int32_t D;
    if (fail(e, dge_model_vectors((dge_model*)m, &syn0, &ids, &V, &D))) return nullptr;
    jfloatArray out = e->NewFloatArray((jsize)(V * D)); e->SetFloatArrayRegion(out, 0, (jsize)(V * D), syn0);
    e->SetIntArrayRegion(idsOut, 0, (jsize)V, (const jint*)ids); return out;
}
JNIEXPORT void JNICALL J(modelFree)(JNIEnv*, jclass, jlong m) { dge_model_free((dge_model*)m); }
}
