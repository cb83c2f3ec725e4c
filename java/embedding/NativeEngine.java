package embedding;

/**
 * JNI view of libdge.so (include/dge.h).  One static native per C entry point the Java surface needs; handles are
 * passed as long.  Loaded once; every non-zero status becomes a RuntimeException carrying dge_last_error()
 * (the reference swallows exceptions at J/DeepWalk.java:137-139 — callers keep doing that).
 * NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no JDK (see INTEGRATION.md).
 */
final class NativeEngine {
    static { System.loadLibrary("dge_jni"); }
    private NativeEngine() {}

    static native long graphCreate(int device);                                          // dge_graph_create
    static native void graphFree(long g);                                                // dge_graph_free
    static native void graphAddEdges(long g, int[] src, int[] dst, double[] w, int n);   // dge_graph_add_edges
    /** vertex ids [0, n) exist even when no edge names them (dge_graph_reserve_vertices) */
    static native void graphReserveVertices(long g, int n);
    /** Vertex.outDegree as the host holds it, for ids [0, n) (dge_graph_set_out_degree) */
    static native void graphSetOutDegree(long g, double[] outDegree, int n);
    static native void graphSetSources(long g, int[] v, int n, boolean streamSum);       // dge_graph_set_sources
    /** LayeredGraph.sourceWeightSum as the host holds it (dge_graph_set_source_weight_sum) */
    static native void graphSetSourceWeightSum(long g, double sum);
    static native void graphKeepTopK(long g, int k);                                     // dge_graph_keep_top_k
    static native void graphBuildAlias(long g, boolean exactReferenceOrder);             // dge_graph_build_alias
    /** fills prob/alias/nbr (length >= degree) and returns {degree, outDegree} packed as double[2] (dge_graph_get_alias) */
    static native double[] graphGetAlias(long g, int v, double[] prob, int[] alias, int[] nbr);
    /** the whole store in CSR order; any array may be null (dge_graph_get_csr) */
    static native void graphGetCsr(long g, long[] rowPtr, int[] nbr, double[] w, double[] prob, int[] alias, double[] outDegree);
    static native void graphGetSourceAlias(long g, double[] prob, int[] alias);          // dge_graph_get_source_alias
    static native int graphSampleNext(long g, int v, double x);                          // dge_graph_sample_next
    /** rngMode 0: the java.util.Random(seed) stream continued at firstIndex draws; returns the draws this call consumed */
    static native long sampleWalks(long g, long nWalks, int maxLen, long seed, int rngMode, long firstIndex, int[] out);
    /** w2v.fit(): returns a model handle (dge_train_sgns) */
    static native long trainSgns(int device, int[] walks, long nWalks, int maxLen, int dim, int window, int negative,
                                 int minCount, int epochs, int workers, float alpha, float minAlpha, long seed, int nVertices,
                                 boolean useHierarchicSoftmax);
    static native void writeVec(long model, String[] names, String path, boolean header); // dge_write_vec
    static native float[] modelVectors(long model, int[] vocabIdsOut);                   // dge_model_vectors
    static native void modelFree(long m);                                                // dge_model_free
}
