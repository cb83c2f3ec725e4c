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

    static native long graphCreate(int device);
    static native void graphFree(long g);
    static native void graphAddEdges(long g, int[] src, int[] dst, double[] w, int n);
    static native void graphSetSources(long g, int[] v, int n, boolean streamSum);
    static native void graphKeepTopK(long g, int k);
    static native void graphBuildAlias(long g, boolean exactReferenceOrder);
    /** fills prob/alias/nbr (length >= degree) and returns {degree, outDegree} packed as double[2] */
    static native double[] graphGetAlias(long g, int v, double[] prob, int[] alias, int[] nbr);
    static native int graphSampleNext(long g, int v, double x);
    /** rngMode 0: java.util.Random(seed) stream continued at drawsConsumed; returns draws consumed by this call */
    static native long sampleWalks(long g, long nWalks, int maxLen, long seed, int rngMode, long firstIndex, int[] out);
    /** w2v.fit(): returns a model handle */
    static native long trainSgns(int device, int[] walks, long nWalks, int maxLen, int dim, int window, int negative,
                                 int minCount, int epochs, int workers, float alpha, float minAlpha, long seed, int nVertices,
                                 boolean useHierarchicSoftmax);
    static native void writeVec(long model, String[] names, String path, boolean header);
    static native float[] modelVectors(long model, int[] vocabIdsOut);
    static native void modelFree(long m);
}
