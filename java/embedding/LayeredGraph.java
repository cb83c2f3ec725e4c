package embedding;

import java.lang.reflect.Field;
import java.util.*;
import java.util.concurrent.atomic.AtomicLong;

/**
 * Drop-in for the reference's embedding.LayeredGraph (same public members), with the edge store, alias tables and
 * walk sampler in HBM behind libdge.so.  Names are interned here; ids are insertion ordinals exactly as the
 * reference assigns them (J/LayeredGraph.java:160,166).
 */
public class LayeredGraph {
    public static Random rnd = new Random();     // J/LayeredGraph.java:14 — assign `new Random(seed)` for seeded walks
    public static int numLayer = 8;              // J/LayeredGraph.java:15

    public static class Vertex {                 // read-back view of J/LayeredGraph.java:29-133
        public String name; public int id; public double outDegree;
        int[] aliasTable; double[] probTable; int[] edgesOutTo;
        final LayeredGraph g;
        Vertex(LayeredGraph g, String n, int i) { this.g = g; name = n; id = i; }
        /** sampleNextVertex(double x) — test overload, J/LayeredGraph.java:123-132 */
        public Vertex sampleNextVertex(double x) {
            int t = NativeEngine.graphSampleNext(g.handle(), id, x);
            return t < 0 ? null : g.vertexById(t);
        }
    }

    public Map<String, Vertex> allVertices = new HashMap<>();
    public List<Vertex> sourceVertices = new ArrayList<>();
    protected double sourceWeightSum;            // kept for source compatibility; the device owns the value
    private final List<Vertex> byId = new ArrayList<>();
    private long h = NativeEngine.graphCreate(Integer.getInteger("dge.device", 0));
    private int[] bs = new int[1 << 16], bd = new int[1 << 16]; private double[] bw = new double[1 << 16]; private int nb = 0;
    private long seedOfRnd; private long drawsOfRnd; private Random boundRnd;

    long handle() { flush(); return h; }
    Vertex vertexById(int id) { return byId.get(id); }

    public void addEdge(String fn, String tn, double weight) {            // J/LayeredGraph.java:157-174
        Vertex f = intern(fn), t = intern(tn);
        if (nb == bs.length) flush();
        bs[nb] = f.id; bd[nb] = t.id; bw[nb] = weight; nb++;
    }
    public void addSourceVertex(String vn) {                               // J/LayeredGraph.java:180-189
        Vertex v = allVertices.get(vn);
        if (v == null) throw new IllegalArgumentException("unknown vertex " + vn);
        sourceVertices.add(v);
    }
    public void initiateAliasTables() { initiateAliasTables(true, false); } // J/LayeredGraph.java:195-226
    public void initiateAliasTables(boolean exactReferenceOrder, boolean streamSum) {
        flush();
        int[] s = new int[sourceVertices.size()];
        for (int i = 0; i < s.length; i++) s[i] = sourceVertices.get(i).id;
        NativeEngine.graphSetSources(h, s, s.length, streamSum);
        NativeEngine.graphBuildAlias(h, exactReferenceOrder);
    }
    /** J/LayeredGraph.java:232-252.  The stream position of `rnd` is tracked through its seed (read once by
     *  reflection when a new Random object is assigned) and the number of draws the device has consumed. */
    public List<String> sampleVertexSequence() {
        int[] row = sampleVertexSequences(1);
        List<String> seq = new LinkedList<>();
        for (int j = 0; j < numLayer && row[j] >= 0; j++) seq.add(byId.get(row[j]).name);
        return seq;
    }
    /** bulk form for the writer loops (J/CrossTimeGraph.java:134-140): n walks of ids, -1 padded */
    public int[] sampleVertexSequences(long n) {
        bindRnd();
        int[] out = new int[(int) (n * numLayer)];
        drawsOfRnd += NativeEngine.sampleWalks(handle(), n, numLayer, seedOfRnd, 0, drawsOfRnd, out);
        return out;
    }
    private void bindRnd() {
        if (boundRnd == rnd) return;
        try {                                   // java.util.Random keeps (seed ^ 0x5DEECE66D) & mask in `seed`
            Field f = Random.class.getDeclaredField("seed"); f.setAccessible(true);
            seedOfRnd = ((AtomicLong) f.get(rnd)).get() ^ 0x5DEECE66DL;
        } catch (ReflectiveOperationException e) { throw new IllegalStateException(e); }
        drawsOfRnd = 0; boundRnd = rnd;
    }
    private Vertex intern(String n) {
        Vertex v = allVertices.get(n);
        if (v == null) { v = new Vertex(this, n, allVertices.size()); allVertices.put(n, v); byId.add(v); }
        return v;
    }
    private void flush() { if (nb > 0) { NativeEngine.graphAddEdges(h, bs, bd, bw, nb); nb = 0; } }
    @Override protected void finalize() { if (h != 0) { NativeEngine.graphFree(h); h = 0; } }
}
