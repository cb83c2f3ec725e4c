package embedding;

import java.util.ArrayList;
import java.util.HashMap;
import java.util.LinkedList;
import java.util.List;
import java.util.Map;
import java.util.Random;

/**
 * Drop-in for the reference's embedding.LayeredGraph: every public / protected member of J/LayeredGraph.java is here with
 * the same name, type and meaning, so that the reference's own callers (J/CrossTimeGraph.java, J/SpatialGraph.java,
 * T/LayeredGraphTest.java) compile against this class as written.  What changes is WHERE the work is done:
 *
 *   - the host keeps what the reference exposes as mutable public state (Vertex.edgesOut, Vertex.outDegree,
 *     allEdges, allVertices, sourceVertices, sourceWeightSum) — callers edit it freely, as SpatialGraph does;
 *   - initiateAliasTables() ships that state to the GPU in one bulk upload (NativeEngine -> libdge.so), builds every
 *     alias table there and reads probTable / aliasTable back into the Java fields;
 *   - sampleVertexSequence() hands out walks that the device sampled in batches from the SAME java.util.Random stream
 *     (one nextDouble per decision, J/LayeredGraph.java:108,234), and keeps LayeredGraph.rnd in step with them;
 *   - sampleVertexSequences(n) is the bulk form the writer loops use.
 *
 * Single-step members (Vertex.sampleNextVertex(), Vertex.sampleNextVertex(double)) evaluate the device-built tables on
 * the host: one table lookup per call is not worth a device round trip.
 *
 * Not carried over: the @deprecated O(V) samplers sampleNextVertex_OV / sampleVertexSequence_OV (J/LayeredGraph.java:84-98,
 * 254-281), which the reference keeps for timing comparisons only and nothing calls.
 *
 * NOT COMPILED IN THIS REPOSITORY'S CI (the build image has no JDK): see INTEGRATION.md.
 */
public class LayeredGraph {

    public static Random rnd = new Random();     // J/LayeredGraph.java:14 — assign `new Random(seed)` for seeded walks
    public static int numLayer = 8;              // J/LayeredGraph.java:15

    /** J/LayeredGraph.java:17-27 */
    static public class Edge {
        public Vertex from;
        public Vertex to;
        public double weight;

        public Edge(Vertex f, Vertex t, double w) {
            from = f;
            to = t;
            weight = w;
        }
    }

    /** J/LayeredGraph.java:29-133 */
    static public class Vertex {
        public String name;
        public int id;
        public List<Edge> edgesOut;
        public double outDegree;

        int[] aliasTable;
        double[] probTable;

        public Vertex(String n, int i) {
            name = n;
            id = i;
            edgesOut = new ArrayList<>();
            outDegree = 0;
        }

        /** J/LayeredGraph.java:46-49: outDegree is the running sum in insertion order */
        public void addOutEdge(Edge e) {
            edgesOut.add(e);
            outDegree += e.weight;
        }

        /**
         * J/LayeredGraph.java:54-82 for ONE vertex (what T/LayeredGraphTest.java calls): a one-vertex store goes to the
         * device, the table is built there in the reference's pairing order and read back.  Graph-wide construction
         * never comes through here — LayeredGraph.initiateAliasTables() builds all tables in one device pass.
         */
        public void initiateAliasTable() {
            int k = edgesOut.size();
            probTable = new double[k];
            aliasTable = new int[k];
            if (k == 0)
                return;
            int[] src = new int[k], dst = new int[k];
            double[] w = new double[k];
            for (int i = 0; i < k; i++) {
                src[i] = 0;
                dst[i] = i + 1;
                w[i] = edgesOut.get(i).weight;
            }
            long g = NativeEngine.graphCreate(Integer.getInteger("dge.device", 0));
            try {
                NativeEngine.graphAddEdges(g, src, dst, w, k);
                double[] od = new double[k + 1];
                od[0] = outDegree;                              // the field as it stands, not a recomputed sum (:62)
                NativeEngine.graphSetOutDegree(g, od, k + 1);
                NativeEngine.graphBuildAlias(g, true);
                NativeEngine.graphGetAlias(g, 0, probTable, aliasTable, dst);
            } finally {
                NativeEngine.graphFree(g);
            }
        }

        /** J/LayeredGraph.java:104-116: no draw is taken from a dead end */
        public Vertex sampleNextVertex() {
            if (edgesOut.size() == 0)
                return null;
            return sampleNextVertex(rnd.nextDouble());
        }

        /** J/LayeredGraph.java:123-132 [test purpose]; alias -1 ("slot is full") keeps the slot's own edge */
        public Vertex sampleNextVertex(double x) {
            int k = edgesOut.size();
            int i = (int) (x * k);
            double y = x * k - i;
            if (y < probTable[i] || aliasTable[i] < 0)
                return edgesOut.get(i).to;
            return edgesOut.get(aliasTable[i]).to;
        }
    }

    // ------------------------------------------------------------------ J/LayeredGraph.java:142-155
    public List<Edge> allEdges;
    public Map<String, Vertex> allVertices;

    public List<Vertex> sourceVertices;
    protected double sourceWeightSum;
    protected double[] probTable;
    protected int[] aliasTable;

    public LayeredGraph() {
        allEdges = new ArrayList<>();
        allVertices = new HashMap<>();
        sourceVertices = new ArrayList<>();
        sourceWeightSum = 0;
    }

    /** J/LayeredGraph.java:157-174: ids are insertion ordinals, duplicate (from, to) pairs are kept */
    public void addEdge(String fn, String tn, double weight) {
        Vertex f = allVertices.get(fn);
        if (f == null) {
            f = new Vertex(fn, allVertices.size());
            allVertices.put(fn, f);
        }
        Vertex t = allVertices.get(tn);
        if (t == null) {
            t = new Vertex(tn, allVertices.size());
            allVertices.put(tn, t);
        }
        Edge e = new Edge(f, t, weight);
        allEdges.add(e);
        f.addOutEdge(e);
        dropDeviceState();
    }

    /**
     * J/LayeredGraph.java:180-189 (call after all edges).  An unknown name yields a Vertex that is NOT registered in
     * allVertices, exactly as the reference does (:182-183); the reference then throws a NullPointerException when a walk
     * starts there (:245), here such a walk is the single token.
     */
    public void addSourceVertex(String vn) {
        Vertex v = allVertices.get(vn);
        if (v == null)
            v = new Vertex(vn, allVertices.size());
        sourceVertices.add(v);
        sourceWeightSum += v.outDegree;
        dropDeviceState();
    }

    /**
     * J/LayeredGraph.java:195-226.  The host state as it stands NOW — every vertex's edgesOut (in list order) and
     * outDegree field, sourceVertices (in list order) and sourceWeightSum — is uploaded in one piece; all alias tables
     * are built on the device in the reference's pairing order and read back into Vertex.probTable / aliasTable and
     * this.probTable / aliasTable.
     */
    public void initiateAliasTables() {
        initiateAliasTables(true);
    }

    /** exactReferenceOrder = false selects the O(k) Vose pairing: same sampling distribution, other alias indices */
    public void initiateAliasTables(boolean exactReferenceOrder) {
        upload();
        int[] s = new int[sourceVertices.size()];
        for (int i = 0; i < s.length; i++)
            s[i] = deviceId(sourceVertices.get(i));
        NativeEngine.graphSetSources(handle, s, s.length, false);
        NativeEngine.graphSetSourceWeightSum(handle, sourceWeightSum);
        NativeEngine.graphBuildAlias(handle, exactReferenceOrder);
        // read-back: the tables in CSR order, cut into the per-vertex arrays the reference keeps
        long[] rowPtr = new long[byId.length + 1];
        double[] prob = new double[(int) uploadedEdges];
        int[] alias = new int[(int) uploadedEdges];
        NativeEngine.graphGetCsr(handle, rowPtr, null, null, prob, alias, null);
        for (int v = 0; v < byId.length; v++) {
            Vertex x = byId[v];
            if (x == null)
                continue;
            int k = (int) (rowPtr[v + 1] - rowPtr[v]);
            x.probTable = new double[k];
            x.aliasTable = new int[k];
            System.arraycopy(prob, (int) rowPtr[v], x.probTable, 0, k);
            System.arraycopy(alias, (int) rowPtr[v], x.aliasTable, 0, k);
        }
        probTable = new double[s.length];
        aliasTable = new int[s.length];
        NativeEngine.graphGetSourceAlias(handle, probTable, aliasTable);
        aliasBuilt = true;
    }

    /**
     * J/LayeredGraph.java:232-252: one walk of at most numLayer names.  Walks are sampled on the device a batch at a time
     * from the stream position of LayeredGraph.rnd and handed out one by one; after every call rnd stands where the
     * reference's rnd would (one nextDouble per node of the walk), so code that draws from rnd between calls sees the
     * reference's numbers.  Draws taken from rnd by OTHER code between two calls are noticed at the next call (rnd's state is read —
     * nothing is consumed — and compared with where this class left it): the rest of the batch is dropped and re-sampled from rnd's
     * actual position, as it is when rnd is replaced and whenever numLayer or the graph changes.  The state check costs a few
     * microseconds per call; sampleVertexSequences(n) is the bulk form the writer loops should use.
     */
    public List<String> sampleVertexSequence() {
        if (!aliasBuilt)
            throw new IllegalStateException("call initiateAliasTables() first (J/LayeredGraph.java:195)");
        if (cachePos >= cacheRows || cacheL != numLayer || cacheRnd != rnd
                || JavaRandomState.peek(rnd) != JavaRandomState.jump(cacheState0, 2 * cacheDraws))     // someone else drew from rnd since our last call
            refill();
        LinkedList<String> seq = new LinkedList<>();
        int base = cachePos * cacheL;
        for (int j = 0; j < cacheL && cache[base + j] >= 0; j++)
            seq.add(nameOfDeviceId(cache[base + j]));
        cachePos++;
        cacheDraws += seq.size();                                // one draw per node: source pick + steps taken
        rnd.setSeed(JavaRandomState.jump(cacheState0, 2 * cacheDraws) ^ JavaRandomState.MULT);
        return seq;
    }

    /**
     * Bulk form for the writer loops (J/CrossTimeGraph.java:134-140, J/SpatialGraph.java:103-113): n walks of device
     * vertex ids, numLayer per row, -1 padded — the same walks n calls of sampleVertexSequence() would return, and rnd
     * advanced the same way.  Use nameOfDeviceId() to turn ids into names.
     */
    public int[] sampleVertexSequences(long n) {
        if (!aliasBuilt)
            throw new IllegalStateException("call initiateAliasTables() first (J/LayeredGraph.java:195)");
        long state = currentState();
        int[] out = new int[Math.toIntExact(n * numLayer)];
        long draws = NativeEngine.sampleWalks(handle, n, numLayer, state ^ JavaRandomState.MULT, 0, 0, out);
        rnd.setSeed(JavaRandomState.jump(state, 2 * draws) ^ JavaRandomState.MULT);
        cachePos = cacheRows = 0;
        return out;
    }

    /** name of a device vertex id as returned by sampleVertexSequences */
    public String nameOfDeviceId(int id) {
        return id < byId.length && byId[id] != null ? byId[id].name : extraNames.get(id - byId.length);
    }

    // ------------------------------------------------------------------ device side
    long handle;                                 // dge_graph*, 0 = nothing on the device
    private boolean aliasBuilt;
    private Vertex[] byId = new Vertex[0];       // registered vertices by id at upload time
    private final List<String> extraNames = new ArrayList<>();   // unregistered source vertices: device ids byId.length + i
    private final Map<Vertex, Integer> extraIds = new HashMap<>();
    private long uploadedEdges;
    // batch of walks sampled ahead for sampleVertexSequence()
    private int[] cache = new int[0];
    private int cacheRows, cachePos, cacheL;
    private long cacheState0, cacheDraws;
    private Random cacheRnd;

    /** any edit of the store invalidates what the device holds */
    protected void dropDeviceState() {
        aliasBuilt = false;
        cachePos = cacheRows = 0;
    }

    /** after keepNearestKVertices-style edits made directly on edgesOut / outDegree, callers need nothing: the next
     *  initiateAliasTables() uploads the lists as they are */
    protected void upload() {
        if (handle != 0) {
            NativeEngine.graphFree(handle);
            handle = 0;
        }
        int maxId = -1;
        for (Vertex v : allVertices.values())
            maxId = Math.max(maxId, v.id);
        byId = new Vertex[maxId + 1];
        long e = 0;
        for (Vertex v : allVertices.values()) {
            byId[v.id] = v;
            e += v.edgesOut.size();
        }
        extraNames.clear();
        extraIds.clear();
        for (Vertex v : sourceVertices)
            if ((v.id >= byId.length || byId[v.id] != v) && !extraIds.containsKey(v)) {
                extraIds.put(v, byId.length + extraNames.size());
                extraNames.add(v.name);
            }
        if (e > Integer.MAX_VALUE - 8)
            throw new IllegalStateException("more than 2^31 edges: build the store with NativeEngine.graphAddEdges in pieces");
        int[] src = new int[(int) e], dst = new int[(int) e];
        double[] w = new double[(int) e];
        int n = 0;
        for (Vertex v : byId) {
            if (v == null)
                continue;
            for (Edge ed : v.edgesOut) {
                src[n] = v.id;
                dst[n] = deviceId(ed.to);
                w[n] = ed.weight;
                n++;
            }
        }
        handle = NativeEngine.graphCreate(Integer.getInteger("dge.device", 0));
        int nv = byId.length + extraNames.size();
        NativeEngine.graphAddEdges(handle, src, dst, w, n);
        NativeEngine.graphReserveVertices(handle, nv);
        double[] od = new double[nv];
        for (Vertex v : byId)
            if (v != null)
                od[v.id] = v.outDegree;
        for (Map.Entry<Vertex, Integer> x : extraIds.entrySet())
            od[x.getValue()] = x.getKey().outDegree;
        NativeEngine.graphSetOutDegree(handle, od, nv);              // Vertex.outDegree is a public field: honoured as it stands
        uploadedEdges = n;
    }

    private int deviceId(Vertex v) {
        if (v.id < byId.length && byId[v.id] == v)
            return v.id;
        Integer x = extraIds.get(v);
        if (x == null)
            throw new IllegalStateException("vertex " + v.name + " is neither in allVertices nor a source vertex");
        return x;
    }

    /** where rnd stands: read from the object itself (two outputs, put back), so draws other code took from it count */
    private long currentState() {
        return JavaRandomState.peek(rnd);
    }

    private void refill() {
        long state = currentState();
        cacheL = numLayer;
        cacheRows = Integer.getInteger("dge.walkBatch", 65536);
        if (cache.length < cacheRows * cacheL)
            cache = new int[cacheRows * cacheL];
        NativeEngine.sampleWalks(handle, cacheRows, cacheL, state ^ JavaRandomState.MULT, 0, 0, cache);
        cachePos = 0;
        cacheState0 = state;
        cacheDraws = 0;
        cacheRnd = rnd;
    }

    /** release the device copy now instead of at garbage collection */
    public void close() {
        if (handle != 0) {
            NativeEngine.graphFree(handle);
            handle = 0;
        }
        dropDeviceState();
    }

    @Override
    protected void finalize() {
        close();
    }
}
