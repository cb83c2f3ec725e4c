package embedding;

import java.io.BufferedWriter;
import java.io.FileWriter;
import java.io.IOException;
import java.io.Writer;

/**
 * Drop-in for the reference's embedding.CrossTimeGraph (J/CrossTimeGraph.java): same public members.  Edge rule (:36-39,
 * :79-82): flow count w > 0 of slice h from region a to region b is the edge "h-a" -> "((h+1) % numLayer)-b" of weight w;
 * sources are the layer-0 vertices that exist, in the region map's order (:43-47, :85-89).  The store, the alias tables
 * and the sampler live on the GPU (LayeredGraph -> NativeEngine -> libdge.so).
 * Tracts / Tract / CommunityAreas / CommunityArea are the reference's own data classes (out of scope, unchanged).
 *
 * NOT COMPILED IN THIS REPOSITORY'S CI (no JDK in the build image): see INTEGRATION.md.
 */
public class CrossTimeGraph extends LayeredGraph {

    public static int numSamples = 10_000_000;              // J/CrossTimeGraph.java:18
    public static int numLayer = LayeredGraph.numLayer;      // J/CrossTimeGraph.java:19

    public CrossTimeGraph() {
        super();
    }

    private static String node(int layer, int region) {
        return layer + "-" + region;
    }

    public static CrossTimeGraph constructGraph_tract() {
        Tracts trts = new Tracts();
        trts.deserialzeTracts(DeepWalk.Year);
        long t1 = System.currentTimeMillis();
        System.out.println("Start generating cross-time graph...");
        int timeStep = 24 / numLayer;
        CrossTimeGraph g = new CrossTimeGraph();
        for (int h = 0; h < numLayer; h++)
            for (Tract a : trts.tracts.values())
                for (Tract b : trts.tracts.values()) {
                    int w = a.getFlowTo(b.id, h, h + timeStep - 1);
                    if (w > 0)
                        g.addEdge(node(h, a.id), node((h + 1) % numLayer, b.id), w);
                }
        for (Tract a : trts.tracts.values())
            if (g.allVertices.containsKey(node(0, a.id)))
                g.addSourceVertex(node(0, a.id));
        System.out.format("Cross-time graph built successfully in %d milliseconds.\n", System.currentTimeMillis() - t1);
        return g;
    }

    public static CrossTimeGraph constructGraph_CA() {
        int timeStep = 24 / numLayer;                         // uniform time slots by default (:55-59)
        int[] timeIntervals = new int[numLayer + 1];
        for (int i = 0; i <= numLayer; i += timeStep)
            timeIntervals[i] = (i * timeStep) % numLayer;
        return constructGraph_CA(timeIntervals);
    }

    /** timeIntervals[h] (inclusive) .. timeIntervals[h+1] (exclusive) is slice h (:68-95) */
    public static CrossTimeGraph constructGraph_CA(int[] timeIntervals) {
        CommunityAreas cas = new CommunityAreas();
        cas.deserialzeCAs(DeepWalk.Year);
        CrossTimeGraph.numLayer = timeIntervals.length - 1;
        long t1 = System.currentTimeMillis();
        System.out.println("Start generating crosstime graph for Communities ...");
        CrossTimeGraph g = new CrossTimeGraph();
        for (int h = 0; h < numLayer; h++)
            for (CommunityArea a : cas.communities.values())
                for (CommunityArea b : cas.communities.values()) {
                    int w = a.getFlowTo(b.id, timeIntervals[h], timeIntervals[h + 1]);
                    if (w > 0)
                        g.addEdge(node(h, a.id), node((h + 1) % numLayer, b.id), w);
                }
        for (CommunityArea a : cas.communities.values())
            if (g.allVertices.containsKey(node(0, a.id)))
                g.addSourceVertex(node(0, a.id));
        System.out.format("Crosstime graph for communities built successfully in %d milliseconds.\n", System.currentTimeMillis() - t1);
        return g;
    }

    /** J/CrossTimeGraph.java:103-112 */
    public static void outputSampleSequence(String regionLevel, int[] timeIntervals) {
        LayeredGraph.numLayer = CrossTimeGraph.numLayer;
        CrossTimeGraph g = regionLevel.equals("tract") ? constructGraph_tract() : constructGraph_CA(timeIntervals);
        g.initiateAliasTables();
        sampleSequenceHelper(g, regionLevel);
    }

    /** J/CrossTimeGraph.java:115-124 */
    public static void outputSampleSequence(String regionLevel) {
        LayeredGraph.numLayer = CrossTimeGraph.numLayer;
        CrossTimeGraph g = regionLevel.equals("tract") ? constructGraph_tract() : constructGraph_CA();
        g.initiateAliasTables();
        sampleSequenceHelper(g, regionLevel);
    }

    /**
     * J/CrossTimeGraph.java:127-148: numSamples lines of space-joined names into
     * ../miscs/&lt;Year&gt;/deepwalkseq-&lt;level&gt;/taxi-crosstime.seq — the walks of the reference's loop, sampled on the device in blocks.
     */
    public static void sampleSequenceHelper(CrossTimeGraph g, String regionLevel) {
        long t2 = System.currentTimeMillis();
        System.out.println("Starting sequence sampling...");
        String path = String.format("../miscs/%d/deepwalkseq-%s/taxi-crosstime.seq", DeepWalk.Year, regionLevel);
        try (BufferedWriter fout = new BufferedWriter(new FileWriter(path))) {
            writeWalks(g, fout, numSamples, false);
        } catch (IOException e) {
            e.printStackTrace();
        }
        System.out.format("Sampling %d sequences finished in %f seconds.\n", numSamples, (System.currentTimeMillis() - t2) / 1000.0);
    }

    /** n walks as text lines; positionPrefix writes token j as "j-name" (J/SpatialGraph.java:105-108) */
    static void writeWalks(LayeredGraph g, Writer out, long n, boolean positionPrefix) throws IOException {
        final int L = LayeredGraph.numLayer;
        final long block = 1 << 18;
        StringBuilder line = new StringBuilder(16 * L);
        long tenth = Math.max(n / 10, 1);
        for (long done = 0; done < n; done += block) {
            int m = (int) Math.min(block, n - done);
            int[] rows = g.sampleVertexSequences(m);
            for (int i = 0; i < m; i++) {
                line.setLength(0);
                for (int j = 0; j < L && rows[i * L + j] >= 0; j++) {
                    if (j > 0)
                        line.append(' ');
                    if (positionPrefix)
                        line.append(j).append('-');
                    line.append(g.nameOfDeviceId(rows[i * L + j]));
                }
                out.write(line.append('\n').toString());
            }
            if ((done + m) / tenth != done / tenth)
                System.out.format("%d%% finished\n", Math.min(100, (done + m) * 100 / n));
        }
    }

    public static void main(String[] argv) {
        int[] numSamplesSet = new int[]{500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000};
        for (int ns : numSamplesSet) {
            numSamples = ns;
            numLayer = 24;
            outputSampleSequence("CA");
        }
    }
}
