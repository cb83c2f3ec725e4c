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

    /** flow count of one slice between two regions: the one thing the two region levels differ in */
    private interface SliceFlow {
        int count(int slice, int fromRegion, int toRegion);
    }

    /**
     * The edge rule once, for both region levels (J/CrossTimeGraph.java:36-47 and :79-89 apply it to Tracts and to CommunityAreas):
     * every positive flow of slice h becomes one edge from layer h to layer (h+1) % numLayer; the sources are the layer-0 vertices an
     * edge has created, in the order of `regions` (the region map's iteration order).  The edges go through addEdge because the
     * reference exposes allEdges / allVertices as public state; the device receives them in ONE bulk call (LayeredGraph.upload).
     */
    private static CrossTimeGraph fromFlows(int[] regions, SliceFlow flow, String what) {
        final long begun = System.currentTimeMillis();
        System.out.println("Start generating " + what + "...");
        final CrossTimeGraph g = new CrossTimeGraph();
        for (int h = 0; h < numLayer; h++) {
            final int next = (h + 1) % numLayer;
            for (int from : regions)
                for (int to : regions) {
                    final int w = flow.count(h, from, to);
                    if (w > 0)
                        g.addEdge(node(h, from), node(next, to), w);
                }
        }
        for (int r : regions)
            if (g.allVertices.containsKey(node(0, r)))
                g.addSourceVertex(node(0, r));
        System.out.format("%s built successfully in %d milliseconds.\n", what, System.currentTimeMillis() - begun);
        return g;
    }

    private static int[] idsOf(java.util.Collection<Integer> keys) {
        int[] ids = new int[keys.size()];
        int n = 0;
        for (int k : keys)
            ids[n++] = k;
        return ids;
    }

    /** J/CrossTimeGraph.java:25-52: tract level, numLayer slices of 24 / numLayer hours */
    public static CrossTimeGraph constructGraph_tract() {
        final Tracts trts = new Tracts();
        trts.deserialzeTracts(DeepWalk.Year);
        final int hours = 24 / numLayer;
        return fromFlows(idsOf(trts.tracts.keySet()),
                (h, a, b) -> trts.tracts.get(a).getFlowTo(b, h, h + hours - 1),
                "cross-time graph");
    }

    /** J/CrossTimeGraph.java:54-66: community-area level with the default, uniform time slots */
    public static CrossTimeGraph constructGraph_CA() {
        final int hours = 24 / numLayer;
        int[] bounds = new int[numLayer + 1];
        for (int i = 0; i <= numLayer; i += hours)
            bounds[i] = (i * hours) % numLayer;
        return constructGraph_CA(bounds);
    }

    /** J/CrossTimeGraph.java:68-95: slice h covers timeIntervals[h] (inclusive) .. timeIntervals[h+1] (exclusive) */
    public static CrossTimeGraph constructGraph_CA(int[] timeIntervals) {
        final CommunityAreas cas = new CommunityAreas();
        cas.deserialzeCAs(DeepWalk.Year);
        CrossTimeGraph.numLayer = timeIntervals.length - 1;
        return fromFlows(idsOf(cas.communities.keySet()),
                (h, a, b) -> cas.communities.get(a).getFlowTo(b, timeIntervals[h], timeIntervals[h + 1]),
                "crosstime graph for communities");
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
