package embedding;

import java.io.BufferedWriter;
import java.io.FileWriter;
import java.io.IOException;
import java.util.ArrayList;
import java.util.LinkedList;
import java.util.List;

/**
 * Drop-in for the reference's embedding.SpatialGraph (J/SpatialGraph.java): same public members; the segmented top-k prune,
 * the alias tables and the walk sampling run on the GPU behind libdge.so.
 * Tracts / Tract / CommunityAreas / CommunityArea are the reference's own geometry classes (out of scope here: they stay
 * as they are in the reference tree this file is dropped into).
 *
 * NOT COMPILED IN THIS REPOSITORY'S CI (no JDK in the build image): see INTEGRATION.md.
 */
public class SpatialGraph extends LayeredGraph {

    public static int numSamples = 5_000_000;               // J/SpatialGraph.java:16
    public static int numLayer = LayeredGraph.numLayer;      // J/SpatialGraph.java:17

    public SpatialGraph() {
        super();
    }

    /**
     * J/SpatialGraph.java:29-35 — per vertex: stable sort of edgesOut by weight descending, keep the first k, outDegree =
     * DoubleStream.sum() of what is left.  Done for all vertices at once by the device's segmented sort
     * (dge_graph_keep_top_k); edgesOut and outDegree of every vertex are then rebuilt from the device's result, so the
     * public fields show the pruned store as they do in the reference.  A vertex with fewer than k edges makes the
     * reference's subList(0, k) throw IndexOutOfBoundsException; so does this.
     */
    public void keepNearestKVertices(int k) {
        upload();
        try {
            NativeEngine.graphKeepTopK(handle, k);
        } catch (RuntimeException e) {
            throw new IndexOutOfBoundsException(e.getMessage());
        }
        int nv = 0;
        for (Vertex v : allVertices.values())
            nv = Math.max(nv, v.id + 1);
        Vertex[] byId = new Vertex[nv];
        for (Vertex v : allVertices.values())
            byId[v.id] = v;
        long[] rowPtr = new long[nv + 1];
        int[] nbr = new int[nv * k];
        double[] w = new double[nv * k], od = new double[nv];
        NativeEngine.graphGetCsr(handle, rowPtr, nbr, w, null, null, od);
        for (int v = 0; v < nv; v++) {
            if (byId[v] == null)
                continue;
            List<Edge> kept = new ArrayList<>(k);
            for (long e = rowPtr[v]; e < rowPtr[v + 1]; e++)
                kept.add(new Edge(byId[v], byId[nbr[(int) e]], w[(int) e]));
            byId[v].edgesOut = kept;
            byId[v].outDegree = od[v];
        }
        dropDeviceState();
    }

    /** the dense distance kernel of J/SpatialGraph.java:37-88: w = exp(-100 d), self loop included (d = 0 -> w = 1) */
    private static SpatialGraph fromDistances(int[] ids, double[][] dist) {
        SpatialGraph g = new SpatialGraph();
        for (int i = 0; i < ids.length; i++)
            for (int j = 0; j < ids.length; j++)
                g.addEdge(Integer.toString(ids[i]), Integer.toString(ids[j]), Math.exp(-dist[i][j] * 100));
        g.keepNearestKVertices(10);                          // issue (#4) of the reference: the 10 nearest only
        g.sourceVertices = new LinkedList<>(g.allVertices.values());
        g.sourceWeightSum = g.sourceVertices.stream().mapToDouble(x -> x.outDegree).sum();
        g.initiateAliasTables();
        return g;
    }

    public static SpatialGraph constructGraph_tract() {
        Tracts trts = new Tracts();
        long t1 = System.currentTimeMillis();
        System.out.println("Start generating spatial graph ...");
        List<Tract> all = new ArrayList<>(trts.tracts.values());
        int[] ids = new int[all.size()];
        double[][] d = new double[all.size()][all.size()];
        for (int i = 0; i < ids.length; i++) {
            ids[i] = all.get(i).id;
            for (int j = 0; j < ids.length; j++)
                d[i][j] = all.get(i).distanceTo(all.get(j));
        }
        SpatialGraph g = fromDistances(ids, d);
        System.out.format("Spatial graph built successfully in %d milliseconds.\n", System.currentTimeMillis() - t1);
        return g;
    }

    public static SpatialGraph constructGraph_CA() {
        CommunityAreas cas = new CommunityAreas();
        long t1 = System.currentTimeMillis();
        System.out.println("Start generating spatial graph for communities ... ");
        List<CommunityArea> all = new ArrayList<>(cas.communities.values());
        int[] ids = new int[all.size()];
        double[][] d = new double[all.size()][all.size()];
        for (int i = 0; i < ids.length; i++) {
            ids[i] = all.get(i).id;
            for (int j = 0; j < ids.length; j++)
                d[i][j] = all.get(i).distanceTo(all.get(j));
        }
        SpatialGraph g = fromDistances(ids, d);
        System.out.format("Spatial graph for community built successfully in %d milliseconds.\n", System.currentTimeMillis() - t1);
        return g;
    }

    /**
     * J/SpatialGraph.java:91-121: numSamples walks into ../miscs/&lt;Year&gt;/deepwalkseq-&lt;level&gt;/taxi-spatial.seq, token j of a walk
     * written as "j-name" (:105-108) so that spatial walks land in the cross-time vocabulary.  Walks come from the device in
     * blocks (sampleVertexSequences); the lines are those the reference's loop over sampleVertexSequence() writes.
     */
    public static void outputSampleSequence(String regionLevel) {
        LayeredGraph.numLayer = SpatialGraph.numLayer;
        SpatialGraph g = regionLevel.equals("tract") ? constructGraph_tract() : constructGraph_CA();
        long t2 = System.currentTimeMillis();
        System.out.println("Starting sequence sampling...");
        String path = String.format("../miscs/%d/deepwalkseq-%s/taxi-spatial.seq", DeepWalk.Year, regionLevel);
        try (BufferedWriter fout = new BufferedWriter(new FileWriter(path))) {
            CrossTimeGraph.writeWalks(g, fout, numSamples, true);
        } catch (IOException e) {
            e.printStackTrace();
        }
        System.out.format("Sampling %d sequences finished in %d seconds.\n", numSamples, (System.currentTimeMillis() - t2) / 1000);
    }

    public static void main(String[] argv) {
        numLayer = 24;
        numSamples = 80_000;                                 // number of nodes * 1000
        outputSampleSequence("CA");
    }
}
