package embedding;

import java.io.*;
import java.nio.file.*;
import java.util.*;

/**
 * Drop-in for the reference's embedding.DeepWalk.learnEmbedding (J/DeepWalk.java:32-83): the DL4J
 * Word2Vec.Builder()...fit()/writeWordVectors calls (:73-82) are replaced by one native call with the same
 * hyper-parameters.  checkInputFile/main are unchanged from the reference and therefore not repeated here.
 */
public class DeepWalk {
    public static int Year = 2013;

    public static void learnEmbedding(String regionLevel, String spatialGF) throws Exception {
        List<Path> files = new ArrayList<>();
        String dir = String.format("../miscs/%d/deepwalkseq-%s", Year, regionLevel);
        if (spatialGF.equals("usespatial")) try (DirectoryStream<Path> ds = Files.newDirectoryStream(Paths.get(dir))) { for (Path p : ds) files.add(p); }
        else if (spatialGF.equals("nospatial")) files.add(Paths.get(dir, "taxi-crosstime.seq"));
        else if (spatialGF.equals("onlyspatial")) files.add(Paths.get(dir, "taxi-spatial.seq"));
        String out = String.format("../miscs/%d/taxi-deepwalk-%s-%s-2D.vec", Year, regionLevel, spatialGF);
        int layerSize = regionLevel.equals("CA") ? 2 : 20;                       // J/DeepWalk.java:62-66

        Map<String, Integer> ids = new HashMap<>(); List<String> names = new ArrayList<>(); List<int[]> rows = new ArrayList<>();
        int maxLen = 1;
        for (Path f : files) try (BufferedReader in = Files.newBufferedReader(f)) {
            for (String line; (line = in.readLine()) != null; ) {
                String[] tok = line.trim().split("\\s+");                       // DefaultTokenizerFactory, :70
                if (tok.length == 0 || tok[0].isEmpty()) continue;
                int[] r = new int[tok.length];
                for (int i = 0; i < tok.length; i++) { Integer id = ids.get(tok[i]); if (id == null) { id = names.size(); ids.put(tok[i], id); names.add(tok[i]); } r[i] = id; }
                rows.add(r); maxLen = Math.max(maxLen, r.length);
            }
        }
        int[] walks = new int[rows.size() * maxLen];
        Arrays.fill(walks, -1);
        for (int i = 0; i < rows.size(); i++) System.arraycopy(rows.get(i), 0, walks, i * maxLen, rows.get(i).length);
        long m = NativeEngine.trainSgns(Integer.getInteger("dge.device", 0), walks, rows.size(), maxLen,
                layerSize, LayeredGraph.numLayer /* .windowSize(LayeredGraph.numLayer) :74 */, 5 /* .negativeSample(5) */,
                2 /* .minWordFrequency(2) */, 1 /* .iterations(1) */, 0 /* .workers(8) -> fill the GPU */,
                0.025f, 1e-4f, 1L, names.size(),
                Boolean.getBoolean("dge.hs") /* DL4J's builder default leaves the hierarchical-softmax term on (:73-76 never
                                                 call useHierarchicSoftmax); -Ddge.hs=true trains it as well */);
        NativeEngine.writeVec(m, names.toArray(new String[0]), out, false);      // WordVectorSerializer.writeWordVectors :82
        NativeEngine.modelFree(m);
    }
}
