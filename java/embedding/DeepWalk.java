package embedding;

import java.io.BufferedReader;
import java.io.File;
import java.io.IOException;
import java.nio.file.Files;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

/**
 * Drop-in for the reference's embedding.DeepWalk (J/DeepWalk.java): same public members.  The DL4J calls of
 * learnEmbedding — new Word2Vec.Builder()...build() (:73-76), w2v.fit() (:79), WordVectorSerializer.writeWordVectors (:82)
 * — become one native training call with the same hyper-parameters and one native writer call; the DL4J / ND4J / slf4j
 * dependencies go away.
 *
 * DL4J 0.7.2's builder leaves the hierarchical-softmax term ON unless useHierarchicSoftmax(false) is called, and the
 * reference never calls it: the authors' runs trained HS + 5 negatives.  That is the default here too
 * (-Ddge.hs=false trains negative sampling alone — the path BASELINE.json's metric is quoted on).
 *
 * NOT COMPILED IN THIS REPOSITORY'S CI (no JDK in the build image): see INTEGRATION.md.
 */
public class DeepWalk {
    public static int Year = 2013;                            // J/DeepWalk.java:25

    public static void learnEmbedding() throws Exception {   // J/DeepWalk.java:28-30
        learnEmbedding("tract", "usespatial");
    }

    /** J/DeepWalk.java:32-83 */
    public static void learnEmbedding(String regionLevel, String spatialGF) throws Exception {
        String dir = String.format("../miscs/%d/deepwalkseq-%s", Year, regionLevel);
        List<File> files = new ArrayList<>();
        if (spatialGF.equals("usespatial"))
            collect(new File(dir), files);                    // FileSentenceIterator(seqDir): every file below the directory (:49-50)
        else if (spatialGF.equals("nospatial"))
            files.add(new File(dir, "taxi-crosstime.seq"));  // LineSentenceIterator (:52-53)
        else if (spatialGF.equals("onlyspatial"))
            files.add(new File(dir, "taxi-spatial.seq"));    // (:55-56)
        String out = String.format("../miscs/%d/taxi-deepwalk-%s-%s-2D.vec", Year, regionLevel, spatialGF);
        int layerSize = regionLevel.equals("CA") ? 2 : 20;   // :62-66

        // sentences -> rows of token ids (DefaultTokenizerFactory: whitespace tokens, :70), -1 padded to the longest
        Map<String, Integer> ids = new HashMap<>();
        List<String> names = new ArrayList<>();
        List<int[]> rows = new ArrayList<>();
        int maxLen = 1;
        for (File f : files)
            try (BufferedReader in = Files.newBufferedReader(f.toPath())) {
                for (String line = in.readLine(); line != null; line = in.readLine()) {
                    String[] tok = line.trim().split("\\s+");
                    if (tok.length == 0 || tok[0].isEmpty())
                        continue;
                    int[] r = new int[tok.length];
                    for (int i = 0; i < tok.length; i++) {
                        Integer id = ids.get(tok[i]);
                        if (id == null) {
                            id = names.size();
                            ids.put(tok[i], id);
                            names.add(tok[i]);
                        }
                        r[i] = id;
                    }
                    rows.add(r);
                    maxLen = Math.max(maxLen, r.length);
                }
            }
        int[] walks = new int[Math.multiplyExact(rows.size(), maxLen)];
        Arrays.fill(walks, -1);
        for (int i = 0; i < rows.size(); i++)
            System.arraycopy(rows.get(i), 0, walks, i * maxLen, rows.get(i).length);

        boolean hs = Boolean.parseBoolean(System.getProperty("dge.hs", "true"));
        long m = NativeEngine.trainSgns(Integer.getInteger("dge.device", 0), walks, rows.size(), maxLen,
                layerSize,                 // .layerSize(layerSize)                    :74
                LayeredGraph.numLayer,     // .windowSize(LayeredGraph.numLayer)       :74
                5,                         // .negativeSample(5)                       :75
                2,                         // .minWordFrequency(2)                     :73
                1,                         // .iterations(1), epochs default 1         :74
                0,                         // .workers(8) -> 0: fill the GPU           :75
                0.025f, 1e-4f,             // DL4J defaults learningRate / minLearningRate
                Long.getLong("dge.seed", 1L), names.size(), hs);
        try {
            NativeEngine.writeVec(m, names.toArray(new String[0]), out, false);   // writeWordVectors(w2v, out)  :82
        } finally {
            NativeEngine.modelFree(m);
        }
    }

    private static void collect(File f, List<File> out) {
        File[] kids = f.listFiles();
        if (kids == null) {
            if (f.isFile())
                out.add(f);
            return;
        }
        Arrays.sort(kids);
        for (File k : kids)
            collect(k, out);
    }

    /** J/DeepWalk.java:85-113: sample the walk files that learnEmbedding needs and that are not there yet */
    public static void checkInputFile(String regionLevel, String spatialGF) {
        boolean tract = regionLevel.equals("tract");
        String dir = String.format("../miscs/%d/deepwalkseq-%s/", Year, regionLevel);
        boolean wantSpatial = spatialGF.equals("usespatial") || spatialGF.equals("onlyspatial");
        boolean wantCrossTime = spatialGF.equals("usespatial") || spatialGF.equals("nospatial");
        if (wantSpatial && !new File(dir + "taxi-spatial.seq").exists()) {
            System.out.format("The spatial graph samples for %s do not exist, but we need it! Generating ...\n", regionLevel);
            SpatialGraph.numSamples = tract ? 600_000 : 80_000;
            SpatialGraph.numLayer = tract ? 8 : 24;
            SpatialGraph.outputSampleSequence(regionLevel);
        }
        if (wantCrossTime && !new File(dir + "taxi-crosstime.seq").exists()) {
            System.out.format("The transition graph samples for %s do not exists! Generating ...\n", regionLevel);
            CrossTimeGraph.numSamples = tract ? 15_000_000 : 8_000_000;
            CrossTimeGraph.numLayer = tract ? 8 : 24;
            CrossTimeGraph.outputSampleSequence(regionLevel);
        }
    }

    /**
     * J/DeepWalk.java:120-140: argv[0] "tract" | "CA", argv[1] "usespatial" | "nospatial" | "onlyspatial", argv[2] year.
     * Exceptions are printed and swallowed, as in the reference (:137-139).
     */
    public static void main(String[] argv) {
        try {
            String regionLevel = argv.length > 0 ? argv[0] : "tract";
            String spatialGF = argv.length > 1 ? argv[1] : "usespatial";
            if (argv.length > 0)
                System.out.format("word2vec learn embedding at %s level.\n", regionLevel);
            if (argv.length > 1)
                System.out.format("Spatial graph use or not: %s.\n", spatialGF);
            if (argv.length > 2)
                DeepWalk.Year = Integer.parseInt(argv[2]);
            checkInputFile(regionLevel, spatialGF);
            learnEmbedding(regionLevel, spatialGF);
        } catch (Exception e) {
            e.printStackTrace();
        }
    }
}
