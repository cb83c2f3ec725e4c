// embedding_host.hpp — C++ host mirror of the reference's operator surface above the C ABI (include/dge.h).
//
// The reference is compiled Java (no JDK in this image), so the host side is C++ with the SAME class and member
// names, argument meaning and error behaviour as
//   J/LayeredGraph.java   (addEdge :157, addSourceVertex :180, initiateAliasTables :195, sampleVertexSequence :232,
//                          Vertex.sampleNextVertex(double) :123, public static rnd :14 / numLayer :15)
//   J/CrossTimeGraph.java (numSamples/numLayer :18-19, outputSampleSequence :115-124, sampleSequenceHelper :127-148)
//   J/SpatialGraph.java   (keepNearestKVertices :29-35, outputSampleSequence :91-121 with the "j-" prefix :105-108)
//   J/DeepWalk.java       (learnEmbedding :32-83: corpus -> Word2Vec(minWordFrequency 2, layerSize, window = numLayer,
//                          negativeSample 5) -> fit -> writeWordVectors)
// Name <-> id interning ("h-regionId") lives here, ids are insertion ordinals (J/LayeredGraph.java:160,166); all
// sampling and training runs in libdge.so on the GPU.  The Java/JNI form of the same surface is in java/.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dge.h"

namespace embedding {

inline void dge_check(int rc) {
    if (rc != DGE_OK) throw std::runtime_error(std::string("libdge: ") + dge_last_error());   // JNI shim: RuntimeException
}

// java.util.Random as the host sees it: a seed plus the number of nextDouble() draws already taken.  The device
// sampler continues the stream from that position (dge_sample_walks rng_mode 0).
class Random {
 public:
    Random() : seed_(0x5EEDC0DEBA5EULL ^ (int64_t)(uintptr_t)this), draws_(0) {}
    explicit Random(int64_t seed) : seed_(seed), draws_(0) {}
    int64_t seed() const { return seed_; }
    int64_t draws() const { return draws_; }
    void advance(int64_t n) { draws_ += n; }
    double nextDouble() {
        uint64_t s = jump(((uint64_t)seed_ ^ 0x5DEECE66DULL) & MASK, 2ULL * (uint64_t)draws_);
        s = (s * 0x5DEECE66DULL + 0xBULL) & MASK; int64_t hi = (int64_t)(s >> 22);
        s = (s * 0x5DEECE66DULL + 0xBULL) & MASK; int64_t lo = (int64_t)(s >> 21);
        draws_++;
        return (double)((hi << 27) + lo) * 0x1.0p-53;
    }

 private:
    static constexpr uint64_t MASK = (1ULL << 48) - 1;
    static uint64_t jump(uint64_t s, uint64_t n) {
        uint64_t am = 1, ap = 0, cm = 0x5DEECE66DULL, cp = 0xBULL;
        while (n) { if (n & 1) { am *= cm; ap = ap * cm + cp; } cp = (cm + 1) * cp; cm *= cm; n >>= 1; }
        return (am * s + ap) & MASK;
    }
    int64_t seed_, draws_;
};

class LayeredGraph {
 public:
    static Random rnd;       // public static Random rnd            J/LayeredGraph.java:14
    static int numLayer;     // public static int numLayer = 8      J/LayeredGraph.java:15

    struct Vertex {          // J/LayeredGraph.java:29-133 (read-back view; the tables live in HBM)
        std::string name;
        int id = -1;
        double outDegree = 0;
        std::vector<int> aliasTable;
        std::vector<double> probTable;
        std::vector<int> edgesOutTo;   // ids of edgesOut[i].to
        const LayeredGraph* g = nullptr;
        // sampleNextVertex(double x)  J/LayeredGraph.java:123-132 ; -1 = null
        int sampleNextVertex(double x) const { int32_t n; dge_check(dge_graph_sample_next(g->h_, id, x, &n)); return n; }
    };

    explicit LayeredGraph(int device = 0) : device_(device) { dge_check(dge_graph_create(&h_, device)); }
    virtual ~LayeredGraph() { dge_graph_free(h_); }
    LayeredGraph(const LayeredGraph&) = delete;

    std::unordered_map<std::string, int> allVertices;   // name -> id   (Map<String,Vertex> allVertices :143)
    std::vector<std::string> vertexNames;                // id -> name
    std::vector<int> sourceVertices;                     // (List<Vertex> sourceVertices :145)
    int64_t numEdges() const { return (int64_t)(src_.size()) + flushed_; }

    // addEdge(fn, tn, weight)  J/LayeredGraph.java:157-174
    void addEdge(const std::string& fn, const std::string& tn, double weight) {
        int f = intern(fn), t = intern(tn);
        src_.push_back(f); dst_.push_back(t); w_.push_back(weight);
        built_ = false;
        if (src_.size() >= (1u << 20)) flush();
    }
    // addSourceVertex(vn): call after all edges  J/LayeredGraph.java:180-189.  An unknown name is accepted by the
    // reference (a Vertex outside allVertices, :182-183) and then breaks the walk; here it is rejected.
    void addSourceVertex(const std::string& vn) {
        auto it = allVertices.find(vn);
        if (it == allVertices.end()) throw std::runtime_error("addSourceVertex: unknown vertex " + vn);
        sourceVertices.push_back(it->second);
        built_ = false;
    }
    // initiateAliasTables()  J/LayeredGraph.java:195-226.  exactReferenceOrder=false selects the O(k) Vose pairing.
    void initiateAliasTables(bool exactReferenceOrder = true, bool streamSumSources = false) {
        flush();
        dge_check(dge_graph_set_sources(h_, sourceVertices.data(), (int64_t)sourceVertices.size(), streamSumSources ? 1 : 0));
        dge_check(dge_graph_build_alias(h_, exactReferenceOrder ? 1 : 0));
        built_ = true;
        cache_.clear(); cache_pos_ = 0;
    }
    // sampleVertexSequence()  J/LayeredGraph.java:232-252: <= numLayer names, consumes LayeredGraph.rnd
    std::vector<std::string> sampleVertexSequence() {
        need_built();
        if (cache_pos_ >= cache_rows_ || cache_seed_ != rnd.seed() || cache_next_draw_ != rnd.draws() || cache_L_ != numLayer) refill();
        std::vector<std::string> seq;
        const int32_t* row = cache_.data() + (size_t)cache_pos_ * cache_L_;
        int n = 0;
        for (int j = 0; j < cache_L_ && row[j] >= 0; j++, n++) seq.push_back(vertexNames[row[j]]);
        cache_pos_++;
        rnd.advance(n);                                   // one draw per node of the walk
        cache_next_draw_ = rnd.draws();
        return seq;
    }
    // bulk form used by the writer loops: n walks, ids, -1 padded; consumes LayeredGraph.rnd like n single calls
    std::vector<int32_t> sampleVertexSequences(int64_t n) {
        need_built();
        std::vector<int32_t> out((size_t)n * numLayer);
        int64_t draws = 0;
        dge_check(dge_sample_walks(h_, n, numLayer, rnd.seed(), 0, rnd.draws(), out.data(), &draws));
        rnd.advance(draws);
        cache_pos_ = cache_rows_ = 0;
        return out;
    }
    Vertex vertex(const std::string& name) const {
        Vertex v; v.g = this; v.name = name; v.id = allVertices.at(name);
        int32_t k = 0;
        const_cast<LayeredGraph*>(this)->flush();
        dge_check(dge_graph_get_alias(h_, v.id, nullptr, nullptr, nullptr, nullptr, 0, &k, &v.outDegree));
        v.aliasTable.resize(k); v.probTable.resize(k); v.edgesOutTo.resize(k);
        if (k) dge_check(dge_graph_get_alias(h_, v.id, built_ ? v.probTable.data() : nullptr, built_ ? v.aliasTable.data() : nullptr,
                                             v.edgesOutTo.data(), nullptr, k, &k, &v.outDegree));
        return v;
    }
    dge_graph* handle() { flush(); return h_; }
    int device() const { return device_; }

 protected:
    int intern(const std::string& n) {
        auto it = allVertices.find(n);
        if (it != allVertices.end()) return it->second;
        int id = (int)vertexNames.size();                 // new Vertex(name, allVertices.size())  :160,166
        allVertices.emplace(n, id); vertexNames.push_back(n);
        return id;
    }
    void flush() {
        if (src_.empty()) return;
        dge_check(dge_graph_add_edges(h_, src_.data(), dst_.data(), w_.data(), (int64_t)src_.size()));
        flushed_ += (int64_t)src_.size();
        src_.clear(); dst_.clear(); w_.clear();
    }
    void need_built() const { if (!built_) throw std::runtime_error("call initiateAliasTables() first (J/LayeredGraph.java:195)"); }
    void refill() {
        cache_L_ = numLayer; cache_rows_ = 4096; cache_pos_ = 0;
        cache_.assign((size_t)cache_rows_ * cache_L_, -1);
        int64_t draws = 0;
        dge_check(dge_sample_walks(h_, cache_rows_, cache_L_, rnd.seed(), 0, rnd.draws(), cache_.data(), &draws));
        cache_seed_ = rnd.seed(); cache_next_draw_ = rnd.draws();
    }
    dge_graph* h_ = nullptr;
    int device_;
    std::vector<int32_t> src_, dst_; std::vector<double> w_;
    int64_t flushed_ = 0;
    bool built_ = false;
    std::vector<int32_t> cache_; int64_t cache_rows_ = 0, cache_pos_ = 0; int cache_L_ = 0;
    int64_t cache_seed_ = 0, cache_next_draw_ = -1;
};
inline Random LayeredGraph::rnd;
inline int LayeredGraph::numLayer = 8;

// One flow observation of the taxi data: trips from region src to region dst in time slice h.  The reference reads
// these from serialized flow maps (J/Tracts.java:474-482), which are out of scope; the mirror takes the tuples.
struct Flow { int slice; int src; int dst; double count; };

class CrossTimeGraph : public LayeredGraph {
 public:
    static int64_t numSamples;   // J/CrossTimeGraph.java:18
    static int numLayer;         // J/CrossTimeGraph.java:19
    using LayeredGraph::LayeredGraph;
    // constructGraph_tract / constructGraph_CA  J/CrossTimeGraph.java:25-52,68-95: edge "h-src" -> "(h+1)%T-dst" for every
    // positive flow, sources = layer-0 vertices that exist, in the order of `regions`
    static void constructGraph(CrossTimeGraph& g, const std::vector<Flow>& flows, const std::vector<int>& regions) {
        for (const Flow& f : flows)
            if (f.count > 0)
                g.addEdge(std::to_string(f.slice) + "-" + std::to_string(f.src),
                          std::to_string((f.slice + 1) % numLayer) + "-" + std::to_string(f.dst), f.count);
        for (int r : regions) {
            std::string n = "0-" + std::to_string(r);
            if (g.allVertices.count(n)) g.addSourceVertex(n);
        }
    }
    // the same graph from the reference's own per-slice edge files "taxi-h<h>.od", one "src dst w" line per flow
    // (J/Tracts.java:236-264, J/CommunityAreas.java:171-186; the CA writer also emits w == 0 lines, dropped here as
    // the graph builders drop them, J/CrossTimeGraph.java:37-38).  files[h] is slice h; sources in ascending region id.
    static void constructGraphFromOD(CrossTimeGraph& g, const std::vector<std::string>& files) {
        numLayer = (int)files.size();
        std::vector<Flow> flows;
        std::vector<int> regions;
        for (size_t h = 0; h < files.size(); h++) {
            std::ifstream in(files[h]);
            if (!in) throw std::runtime_error("cannot open " + files[h]);
            long long a, b; double w;
            while (in >> a >> b >> w) {
                flows.push_back({(int)h, (int)a, (int)b, w});
                if (h == 0) regions.push_back((int)a);
            }
        }
        std::sort(regions.begin(), regions.end());
        regions.erase(std::unique(regions.begin(), regions.end()), regions.end());
        constructGraph(g, flows, regions);
    }
    // outputSampleSequence + sampleSequenceHelper  J/CrossTimeGraph.java:115-148: numSamples lines of space-joined names
    static void outputSampleSequence(CrossTimeGraph& g, const std::string& path, bool exactReferenceOrder = true) {
        LayeredGraph::numLayer = CrossTimeGraph::numLayer;            // :116 (global side effect kept)
        g.initiateAliasTables(exactReferenceOrder);
        write_seq(g, path, numSamples, false);
    }

 protected:
    friend class SpatialGraph;
    static void write_seq(LayeredGraph& g, const std::string& path, int64_t n, bool positionPrefix) {
        std::ofstream out(path);
        if (!out) throw std::runtime_error("cannot open " + path);
        const int L = LayeredGraph::numLayer;
        const int64_t chunk = 1 << 18;
        for (int64_t done = 0; done < n; done += chunk) {
            int64_t m = std::min(chunk, n - done);
            std::vector<int32_t> w = g.sampleVertexSequences(m);
            std::string line;
            for (int64_t i = 0; i < m; i++) {
                line.clear();
                for (int j = 0; j < L && w[(size_t)i * L + j] >= 0; j++) {
                    if (j) line += ' ';
                    if (positionPrefix) { line += std::to_string(j); line += '-'; }   // J/SpatialGraph.java:105-108
                    line += g.vertexNames[w[(size_t)i * L + j]];
                }
                line += '\n';
                out << line;
            }
        }
    }
};
inline int64_t CrossTimeGraph::numSamples = 10000000;
inline int CrossTimeGraph::numLayer = 8;

class SpatialGraph : public LayeredGraph {
 public:
    static int64_t numSamples;   // J/SpatialGraph.java:16
    static int numLayer;         // J/SpatialGraph.java:17
    using LayeredGraph::LayeredGraph;
    // keepNearestKVertices(k)  J/SpatialGraph.java:29-35 (before sources / alias tables)
    void keepNearestKVertices(int k) { dge_check(dge_graph_keep_top_k(handle(), k)); }
    // constructGraph_*  J/SpatialGraph.java:37-88: complete graph with w = exp(-100 d) (self loop included), top-10,
    // every vertex a source (order of `names`), sourceWeightSum by DoubleStream.sum()
    static void constructGraph(SpatialGraph& g, const std::vector<std::string>& names, const std::vector<double>& weight /* n x n */) {
        size_t n = names.size();
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < n; j++) g.addEdge(names[i], names[j], weight[i * n + j]);
        g.keepNearestKVertices(10);
        g.sourceVertices.clear();
        for (const std::string& s : names) g.sourceVertices.push_back(g.allVertices.at(s));
        g.initiateAliasTables(true, /*streamSumSources=*/true);
    }
    static void outputSampleSequence(SpatialGraph& g, const std::string& path) {
        LayeredGraph::numLayer = SpatialGraph::numLayer;              // J/SpatialGraph.java:92
        CrossTimeGraph::write_seq(g, path, numSamples, true);
    }
};
inline int64_t SpatialGraph::numSamples = 5000000;
inline int SpatialGraph::numLayer = 8;

// J/DeepWalk.java:32-83
class DeepWalk {
 public:
    static int Year;   // J/DeepWalk.java:25
    // The reference's builder (:73-76) never calls .useHierarchicSoftmax(false), so DL4J trained the hierarchical-softmax
    // term next to the 5 negatives; true reproduces that, false is the negative-sampling path BASELINE.json names.
    static bool useHierarchicSoftmax;
    // learnEmbedding: every line of the .seq files is a sentence of whitespace-separated names (DefaultTokenizerFactory,
    // :70); trains SGNS with the reference's builder values and writes "name v1 .. vD" lines (:82).
    static dge_train_stats learnEmbedding(const std::vector<std::string>& seqFiles, const std::string& outVec, int layerSize,
                                          int device = 0, int workers = 0, uint64_t seed = 1) {
        std::unordered_map<std::string, int> ids;
        std::vector<std::string> names;
        std::vector<std::vector<int32_t>> rows;
        size_t maxLen = 1;
        for (const std::string& f : seqFiles) {
            std::ifstream in(f);
            if (!in) throw std::runtime_error("cannot open " + f);
            std::string line, tok;
            while (std::getline(in, line)) {
                std::istringstream ss(line);
                std::vector<int32_t> r;
                while (ss >> tok) {
                    auto it = ids.find(tok);
                    if (it == ids.end()) { it = ids.emplace(tok, (int)names.size()).first; names.push_back(tok); }
                    r.push_back(it->second);
                }
                if (!r.empty()) { maxLen = std::max(maxLen, r.size()); rows.push_back(std::move(r)); }
            }
        }
        std::vector<int32_t> walks(rows.size() * maxLen, -1);
        for (size_t i = 0; i < rows.size(); i++) std::copy(rows[i].begin(), rows[i].end(), walks.begin() + i * maxLen);
        dge_train_config cfg{};
        cfg.dim = layerSize;                       // .layerSize(layerSize)
        cfg.window = LayeredGraph::numLayer;       // .windowSize(LayeredGraph.numLayer)   :74 (the global, as in the reference)
        cfg.negative = 5;                          // .negativeSample(5)
        cfg.min_count = 2;                         // .minWordFrequency(2)
        cfg.epochs = 1;                            // .iterations(1), epochs default 1
        cfg.workers = workers;                     // .workers(8) -> GPU workers
        cfg.alpha = 0.025f; cfg.min_alpha = 1e-4f; // DL4J defaults
        cfg.seed = seed; cfg.table_size = 0;
        cfg.n_vertices = (int32_t)std::max<size_t>(names.size(), 1);
        cfg.use_hs = useHierarchicSoftmax ? 1 : 0;
        dge_model* m = nullptr;
        dge_check(dge_train_sgns(device, walks.data(), (int64_t)rows.size(), (int32_t)maxLen, &cfg, &m));
        std::vector<const char*> cn(names.size());
        for (size_t i = 0; i < names.size(); i++) cn[i] = names[i].c_str();
        dge_check(dge_write_vec(m, cn.data(), outVec.c_str(), 0));
        dge_train_stats st{};
        dge_check(dge_model_stats(m, &st));
        dge_model_free(m);
        return st;
    }
};
inline int DeepWalk::Year = 2013;
inline bool DeepWalk::useHierarchicSoftmax = false;

}  // namespace embedding
