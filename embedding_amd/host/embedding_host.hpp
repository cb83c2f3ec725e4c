// embedding_host.hpp — C++ host mirror of the reference's operator surface above the C ABI (include/dge.h).
//
// The reference is compiled Java (no JDK in this image), so the host side is C++ with the SAME class and member
// names, argument meaning and error behaviour as
//   J/LayeredGraph.java   (Edge :17-27; Vertex :29-133 with name/id/edgesOut/outDegree, addOutEdge :46, initiateAliasTable :54,
//                          sampleNextVertex() :104, sampleNextVertex(double) :123; allEdges/allVertices/sourceVertices :142-145,
//                          addEdge :157, addSourceVertex :180, initiateAliasTables :195, sampleVertexSequence :232,
//                          public static rnd :14 / numLayer :15)
//   J/CrossTimeGraph.java (numSamples/numLayer :18-19, outputSampleSequence :115-124, sampleSequenceHelper :127-148)
//   J/SpatialGraph.java   (keepNearestKVertices :29-35, outputSampleSequence :91-121 with the "j-" prefix :105-108)
//   J/DeepWalk.java       (learnEmbedding :32-83: corpus -> Word2Vec(minWordFrequency 2, layerSize, window = numLayer,
//                          negativeSample 5) -> fit -> writeWordVectors)
// As in the reference, the host owns the PUBLIC MUTABLE state (edgesOut, outDegree, sourceVertices, sourceWeightSum: callers such
// as SpatialGraph edit it directly); initiateAliasTables() uploads that state as it stands in one piece, every alias table is built
// on the GPU and read back into the Vertex fields, and walks are sampled on the GPU from LayeredGraph::rnd's stream.  The
// Java/JNI form of the same surface is in java/ (same design, member for member).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <deque>
#include <fstream>
#include <set>
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
    static int device;       // GPU the stores live on (-Ddge.device of the Java form)

    struct Vertex;
    struct Edge {            // J/LayeredGraph.java:17-27
        Vertex* from; Vertex* to; double weight;
        Edge(Vertex* f, Vertex* t, double w) : from(f), to(t), weight(w) {}
    };
    struct Vertex {          // J/LayeredGraph.java:29-133
        std::string name;
        int id;
        std::vector<Edge> edgesOut;
        double outDegree = 0;
        std::vector<int> aliasTable;
        std::vector<double> probTable;

        Vertex(const std::string& n, int i) : name(n), id(i) {}
        void addOutEdge(const Edge& e) { edgesOut.push_back(e); outDegree += e.weight; }          // :46-49
        // :54-82 for ONE vertex (what T/LayeredGraphTest.java calls): a one-vertex store on the device, table read back
        void initiateAliasTable() {
            const int32_t k = (int32_t)edgesOut.size();
            probTable.assign(k, 0.0); aliasTable.assign(k, -1);
            if (k == 0) return;
            std::vector<int32_t> src(k, 0), dst(k); std::vector<double> w(k), od(k + 1, 0.0);
            for (int32_t i = 0; i < k; i++) { dst[i] = i + 1; w[i] = edgesOut[i].weight; }
            od[0] = outDegree;                                       // the field as it stands (:62)
            dge_graph* g = nullptr;
            dge_check(dge_graph_create(&g, LayeredGraph::device));
            int rc = dge_graph_add_edges(g, src.data(), dst.data(), w.data(), k);
            if (!rc) rc = dge_graph_set_out_degree(g, od.data(), k + 1);
            if (!rc) rc = dge_graph_build_alias(g, 1);
            int32_t kk = 0;
            if (!rc) rc = dge_graph_get_alias(g, 0, probTable.data(), aliasTable.data(), nullptr, nullptr, k, &kk, nullptr);
            dge_graph_free(g);
            dge_check(rc);
        }
        // :104-116: no draw is taken from a dead end; null = nullptr
        Vertex* sampleNextVertex() { return edgesOut.empty() ? nullptr : sampleNextVertex(LayeredGraph::rnd.nextDouble()); }
        // :123-132 [test purpose]: the device-built table evaluated on the host; alias -1 keeps the slot's own edge
        Vertex* sampleNextVertex(double x) {
            const int k = (int)edgesOut.size();
            const int i = (int)(x * k);
            const double y = x * k - i;
            return (y < probTable[i] || aliasTable[i] < 0) ? edgesOut[i].to : edgesOut[aliasTable[i]].to;
        }
    };

    std::vector<Edge> allEdges;                                  // :142
    std::unordered_map<std::string, Vertex*> allVertices;        // :143
    std::vector<Vertex*> sourceVertices;                         // :145
    double sourceWeightSum = 0;                                  // :146 (protected in Java; SpatialGraph assigns it)
    std::vector<double> probTable;                               // :147
    std::vector<int> aliasTable;                                 // :148

    LayeredGraph() = default;
    virtual ~LayeredGraph() { if (h_) dge_graph_free(h_); }
    LayeredGraph(const LayeredGraph&) = delete;

    // addEdge(fn, tn, weight)  :157-174: ids are insertion ordinals, duplicates are kept
    void addEdge(const std::string& fn, const std::string& tn, double weight) {
        Vertex* f = intern(fn); Vertex* t = intern(tn);
        Edge e(f, t, weight);
        allEdges.push_back(e);
        f->addOutEdge(e);
        built_ = false;
    }
    // addSourceVertex(vn): after all edges  :180-189.  An unknown name yields a Vertex that is NOT registered in allVertices, as in
    // the reference (:182-183); the reference then fails with a NullPointerException when a walk starts there, here that walk is
    // the single token.
    void addSourceVertex(const std::string& vn) {
        auto it = allVertices.find(vn);
        Vertex* v = it != allVertices.end() ? it->second : &store_.emplace_back(vn, (int)allVertices.size());
        sourceVertices.push_back(v);
        sourceWeightSum += v->outDegree;
        built_ = false;
    }
    // initiateAliasTables()  :195-226.  exactReferenceOrder=false selects the O(k) Vose pairing.
    void initiateAliasTables(bool exactReferenceOrder = true) {
        upload();
        std::vector<int32_t> s(sourceVertices.size());
        for (size_t i = 0; i < s.size(); i++) s[i] = deviceId(sourceVertices[i]);
        dge_check(dge_graph_set_sources(h_, s.data(), (int64_t)s.size(), 0));
        dge_check(dge_graph_set_source_weight_sum(h_, sourceWeightSum));
        dge_check(dge_graph_build_alias(h_, exactReferenceOrder ? 1 : 0));
        const int32_t V = (int32_t)byId_.size();
        std::vector<int64_t> rp((size_t)nv_ + 1); std::vector<double> prob((size_t)std::max<int64_t>(ne_, 1)); std::vector<int32_t> alias(prob.size());
        dge_check(dge_graph_get_csr(h_, rp.data(), nullptr, nullptr, prob.data(), alias.data(), nullptr, nv_, (int64_t)prob.size()));
        for (int32_t v = 0; v < V; v++) {
            if (!byId_[v]) continue;
            byId_[v]->probTable.assign(prob.begin() + rp[v], prob.begin() + rp[v + 1]);
            byId_[v]->aliasTable.assign(alias.begin() + rp[v], alias.begin() + rp[v + 1]);
        }
        probTable.assign(s.size(), 0.0); aliasTable.assign(s.size(), -1);
        int32_t k = 0;
        if (!s.empty()) dge_check(dge_graph_get_source_alias(h_, probTable.data(), aliasTable.data(), nullptr, (int32_t)s.size(), &k, nullptr));
        built_ = true;
        cache_rows_ = cache_pos_ = 0;
    }
    // sampleVertexSequence()  :232-252: <= numLayer names, consumes LayeredGraph::rnd (one draw per node of the walk)
    std::vector<std::string> sampleVertexSequence() {
        need_built();
        if (cache_pos_ >= cache_rows_ || cache_seed_ != rnd.seed() || cache_next_draw_ != rnd.draws() || cache_L_ != numLayer) refill();
        std::vector<std::string> seq;
        const int32_t* row = cache_.data() + (size_t)cache_pos_ * cache_L_;
        for (int j = 0; j < cache_L_ && row[j] >= 0; j++) seq.push_back(nameOfDeviceId(row[j]));
        cache_pos_++;
        rnd.advance((int64_t)seq.size());
        cache_next_draw_ = rnd.draws();
        return seq;
    }
    // bulk form used by the writer loops: n walks of device ids, -1 padded; consumes LayeredGraph::rnd like n single calls
    std::vector<int32_t> sampleVertexSequences(int64_t n) {
        need_built();
        std::vector<int32_t> out((size_t)n * numLayer);
        int64_t draws = 0;
        dge_check(dge_sample_walks(h_, n, numLayer, rnd.seed(), 0, rnd.draws(), out.data(), &draws));
        rnd.advance(draws);
        cache_pos_ = cache_rows_ = 0;
        return out;
    }
    const std::string& nameOfDeviceId(int32_t id) const { return id < (int32_t)byId_.size() && byId_[id] ? byId_[id]->name : extra_[id - (int32_t)byId_.size()]->name; }
    dge_graph* handle() { return h_; }
    int64_t numEdges() const { int64_t e = 0; for (const auto& kv : allVertices) e += (int64_t)kv.second->edgesOut.size(); return e; }

 protected:
    Vertex* intern(const std::string& n) {
        auto it = allVertices.find(n);
        if (it != allVertices.end()) return it->second;
        Vertex* v = &store_.emplace_back(n, (int)allVertices.size());     // new Vertex(name, allVertices.size())  :160,166
        allVertices.emplace(n, v);
        return v;
    }
    // the host state as it stands -> one device store (edges of a vertex in list order, outDegree as the field holds it)
    void upload() {
        if (h_) { dge_graph_free(h_); h_ = nullptr; }
        int maxId = -1;
        for (const auto& kv : allVertices) maxId = std::max(maxId, kv.second->id);
        byId_.assign((size_t)(maxId + 1), nullptr);
        int64_t E = 0;
        for (const auto& kv : allVertices) { byId_[kv.second->id] = kv.second; E += (int64_t)kv.second->edgesOut.size(); }
        extra_.clear();
        for (Vertex* v : sourceVertices)
            if ((v->id >= (int)byId_.size() || byId_[v->id] != v) && std::find(extra_.begin(), extra_.end(), v) == extra_.end()) extra_.push_back(v);
        std::vector<int32_t> src, dst; std::vector<double> w;
        src.reserve((size_t)E); dst.reserve((size_t)E); w.reserve((size_t)E);
        for (Vertex* v : byId_) {
            if (!v) continue;
            for (const Edge& e : v->edgesOut) { src.push_back(v->id); dst.push_back(deviceId(e.to)); w.push_back(e.weight); }
        }
        nv_ = (int32_t)(byId_.size() + extra_.size()); ne_ = E;
        dge_check(dge_graph_create(&h_, device));
        dge_check(dge_graph_add_edges(h_, src.data(), dst.data(), w.data(), E));
        dge_check(dge_graph_reserve_vertices(h_, nv_));
        std::vector<double> od((size_t)nv_, 0.0);
        for (Vertex* v : byId_) if (v) od[v->id] = v->outDegree;
        for (size_t i = 0; i < extra_.size(); i++) od[byId_.size() + i] = extra_[i]->outDegree;
        if (nv_) dge_check(dge_graph_set_out_degree(h_, od.data(), nv_));
    }
    int32_t deviceId(const Vertex* v) const {
        if (v->id < (int)byId_.size() && byId_[v->id] == v) return v->id;
        auto it = std::find(extra_.begin(), extra_.end(), v);
        if (it == extra_.end()) throw std::runtime_error("vertex " + v->name + " is neither in allVertices nor a source vertex");
        return (int32_t)(byId_.size() + (it - extra_.begin()));
    }
    void need_built() const { if (!built_) throw std::runtime_error("call initiateAliasTables() first (J/LayeredGraph.java:195)"); }
    void refill() {
        cache_L_ = numLayer; cache_rows_ = 4096; cache_pos_ = 0;
        cache_.assign((size_t)cache_rows_ * cache_L_, -1);
        int64_t draws = 0;
        dge_check(dge_sample_walks(h_, cache_rows_, cache_L_, rnd.seed(), 0, rnd.draws(), cache_.data(), &draws));
        cache_seed_ = rnd.seed(); cache_next_draw_ = rnd.draws();
    }
    std::deque<Vertex> store_;                 // owns every Vertex (stable addresses)
    dge_graph* h_ = nullptr;
    bool built_ = false;
    std::vector<Vertex*> byId_; std::vector<Vertex*> extra_;
    int32_t nv_ = 0; int64_t ne_ = 0;
    std::vector<int32_t> cache_; int64_t cache_rows_ = 0, cache_pos_ = 0; int cache_L_ = 0;
    int64_t cache_seed_ = 0, cache_next_draw_ = -1;
};
inline Random LayeredGraph::rnd;
inline int LayeredGraph::numLayer = 8;
inline int LayeredGraph::device = 0;

// One flow observation of the taxi data: trips from region src to region dst in time slice h.  The reference reads
// these from serialized flow maps (J/Tracts.java:474-482), which are out of scope; the mirror takes the tuples.
struct Flow { int slice; int src; int dst; double count; };

class CrossTimeGraph : public LayeredGraph {
 public:
    static int64_t numSamples;   // J/CrossTimeGraph.java:18
    static int numLayer;         // J/CrossTimeGraph.java:19
    // constructGraph_tract / constructGraph_CA  J/CrossTimeGraph.java:25-52,68-95: edge "h-src" -> "(h+1)%T-dst" for every
    // positive flow; sources = EVERY layer-0 vertex that exists in allVertices (:43-47) — also one that only occurs as a
    // destination of slice T-1 and has no out-edge — in the order of `regions` (the reference: its region map's order)
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
    // the graph builders drop them, J/CrossTimeGraph.java:37-38).  files[h] is slice h; candidate sources are all region ids
    // that occur in a positive flow of any slice, ascending (embedding_amd/io.py:read_od_slices does the same).
    static void constructGraphFromOD(CrossTimeGraph& g, const std::vector<std::string>& files) {
        numLayer = (int)files.size();
        std::vector<Flow> flows;
        std::set<int> seen;
        for (size_t h = 0; h < files.size(); h++) {
            std::ifstream in(files[h]);
            if (!in) throw std::runtime_error("cannot open " + files[h]);
            long long a, b; double w;
            while (in >> a >> b >> w) {
                flows.push_back({(int)h, (int)a, (int)b, w});
                if (w > 0) { seen.insert((int)a); seen.insert((int)b); }
            }
        }
        constructGraph(g, flows, std::vector<int>(seen.begin(), seen.end()));
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
                    line += g.nameOfDeviceId(w[(size_t)i * L + j]);
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
    // keepNearestKVertices(k)  J/SpatialGraph.java:29-35 (before sources / alias tables): stable sort by weight descending, first k,
    // outDegree = DoubleStream.sum() — for all vertices at once on the device; edgesOut / outDegree are rebuilt from its result.
    // A vertex with fewer than k edges: the reference's subList throws IndexOutOfBoundsException -> std::out_of_range here.
    void keepNearestKVertices(int k) {
        upload();
        if (dge_graph_keep_top_k(h_, k) != DGE_OK) throw std::out_of_range(dge_last_error());
        std::vector<int64_t> rp((size_t)nv_ + 1); std::vector<int32_t> nbr((size_t)nv_ * k + 1); std::vector<double> w(nbr.size()), od((size_t)nv_ + 1);
        dge_check(dge_graph_get_csr(h_, rp.data(), nbr.data(), w.data(), nullptr, nullptr, od.data(), nv_, (int64_t)nbr.size()));
        for (int32_t v = 0; v < (int32_t)byId_.size(); v++) {
            if (!byId_[v]) continue;
            std::vector<Edge> kept;
            for (int64_t e = rp[v]; e < rp[v + 1]; e++) kept.emplace_back(byId_[v], byId_[nbr[(size_t)e]], w[(size_t)e]);
            byId_[v]->edgesOut = std::move(kept);
            byId_[v]->outDegree = od[v];
        }
        built_ = false;
    }
    // constructGraph_*  J/SpatialGraph.java:37-88: complete graph with w = exp(-100 d) (self loop included), top-10,
    // every vertex a source (order of `names`), sourceWeightSum = DoubleStream.sum() of the outDegrees (:57)
    static void constructGraph(SpatialGraph& g, const std::vector<std::string>& names, const std::vector<double>& weight /* n x n */) {
        size_t n = names.size();
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < n; j++) g.addEdge(names[i], names[j], weight[i * n + j]);
        g.keepNearestKVertices(10);
        g.sourceVertices.clear();
        std::vector<double> od;
        for (const std::string& s : names) { g.sourceVertices.push_back(g.allVertices.at(s)); od.push_back(g.sourceVertices.back()->outDegree); }
        g.sourceWeightSum = java8StreamSum(od);
        g.initiateAliasTables(true);
    }
    static void outputSampleSequence(SpatialGraph& g, const std::string& path) {
        LayeredGraph::numLayer = SpatialGraph::numLayer;              // J/SpatialGraph.java:92
        CrossTimeGraph::write_seq(g, path, numSamples, true);
    }
    // java.util.stream.DoubleStream.sum() (JDK 8: Kahan running sum, final sum + compensation)
    static double java8StreamSum(const std::vector<double>& x) {
        double sum = 0, comp = 0, simple = 0;
        for (double v : x) { double t = v - comp, vv = sum + t; comp = (vv - sum) - t; sum = vv; simple += v; }
        double t = sum + comp;
        return (t != t && (simple - simple) != 0) ? simple : t;
    }
};
inline int64_t SpatialGraph::numSamples = 5000000;
inline int SpatialGraph::numLayer = 8;

// J/DeepWalk.java:32-83
class DeepWalk {
 public:
    static int Year;   // J/DeepWalk.java:25
    // The reference's builder (:73-76) never calls .useHierarchicSoftmax(false), so DL4J trained the hierarchical-softmax
    // term next to the 5 negatives; true (the default, as for the authors' runs) reproduces that, false is the
    // negative-sampling path BASELINE.json's metric is quoted on.
    static bool useHierarchicSoftmax;
    // learnEmbedding: every line of the .seq files is a sentence of whitespace-separated names (DefaultTokenizerFactory,
    // :70); trains with the reference's builder values and writes "name v1 .. vD" lines (:82).
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
inline bool DeepWalk::useHierarchicSoftmax = true;

}  // namespace embedding
