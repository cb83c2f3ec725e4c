// Port of T/LayeredGraphTest.java:12-44 (in its original standalone-Vertex form) to the C++ host mirror (embedding_amd/host/embedding_host.hpp), plus the
// writer-loop / DeepWalk plumbing (.seq -> .vec).  Needs a GPU: built and run by tests/test_gpu_host_mirror.py.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../embedding_amd/host/embedding_host.hpp"
using namespace embedding;

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    std::string tmp = argc > 1 ? argv[1] : "/tmp";
    {   // T/LayeredGraphTest.java:12-44 as written: standalone Vertex / Edge objects, no graph
        LayeredGraph::Vertex org("start", 0), d1("d1", 1), d2("d2", 2), d3("d3", 3);
        LayeredGraph::Edge e1(&org, &d1, 2), e2(&org, &d2, 10), e3(&org, &d3, 8);
        org.addOutEdge(e1); org.addOutEdge(e2); org.addOutEdge(e3);
        org.initiateAliasTable();
        CHECK(org.aliasTable[0] == 1); CHECK(org.aliasTable[1] == 2); CHECK(org.aliasTable[2] == -1);
        CHECK(org.probTable[0] == 0.3); CHECK(org.probTable[1] == 0.8); CHECK(org.probTable[2] == 1.0);
        CHECK(org.outDegree == 20.0);
        CHECK(org.sampleNextVertex(0.05)->id == 1); CHECK(org.sampleNextVertex(0.3)->id == 2); CHECK(org.sampleNextVertex(0.4)->id == 2);
        CHECK(org.sampleNextVertex(0.65)->id == 3); CHECK(org.sampleNextVertex(0.9)->id == 3);
        CHECK(d1.sampleNextVertex() == nullptr);                      // dead end: null, and no draw taken (:106-107)
        LayeredGraph::rnd = Random(42);
        LayeredGraph::Vertex* nx = org.sampleNextVertex();            // x = 0.7275636800328681 -> slot 2, y = 0.18 < 1 -> d3
        CHECK(nx == &d3 && LayeredGraph::rnd.draws() == 1);
    }
    {   // the same through a graph: tables land in the Vertex fields, ids are insertion ordinals, public state is honoured
        LayeredGraph g;
        g.addEdge("start", "d1", 2);
        g.addEdge("start", "d2", 10);
        g.addEdge("start", "d3", 8);
        g.addSourceVertex("start");
        CHECK(g.allEdges.size() == 3 && g.allEdges[1].to->name == "d2" && g.sourceWeightSum == 20.0);
        g.initiateAliasTables();
        LayeredGraph::Vertex& org = *g.allVertices.at("start");
        CHECK(org.aliasTable[0] == 1 && org.aliasTable[1] == 2 && org.aliasTable[2] == -1);
        CHECK(org.probTable[0] == 0.3 && org.probTable[1] == 0.8 && org.probTable[2] == 1.0 && org.outDegree == 20.0);
        CHECK(org.sampleNextVertex(0.05)->id == 1 && org.sampleNextVertex(0.65)->id == 3);
        CHECK(g.allVertices.at("d1")->id == 1 && g.allVertices.at("d3")->id == 3);
        CHECK(g.probTable.size() == 1 && g.probTable[0] == 1.0 && g.aliasTable[0] == -1);
        // a caller edits the public fields (as J/SpatialGraph.java:31-33 does): the next initiateAliasTables() uses them as they stand
        org.edgesOut.pop_back(); org.outDegree = 12.0;
        g.initiateAliasTables();
        CHECK(org.probTable.size() == 2 && org.probTable[0] == 2 * 2.0 / 12.0 && org.aliasTable[0] == 1);
        // an unknown source name: unregistered vertex (J/LayeredGraph.java:182-183); a walk from it is the single token
        LayeredGraph g2;
        g2.addEdge("a", "b", 1);
        g2.addSourceVertex("ghost");
        CHECK(g2.allVertices.count("ghost") == 0 && g2.sourceVertices.size() == 1 && g2.sourceVertices[0]->id == 2);
        g2.initiateAliasTables();
        LayeredGraph::numLayer = 3;
        std::vector<std::string> w = g2.sampleVertexSequence();
        CHECK(w.size() == 1 && w[0] == "ghost");
    }
    {   // java.util.Random mirror: seed 42 -> 0.7275636800328681 ; seed 0 -> 0.730967787376657
        Random r(42); CHECK(r.nextDouble() == 0.7275636800328681);
        Random z(0);  CHECK(z.nextDouble() == 0.730967787376657);
    }
    // cross-time graph: 3 slices x 6 regions, every flow positive
    CrossTimeGraph::numLayer = 3; CrossTimeGraph::numSamples = 2000;
    std::vector<Flow> flows; std::vector<int> regions;
    for (int r = 0; r < 6; r++) regions.push_back(100 + r);
    for (int h = 0; h < 3; h++)
        for (int s = 0; s < 6; s++)
            for (int d = 0; d < 6; d++) flows.push_back({h, 100 + s, 100 + d, (double)(1 + (s * 7 + d * 3 + h) % 5)});
    std::string seq = tmp + "/taxi-crosstime.seq", vec = tmp + "/taxi-deepwalk.vec";
    {
        CrossTimeGraph g;
        CrossTimeGraph::constructGraph(g, flows, regions);
        LayeredGraph::rnd = Random(2017);
        CrossTimeGraph::outputSampleSequence(g, seq);
        CHECK(LayeredGraph::rnd.draws() == 2000 * 3);                 // one draw per node, no dead ends
        // single-call API consumes the same stream as the bulk call
        LayeredGraph::rnd = Random(2017);
        std::ifstream in(seq); std::string line;
        for (int i = 0; i < 50; i++) {
            std::getline(in, line);
            std::vector<std::string> w = g.sampleVertexSequence();
            std::string joined;
            for (size_t j = 0; j < w.size(); j++) joined += (j ? " " : "") + w[j];
            CHECK(joined == line);
            CHECK(w.size() == 3 && w[0].substr(0, 2) == "0-" && w[1].substr(0, 2) == "1-" && w[2].substr(0, 2) == "2-");
        }
    }
    {   // the same graph from the reference's .od edge files (J/Tracts.java:236-264): one "src dst w" line per flow and slice
        std::vector<std::string> files;
        for (int h = 0; h < 3; h++) {
            files.push_back(tmp + "/taxi-h" + std::to_string(h) + ".od");
            std::ofstream out(files.back());
            for (const Flow& f : flows) if (f.slice == h) out << f.src << " " << f.dst << " " << (long long)f.count << "\n";
            out << "100 101 0\n";                                     // zero flows are written by the CA exporter and must be dropped
        }
        CrossTimeGraph g2;
        CrossTimeGraph::constructGraphFromOD(g2, files);
        CHECK(CrossTimeGraph::numLayer == 3 && g2.allVertices.size() == 18 && g2.sourceVertices.size() == 6);
        CHECK(g2.numEdges() == (int64_t)flows.size());
        g2.initiateAliasTables();
        CrossTimeGraph g1;
        CrossTimeGraph::constructGraph(g1, flows, regions);
        g1.initiateAliasTables();
        LayeredGraph::numLayer = 3;
        LayeredGraph::rnd = Random(5); std::vector<int32_t> w1 = g1.sampleVertexSequences(500);
        LayeredGraph::rnd = Random(5); std::vector<int32_t> w2 = g2.sampleVertexSequences(500);
        CHECK(w1 == w2);                                              // same ids, same tables, same stream -> same walks
    }
    {   // spatial graph: position prefix + top-10 prune
        SpatialGraph::numLayer = 3; SpatialGraph::numSamples = 500;
        std::vector<std::string> names; std::vector<double> wt;
        int n = 12;
        for (int i = 0; i < n; i++) names.push_back(std::to_string(100 + i));
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) wt.push_back(std::exp(-100.0 * std::fabs(i - j) * 0.004));
        SpatialGraph g;
        SpatialGraph::constructGraph(g, names, wt);
        CHECK(g.allVertices.at("105")->edgesOut.size() == 10);
        {   // the pruned store is visible in the public fields: nearest first (self loop, w = 1), outDegree = DoubleStream.sum()
            const LayeredGraph::Vertex& v = *g.allVertices.at("105");
            CHECK(v.edgesOut[0].to == &v && v.edgesOut[0].weight == 1.0);
            std::vector<double> ws; for (const auto& e : v.edgesOut) ws.push_back(e.weight);
            CHECK(v.outDegree == SpatialGraph::java8StreamSum(ws));
            for (size_t i = 1; i < v.edgesOut.size(); i++) CHECK(v.edgesOut[i - 1].weight >= v.edgesOut[i].weight);
        }
        LayeredGraph::rnd = Random(7);
        SpatialGraph::outputSampleSequence(g, tmp + "/taxi-spatial.seq");
        std::ifstream in(tmp + "/taxi-spatial.seq"); std::string line; std::getline(in, line);
        CHECK(line.substr(0, 2) == "0-" && line.find(" 1-") != std::string::npos && line.find(" 2-") != std::string::npos);
    }
    {   // DeepWalk.learnEmbedding on both corpora (the "usespatial" directory case, J/DeepWalk.java:47-50), negative sampling alone
        LayeredGraph::numLayer = 3;
        CHECK(DeepWalk::useHierarchicSoftmax);                        // the default is what DL4J's builder leaves on
        DeepWalk::useHierarchicSoftmax = false;
        dge_train_stats st = DeepWalk::learnEmbedding({seq, tmp + "/taxi-spatial.seq"}, vec, 20);
        CHECK(st.pairs > 0);
        std::ifstream in(vec); std::string line; int lines = 0;
        while (std::getline(in, line)) {
            std::istringstream ss(line); std::string name; ss >> name; double x; int cnt = 0;
            while (ss >> x) cnt++;
            CHECK(cnt == 20 && name.find('-') != std::string::npos);  // "h-id v1 .. v20", no header
            lines++;
        }
        CHECK(lines == 36);                                           // shared "h-id" vocabulary: 3 layers x 12 spatial regions (the 6 flow regions are a subset)
    }
    {   // the same with the hierarchical-softmax term DL4J's builder default leaves on (J/DeepWalk.java:73-76)
        DeepWalk::useHierarchicSoftmax = true;
        const std::string vec_hs = tmp + "/taxi-deepwalk-hs.vec";
        dge_train_stats st = DeepWalk::learnEmbedding({seq, tmp + "/taxi-spatial.seq"}, vec_hs, 20, 0, 1);
        CHECK(st.pairs > 0);
        std::ifstream a(vec), b(vec_hs); std::string la, lb; int lines = 0, differ = 0;
        while (std::getline(a, la) && std::getline(b, lb)) { lines++; if (la != lb) differ++; }
        CHECK(lines == 36 && differ > 0);                             // same vocabulary, different vectors
    }
    std::printf("HOST MIRROR OK\n");
    return 0;
}
