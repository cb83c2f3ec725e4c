// Test-only host build of embedding_amd/csrc/dge_algos.h: lets the CPU test-suite check the scalar
// logic the HIP kernels run per lane (bit-set alias pairing, Vose, Java LCG jump) against the oracle
// without a GPU.  Not part of libdge.so.
#include "../../embedding_amd/csrc/dge_algos.h"
#include <vector>
extern "C" {
void harness_alias_reference(const double* w, int64_t k, double total, double* prob, int32_t* alias) {
    std::vector<uint64_t> scratch(2 * dge_bs_words(k) + 1);
    dge_alias_reference(w, k, total, prob, alias, scratch.data());
}
void harness_alias_vose(const double* w, int64_t k, double total, double* prob, int32_t* alias) {
    std::vector<int32_t> scratch(k + 1);
    dge_alias_vose(w, k, total, prob, alias, scratch.data());
}
uint64_t harness_jr_jump(int64_t seed, uint64_t n) { return dge_jr_jump(dge_jr_scramble(seed), n); }
double harness_jr_next_double(uint64_t* s) { return dge_jr_next_double(*s); }
uint64_t harness_mix64(uint64_t x) { return dge_mix64(x); }
uint64_t harness_w2v_jump(uint64_t s, uint64_t n) { return dge_w2v_jump(s, n); }
double harness_stream_sum(const double* x, int64_t n) { return dge_java8_stream_sum(x, n); }
int harness_huffman(const int64_t* counts, int64_t V, int64_t* off, int32_t* points, int64_t cap, uint64_t* codes) {
    std::vector<int64_t> o; std::vector<int32_t> p; std::vector<uint64_t> c;
    int longest = dge_huffman_paths(counts, V, o, p, c);
    for (int64_t i = 0; i <= V; i++) off[i] = o[i];
    for (int64_t i = 0; i < (int64_t)p.size() && i < cap; i++) points[i] = p[i];
    for (int64_t i = 0; i < V; i++) codes[i] = c[i];
    return longest;
}
int64_t harness_bitset_selftest(int64_t k, uint64_t seed, int64_t ops) {
    std::vector<uint64_t> mem(dge_bs_words(k) + 1);
    dge_bitset4 s; dge_bs_init(s, mem.data(), k);
    std::vector<char> ref(k, 0);
    uint64_t r = seed; int64_t bad = 0;
    for (int64_t t = 0; t < ops; t++) {
        r = dge_mix64(r); int64_t i = (int64_t)(r % (uint64_t)k); int op = (int)((r >> 40) % 3);
        if (op == 0) { dge_bs_set(s, i); ref[i] = 1; }
        else if (op == 1) { dge_bs_clear(s, i); ref[i] = 0; }
        else { int64_t e = -1; for (int64_t j = i; j < k; j++) if (ref[j]) { e = j; break; }
               if (dge_bs_next(s, i) != e) bad++; }
    }
    return bad;
}
}
