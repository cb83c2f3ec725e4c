// dge_algos.h — scalar building blocks shared by the HIP kernels of libdge.so (and, compiled for the
// host, by tests/native/algos_harness.cpp so their logic can be checked without a GPU).
//
//   * java.util.Random LCG + O(log n) jump-ahead        (RNG of J/LayeredGraph.java:14,108,234)
//   * the reference's alias pairing, restated with two ordered bit-sets so that it runs in
//     O(k log k) per table instead of the reference's O(k^2)   (J/LayeredGraph.java:54-82,199-225)
//   * Vose pairing (scalable form, not in the reference)
//   * DoubleStream.sum() of JDK 8                       (J/SpatialGraph.java:33,57)
//
// Everything here is sequential per table / per walk; the kernels give one lane one table or one walk.
// Compile with -ffp-contract=off: `x*k - i` must be a rounded product followed by a subtraction, as in Java.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DGE_HD __host__ __device__ inline
#else
#define DGE_HD inline
#endif

// ---------------------------------------------------------------- java.util.Random
#define DGE_JR_MULT 0x5DEECE66DULL
#define DGE_JR_ADD 0xBULL
#define DGE_JR_MASK ((1ULL << 48) - 1)

DGE_HD uint64_t dge_jr_scramble(int64_t seed) { return ((uint64_t)seed ^ DGE_JR_MULT) & DGE_JR_MASK; }

DGE_HD int32_t dge_jr_next(uint64_t& s, int bits) {
    s = (s * DGE_JR_MULT + DGE_JR_ADD) & DGE_JR_MASK;
    return (int32_t)(s >> (48 - bits));
}

DGE_HD double dge_jr_next_double(uint64_t& s) {
    int64_t hi = (int64_t)dge_jr_next(s, 26);
    int64_t lo = (int64_t)dge_jr_next(s, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

// advance by n LCG steps: compose the affine map with itself by squaring (mod 2^48)
DGE_HD uint64_t dge_jr_jump(uint64_t s, uint64_t n) {
    uint64_t acc_m = 1, acc_p = 0, cur_m = DGE_JR_MULT, cur_p = DGE_JR_ADD;
    while (n) {
        if (n & 1) { acc_m = acc_m * cur_m; acc_p = acc_p * cur_m + cur_p; }
        cur_p = (cur_m + 1) * cur_p;
        cur_m = cur_m * cur_m;
        n >>= 1;
    }
    return (acc_m * s + acc_p) & DGE_JR_MASK;
}

// ---------------------------------------------------------------- word2vec LCG (mod 2^64) + splitmix64
#define DGE_W2V_MULT 25214903917ULL

DGE_HD uint64_t dge_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

DGE_HD uint64_t dge_w2v_jump(uint64_t s, uint64_t n) {
    uint64_t acc_m = 1, acc_p = 0, cur_m = DGE_W2V_MULT, cur_p = 11;
    while (n) {
        if (n & 1) { acc_m = acc_m * cur_m; acc_p = acc_p * cur_m + cur_p; }
        cur_p = (cur_m + 1) * cur_p;
        cur_m = cur_m * cur_m;
        n >>= 1;
    }
    return acc_m * s + acc_p;
}

// ---------------------------------------------------------------- one alias draw
// J/LayeredGraph.java:104-116 / :234-242: i=(int)(x*k); y=x*k-i; y<prob[i] ? i : alias[i].
// Returns the slot index and y through *y_out.
DGE_HD int64_t dge_alias_slot(double x, int64_t k, double* y_out) {
    double xk = x * (double)k;
    int64_t i = (int64_t)xk;
    if (i > k - 1) i = k - 1;
    *y_out = xk - (double)i;
    return i;
}

// ---------------------------------------------------------------- DoubleStream.sum() (JDK 8)
DGE_HD double dge_java8_stream_sum(const double* x, int64_t n) {
    double sum = 0.0, comp = 0.0, simple = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double tmp = x[i] - comp;
        double velvel = sum + tmp;
        comp = (velvel - sum) - tmp;
        sum = velvel;
        simple += x[i];
    }
    double tmp = sum + comp;
    if (tmp != tmp && (simple - simple) != 0.0 && simple == simple) return simple;   // NaN result, infinite simple sum
    return tmp;
}

// ---------------------------------------------------------------- 4-level 64-ary bit-set
// Ordered set over [0,k) with insert / erase / successor in <= 4 word operations per level.
struct dge_bitset4 {
    uint64_t* w[4];
    int64_t nw[4];
};

DGE_HD int64_t dge_bs_words(int64_t k) {
    int64_t n = (k + 63) >> 6, tot = 0;
    if (n < 1) n = 1;
    for (int l = 0; l < 4; l++) { tot += n; n = (n + 63) >> 6; }
    return tot;
}

DGE_HD void dge_bs_init(dge_bitset4& s, uint64_t* mem, int64_t k) {
    int64_t n = (k + 63) >> 6;
    if (n < 1) n = 1;
    for (int l = 0; l < 4; l++) {
        s.w[l] = mem; s.nw[l] = n;
        for (int64_t i = 0; i < n; i++) mem[i] = 0;
        mem += n; n = (n + 63) >> 6;
    }
}

DGE_HD void dge_bs_set(dge_bitset4& s, int64_t i) {
    for (int l = 0; l < 4; l++) {
        int64_t word = i >> 6;
        uint64_t old = s.w[l][word];
        s.w[l][word] = old | (1ULL << (i & 63));
        if (old != 0) return;
        i = word;
    }
}

DGE_HD void dge_bs_clear(dge_bitset4& s, int64_t i) {
    for (int l = 0; l < 4; l++) {
        int64_t word = i >> 6;
        uint64_t nv = s.w[l][word] & ~(1ULL << (i & 63));
        s.w[l][word] = nv;
        if (nv != 0) return;
        i = word;
    }
}

// smallest member >= i, or -1
DGE_HD int64_t dge_bs_next(const dge_bitset4& s, int64_t i) {
    int64_t pos = i;
    int l = 0;
    for (;;) {
        int64_t word = pos >> 6;
        if (word >= s.nw[l]) return -1;
        uint64_t m = s.w[l][word] & (~0ULL << (pos & 63));
        if (m) { pos = (word << 6) + __builtin_ctzll(m); break; }
        if (l == 3) return -1;
        pos = word + 1;
        l++;
    }
    while (l > 0) {
        l--;
        uint64_t m = s.w[l][pos];
        pos = (pos << 6) + __builtin_ctzll(m);
    }
    return pos;
}

// ---------------------------------------------------------------- the reference's alias pairing
// Vertex.initiateAliasTable (J/LayeredGraph.java:54-82) and the identical loop over the source
// vertices (J/LayeredGraph.java:199-225).  The reference scans l2 = 0..k-1 for every l1; scanning a
// slot that matches neither branch has no side effect, so the scan is replaced by successor queries on
//   U = { i : alias[i] == -1 && prob[i] < 1 }   (candidates of the first branch, :70)
//   O = { i : prob[i] > 1 }                      (candidates of the second branch, :73; such a slot
//                                                 never has an alias: aliases are only given to slots
//                                                 below 1 and prob never increases)
// The floating-point operations and their order are exactly the reference's, so prob[] and alias[]
// are bit-identical to the Java arrays.  scratch: 2*dge_bs_words(k) words.
DGE_HD void dge_alias_reference(const double* w, int64_t k, double total, double* prob, int32_t* alias,
                                uint64_t* scratch) {
    dge_bitset4 U, O;
    dge_bs_init(U, scratch, k);
    dge_bs_init(O, scratch + dge_bs_words(k), k);
    for (int64_t i = 0; i < k; i++) {
        double p = (double)k * w[i] / total;                       // :62
        prob[i] = p; alias[i] = -1;                                // :58-59
        if (p < 1.0) dge_bs_set(U, i);
        else if (p > 1.0) dge_bs_set(O, i);
    }
    for (int64_t l1 = 0; l1 < k; l1++) {                           // :65
        double p1 = prob[l1];
        if (p1 == 1.0 || alias[l1] != -1) continue;                // :66
        if (p1 < 1.0) {
            // only the second branch (:73) can fire: first slot above 1
            int64_t g = dge_bs_next(O, 0);
            if (g >= 0) {
                alias[l1] = (int32_t)g;                            // :74
                dge_bs_clear(U, l1);
                double pg = prob[g] - (1 - p1);                    // :75
                prob[g] = pg;
                if (!(pg > 1.0)) { dge_bs_clear(O, g); if (pg < 1.0) dge_bs_set(U, g); }
            }                                                       // :77 break
        } else if (p1 > 1.0) {
            // first branch (:70) for every un-aliased slot below 1, in index order, while l1 stays above 1
            int64_t c = dge_bs_next(U, 0), last = -1;
            while (c >= 0 && p1 > 1.0) {
                alias[c] = (int32_t)l1;                            // :71
                dge_bs_clear(U, c);
                p1 = p1 - (1 - prob[c]);                           // :72
                last = c;
                c = dge_bs_next(U, c + 1);
            }
            prob[l1] = p1;
            if (!(p1 > 1.0)) {
                dge_bs_clear(O, l1);
                if (p1 < 1.0) {
                    // l1 fell below 1 at slot `last`; the scan goes on from last+1 and now only :73 can fire
                    int64_t g = dge_bs_next(O, last + 1);
                    if (g >= 0) {
                        alias[l1] = (int32_t)g;
                        double pg = prob[g] - (1 - p1);
                        prob[g] = pg;
                        if (!(pg > 1.0)) { dge_bs_clear(O, g); if (pg < 1.0) dge_bs_set(U, g); }
                    } else {
                        dge_bs_set(U, l1);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- Vose pairing (scalable form)
// prob initialised as the reference does (:62); "small" stack grows up from scratch[0], "large" stack
// grows down from scratch[k-1]; left-overs get prob 1 / alias -1.  scratch: k int32.
DGE_HD void dge_alias_vose(const double* w, int64_t k, double total, double* prob, int32_t* alias,
                           int32_t* scratch) {
    int64_t ns = 0, nl = 0;
    for (int64_t i = 0; i < k; i++) {
        double p = (double)k * w[i] / total;
        alias[i] = -1; prob[i] = p;
        if (p < 1.0) scratch[ns++] = (int32_t)i; else scratch[k - 1 - (nl++)] = (int32_t)i;
    }
    // (a large slot that stays >= 1 would go back on top of its stack and be taken again at once: it stays in registers instead)
    while (ns > 0 && nl > 0) {
        const int32_t l = scratch[k - nl]; nl--;
        double pl = prob[l];
        for (;;) {
            const int32_t s = scratch[--ns];
            alias[s] = l;
            pl = (pl + prob[s]) - 1.0;
            if (pl < 1.0) { prob[l] = pl; scratch[ns++] = l; break; }
            if (ns == 0) { prob[l] = pl; scratch[k - 1 - (nl++)] = l; break; }
        }
    }
    while (ns > 0) { int32_t s = scratch[--ns]; prob[s] = 1.0; }
    while (nl > 0) { int32_t l = scratch[k - nl]; nl--; prob[l] = 1.0; }
}

// ---------------------------------------------------------------- Huffman paths (hierarchical softmax)
// The tree word2vec.c's CreateBinaryTree builds over counts sorted descending: leaves are consumed from the rare end,
// merged nodes are appended in creation order (so they form a second, ascending queue); of two equal heads the MERGED
// node is taken first (the leaf test is a strict <); the second node taken gets branch bit 1.  Inner node V+a is row a
// of syn1, the root is row V-2.  Output in CSR form: path of word r = points[off[r] .. off[r+1]) from the root down,
// bit d of codes[r] = branch at step d.  Returns the longest code; a code longer than 40 (word2vec.c's MAX_CODE_LENGTH)
// is reported, not truncated.  Host only: a serial two-queue merge, O(V).
#include <stddef.h>
#include <vector>
inline int dge_huffman_paths(const int64_t* counts, int64_t V, std::vector<int64_t>& off, std::vector<int32_t>& points,
                             std::vector<uint64_t>& codes, std::vector<int64_t>* node_weight = nullptr) {
    off.assign((size_t)V + 1, 0); points.clear(); codes.assign((size_t)V, 0);
    if (V < 2) return 0;
    const int64_t n_inner = V - 1;
    std::vector<int64_t> weight((size_t)n_inner);          // merged nodes, in creation order
    std::vector<int32_t> up((size_t)(V + n_inner), -1);    // parent (as inner-node row) of leaf r / of inner node V+a
    std::vector<uint8_t> bit((size_t)(V + n_inner), 0);
    int64_t leaf = V - 1, merged = 0;                      // heads of the two queues
    for (int64_t a = 0; a < n_inner; a++) {
        for (int pick = 0; pick < 2; pick++) {
            int64_t node;
            if (leaf >= 0 && (merged >= a || counts[leaf] < weight[(size_t)merged])) node = leaf--;
            else node = V + merged++;
            up[(size_t)node] = (int32_t)a; bit[(size_t)node] = (uint8_t)pick;
            weight[(size_t)a] = pick == 0 ? (node < V ? counts[node] : weight[(size_t)(node - V)])
                                          : weight[(size_t)a] + (node < V ? counts[node] : weight[(size_t)(node - V)]);
        }
    }
    if (node_weight) *node_weight = weight;                // tokens below inner node a: non-decreasing in a (the merge takes the two lightest)
    // depth of every inner node from the root (row V-2), parents are created after their children
    std::vector<int32_t> depth((size_t)n_inner, 0);
    for (int64_t a = n_inner - 2; a >= 0; a--) depth[(size_t)a] = depth[(size_t)up[(size_t)(V + a)]] + 1;
    int longest = 0;
    for (int64_t r = 0; r < V; r++) {
        int len = depth[(size_t)up[(size_t)r]] + 1;
        off[(size_t)r + 1] = off[(size_t)r] + len;
        if (len > longest) longest = len;
    }
    if (longest > 64) return longest;
    points.resize((size_t)off[(size_t)V]);
    for (int64_t r = 0; r < V; r++) {
        int64_t node = r; uint64_t c = 0;
        int32_t* out = points.data() + off[(size_t)r];
        for (int d = (int)(off[(size_t)r + 1] - off[(size_t)r]) - 1; d >= 0; d--) {
            c |= (uint64_t)bit[(size_t)node] << d;
            out[d] = up[(size_t)node];
            node = V + up[(size_t)node];
        }
        codes[(size_t)r] = c;
    }
    return longest;
}
