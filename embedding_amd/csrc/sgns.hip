// sgns.hip — vocabulary, unigram table and the skip-gram negative-sampling trainer of libdge.so (gfx950).
//
// Replaces `new Word2Vec.Builder()...build(); w2v.fit()` (J/DeepWalk.java:73-79).  The arithmetic of that call
// lives in DL4J-NLP 0.7.2 / ND4J-native 0.7.2 (not under /root/reference); what is implemented is the word2vec
// skip-gram negative-sampling update with DL4J's pair enumeration, as restated in oracle/dge_oracle.c
// (SURVEY.md §3.3, row a9).
//
// HBM layout: syn0, syn1neg (and syn1 with use_hs)  float32 [V x stride], stride = round_up(dim, 64) floats (zero padded)
// so that a row is 1..8 chunks of 256 B.
// Work decomposition: one 16-lane group ("worker") per walk; 4 workers per wave.  A worker owns D/16 floats of
// every row it touches in registers, dot products are 16-lane xor-butterflies, and the K negative rows of a pair are in
// flight together.  workers == 1 gives the in-order schedule the oracle follows, bit for bit.
// Two trainer kernels (dge_train_config.update_policy):
//   k_sgns_train         rows move 4 B per lane (lane j owns elements 64c+16m+j): in-order plain accesses, or Hogwild
//                        with agent-scope loads and memory-side float atomics; also carries the hierarchical softmax.
//   k_sgns_train_locked  rows move 16 B per lane under per-row commit locks (the default on large vocabularies);
//                        HOTMIX: the vocabulary's head takes atomics instead; PART: one block of the multi-GPU schedule.
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "dge_algos.h"
#include "dge_internal.h"

#define EXP_TABLE_SIZE 1000
#define MAX_EXP 6
#define NEG_BATCH 5
#ifndef DGE_LOCKED_WAVES
#define DGE_LOCKED_WAVES 3
#endif
#ifndef DGE_HOTMIX_WAVES
#define DGE_HOTMIX_WAVES 3
#endif
#ifndef DGE_HS_WAVES
#define DGE_HS_WAVES 4
#endif

struct EventPair { hipEvent_t a, b; int kind; };

struct dge_model {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    dge_train_config cfg{};
    int64_t V = 0;
    int32_t D = 0, stride = 0, NV = 0;
    int64_t T = 0;
    int64_t total_words = 0;
    double neg_collision = 1.0;                 // sum of squared negative-sampling probabilities: P(two draws hit one row)
    int64_t hot_rows_auto = 0;                  // head rows that policy 7 keeps out of the lock protocol (see dge_model_create)
    int64_t hot_rows_serial = 0;                // head rows whose own pairs, serialised by the row's lock, would outlast a launch
    int n_cus = 256;
    float *d_syn0 = nullptr, *d_syn1neg = nullptr, *d_snap = nullptr;
    // hierarchical softmax (cfg.use_hs): inner-node table and the Huffman paths in CSR form
    float* d_syn1 = nullptr;
    int64_t* d_hs_off = nullptr; int32_t* d_hs_points = nullptr; uint64_t* d_hs_codes = nullptr;
    std::vector<int64_t> h_hs_off; std::vector<int32_t> h_hs_points; std::vector<uint64_t> h_hs_codes;
    std::vector<float> h_syn1;

    int32_t* d_vocab_ids = nullptr;
    int64_t* d_counts = nullptr;
    int32_t* d_remap = nullptr;
    int32_t* d_table = nullptr;
    float* d_exp = nullptr;
    // per-call work buffers
    int64_t cap_rows = 0; int32_t cap_L = 0;
    int32_t* d_sen = nullptr; int64_t* d_len = nullptr; int64_t* d_wb = nullptr;
    void* d_scan_tmp = nullptr; size_t scan_tmp_bytes = 0;
    unsigned long long* d_counters = nullptr;   // [0]=pairs [1]=words
    int* d_locks = nullptr;                     // commit-lock word per syn1neg row
    // host mirrors for the read-back API
    std::vector<float> h_syn0, h_syn1neg;
    std::vector<int32_t> h_vocab_ids, h_table;
    std::vector<int64_t> h_counts;
    // stats
    std::vector<EventPair> pending;
    double kernel_ms = 0, walk_ms = 0;
    int64_t launches = 0;
    int last_policy = -1; int64_t last_workers = 0; int32_t last_hot_rows = 0;   // what the latest launch ran with
    int32_t part_n = 1, part_ctx = 0, part_tgt = 0;                              // block schedule (dge_model_set_partition)
};

// ------------------------------------------------------------------------------------------ vocabulary
__global__ void k_count_tokens(const int32_t* __restrict__ walks, int64_t n, int32_t NV, unsigned long long* counts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t t = walks[i];
        if (t >= 0 && t < NV) atomicAdd(&counts[t], 1ULL);
    }
}

__global__ void k_iota_i32(int32_t* p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

__global__ void k_count_kept(const int64_t* sorted_counts, int64_t n, int64_t min_count, unsigned long long* out) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        c += (sorted_counts[i] >= min_count && sorted_counts[i] > 0) ? 1ULL : 0ULL;
    for (int o = 32; o > 0; o >>= 1) c += (unsigned long long)__shfl_xor((long long)c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

__global__ void k_scatter_remap(const int32_t* vocab_ids, int64_t V, int32_t* remap) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < V) remap[vocab_ids[r]] = (int32_t)r;
}

// word2vec.c InitUnigramTable without the serial loop.  The loop advances the word index i by at most one per
// slot a, whenever a/T > cum[i]; with j(a) = #{i : cum[i] < a/T} this is i(a+1) = min(i(a)+1, j(a)), whose
// closed form is i(a) = a + min(0, min_{b<a}(j(b) - b - 1)): a binary search, an exclusive prefix-min, a clamp.
__global__ void k_table_chase(const double* __restrict__ cum, int64_t V, int64_t T, int32_t* g) {
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= T) return;
    double x = (double)a / (double)T;
    int64_t lo = 0, hi = V;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (cum[mid] < x) lo = mid + 1; else hi = mid; }
    g[a] = (int32_t)(lo - a - 1);
}
__global__ void k_table_fill(const int32_t* __restrict__ m, int64_t V, int64_t T, int32_t* table) {
    int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= T) return;
    int64_t i = a + (int64_t)min(0, m[a]);
    table[a] = (int32_t)min(i, V - 1);
}

// word2vec.c InitNet: syn0[a][b] = ((lcg & 0xFFFF)/65536 - 0.5)/dim, one LCG stream over the whole table
__global__ void k_init_syn0(float* syn0, int64_t V, int32_t D, int32_t stride, uint64_t seed) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= V) return;
    uint64_t s = dge_w2v_jump(seed, (uint64_t)r * (uint64_t)D);
    float* row = syn0 + r * stride;
    for (int b = 0; b < D; b++) {
        s = s * DGE_W2V_MULT + 11;
        row[b] = (((float)(s & 0xFFFF) / (float)65536) - 0.5f) / (float)D;
    }
    for (int b = D; b < stride; b++) row[b] = 0.0f;
}

// vertex ids -> vocabulary rows, out-of-vocabulary tokens dropped and the walk left-packed (word2vec / DL4J
// filter the sentence before windowing); len = tokens kept
__global__ void k_remap_compact(const int32_t* __restrict__ walks, int64_t n_rows, int32_t L, const int32_t* __restrict__ remap,
                                int32_t NV, int32_t* __restrict__ sen, int64_t* __restrict__ len_out) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t* in = walks + r * L;
    int32_t* out = sen + r * L;
    int len = 0;
    for (int j = 0; j < L; j++) {
        int32_t t = in[j];
        int32_t v = (t >= 0 && t < NV) ? remap[t] : -1;
        if (v >= 0) out[len++] = v;
    }
    for (int j = len; j < L; j++) out[j] = -1;
    len_out[r] = len;
}

// ------------------------------------------------------------------------------------------ trainer
struct TrainParams {
    const int32_t* sen; const int64_t* len; const int64_t* wb;
    float* syn0; float* syn1neg; const int32_t* table; const float* exp_table;
    int64_t n_rows; int32_t L, W, K, stride;
    int64_t V, T;
    uint64_t seed;
    int64_t gidx_base;        // (epoch*total_walks + walk_index_base): RNG stream key of row 0
    int64_t words_done_base;  // epoch*total_words + words_before
    int64_t all_words;        // epochs*total_words
    double words_scale;
    float alpha0, min_alpha;
    int64_t n_workers;
    unsigned long long* counters;
    int* locks;               // one commit-lock word per syn1neg row (all zero between launches)
    float* syn1;              // hierarchical softmax: inner-node rows, Huffman paths (null when off)
    const int64_t* hs_off; const int32_t* hs_points; const uint64_t* hs_codes;
    int32_t hs_hot0, hs_n_hot; // inner nodes [hs_hot0, hs_hot0 + hs_n_hot) — the ones nearest the root — combine in LDS
    int32_t hs_drain;         // an LDS accumulator is drained to memory every hs_drain additions
    int32_t hot_rows;         // policy 7: vocabulary rows [0, hot_rows) — the most frequent — are never locked, they take atomics
    // multi-GPU block schedule (dge_model_set_partition): only pairs whose context row is in partition part_ctx and whose
    // centre row is in partition part_tgt (row % part_n) are trained; negatives are moved into partition part_tgt
    int32_t part_n, part_ctx, part_tgt;
    int32_t filler_row;       // a row index whose offset is outside every table descriptor (see row_load): loads of it cost no traffic
    int32_t big_seg_shift;    // BIG: 0, or (tests) a smaller segment size than the 4 GiB window allows
    int32_t syn0_free;        // HOTMIX kernels: the pair's syn0 row is never locked either (read agent-scope, updated with atomics)
};

// PART: which of a walk's (<= 64, register-resident) tokens lie in partition `part`: bit j of the result = token j.
// Lane j of the group holds tokens j, j+16, j+32, j+48; a ballot collects 16 of them at a time.
__device__ __forceinline__ uint64_t part_token_mask(int32_t tk0, int32_t tk1, int32_t tk2, int32_t tk3, int32_t n, int32_t part) {
    const int sh = threadIdx.x & 48;
    uint64_t m = (uint64_t)((__ballot(tk0 >= 0 && tk0 % n == part) >> sh) & 0xFFFFull);
    m |= (uint64_t)((__ballot(tk1 >= 0 && tk1 % n == part) >> sh) & 0xFFFFull) << 16;
    m |= (uint64_t)((__ballot(tk2 >= 0 && tk2 % n == part) >> sh) & 0xFFFFull) << 32;
    m |= (uint64_t)((__ballot(tk3 >= 0 && tk3 % n == part) >> sh) & 0xFFFFull) << 48;
    return m;
}
// PART: a block visits every walk of the batch for a few of its pairs (at 8 ranks: 6 of 383), so the next walk's length,
// word offset and tokens are fetched while the current walk is trained
__device__ __forceinline__ void walk_fetch(const TrainParams& p, int64_t w, int L, int lane, int& len, int64_t& wb,
                                           int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
    len = 0; wb = 0; t0 = t1 = t2 = t3 = -1;
    if (w < p.n_rows) {
        len = (int)p.len[w]; wb = p.wb[w];
        const int32_t* sen = p.sen + w * L;
        if (lane < L) t0 = sen[lane];
        if (lane + 16 < L) t1 = sen[lane + 16];
        if (lane + 32 < L) t2 = sen[lane + 32];
        if (lane + 48 < L) t3 = sen[lane + 48];
    }
}
__device__ __forceinline__ int first_bit_from(uint64_t m, int from, int none) {       // lowest set bit >= from, else `none`
    const uint64_t r = from < 64 ? (m >> from) : 0ull;
    return r ? from + (int)__builtin_ctzll(r) : none;
}

// a negative drawn from the whole table, moved to the row of partition `part` nearest below it: rows are ordered by
// count, so the row keeps (almost exactly) the frequency rank it was drawn with
__device__ __forceinline__ int32_t part_row(int32_t t, int32_t n, int32_t part, int64_t V) {
    int32_t r = (t / n) * n + part;
    if (r >= V) r -= n;
    return r;
}

template <int DCH> struct Row { float4 v[DCH]; };
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Cache policy of the table traffic.  The eight XCDs have private L2s that are not coherent with each other, and a
// plain store parks its line dirty in the writer's L2: with plain loads/stores every XCD would train its own stale
// copy of a row and the last write-back would win (measured: >90 % of the updates lost on a 10 MB table).  So the
// Hogwild schedules move rows with agent-scope (sc1) loads and either write-through (sc1) stores or memory-side
// float atomics; the in-order schedule (one worker, one CU) keeps plain accesses.
//   POL 0  plain loads / plain stores            (workers == 1: bit-exact with the oracle)
//   POL 1  sc1 loads / sc1 write-through stores  (Hogwild, row granularity: last writer of a row wins)
//   POL 2  sc1 loads / float atomic adds         (Hogwild, element granularity: no update is lost)
//   (policy 5, every row update under a per-row commit lock, has its own kernel: k_sgns_train_locked)
template <int POL> struct Policy {
    static constexpr int LOAD_AUX = POL == 0 ? 0 : 16;    // aux bit 4 = sc1 on gfx950
    static constexpr int STORE_AUX = POL == 0 ? 0 : 16;
    static constexpr bool ATOMIC = POL == 2;              // updates are float atomics
};

// try-lock of one row: true when this lane took it.  The caller makes the row's load address depend on the result,
// so the load cannot be issued before the exchange has returned.
__device__ __forceinline__ bool row_trylock(int* locks, int32_t row) { return atomicExch(&locks[row], 1) == 0; }
// the row's write-through stores are drained (vmcnt(0), which the workgroup-scope release fence emits) before the
// lock word is cleared with an agent-scope store
// Before a lock word is cleared, the row's stores must be visible to every XCD.  Draining the wave's stores
// (s_waitcnt vmcnt(0)) is NOT enough even for sc1 "write-through" stores: measured with dge_selftest_locked_rows,
// 256..1024 hot rows lose up to ~40 of 10^4 increments per row that way.  None are lost with an agent-scope release
// (buffer_wbl2 sc1 + vmcnt(0)) — but that fence costs the trainer a factor 19 — and none with the per-line commit
// probes below, with or without an acquire on the reading side (the sc1 loads are enough there).
// consume the probes' return values: forces the s_waitcnt on them (and, being a workgroup-scope release, on the stores)
__device__ __forceinline__ void row_commit_wait(float probes) {
    asm volatile("" ::"v"(probes));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
// STRICT commit: the row is known to be in memory (probes), the lock word is cleared by a memory-side atomic and so
// changes promptly where the try-lock exchanges execute.  Relaxed commit: the lock word follows the row's write-through
// stores as one more write-through store — the same path, which is what keeps the overtaking of data by a re-lock rare
// (measured: an atomic unlock there loses 9 % of a 1024-row hot set's updates instead of 1.5 %).
template <bool STRICT>
__device__ __forceinline__ void row_unlock(int* locks, int32_t row) {
    if (STRICT) (void)__hip_atomic_exchange(&locks[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(&locks[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float group16_sum(float p) {
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    p += __shfl_xor(p, 4);
    p += __shfl_xor(p, 8);
    return p;
}

// one table seen through a buffer descriptor: byte offset of (row, lane) = row*stride*4 + lane*16 (< 4 GiB)
// Tables of 4 GiB and more (cfg5: 10 M rows x 256 floats = 10 GB) do not fit one descriptor's 32-bit window: their
// accesses build the descriptor of the row's SEGMENT (a power-of-two number of rows that fits a 4 GiB window).  That
// descriptor can differ between the four groups of a wave, so the compiler serialises the instruction per distinct
// segment (a "waterfall"); the common case keeps the single table-wide descriptor (template parameter BIG of the kernels).
struct TableView {
    __amdgpu_buffer_rsrc_t rsrc;
    float* base;
    uint32_t row_bytes;
    uint32_t seg_shift;       // BIG: rows per segment = 1 << seg_shift (the largest power of two whose rows fit a 4 GiB window)
    bool big;
};
__device__ __forceinline__ TableView make_view(float* base, int64_t rows, int stride, int seg_shift_override = 0) {
    TableView t;
    t.base = base;
    t.row_bytes = (uint32_t)stride * 4u;
    t.big = (uint64_t)rows * (uint64_t)stride * 4ull >= 0xFFFFFFFFull;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, t.big ? 0 : (int)(uint32_t)(rows * stride * 4), 0x00020000);   // unused when BIG
    t.seg_shift = 31u - (uint32_t)__builtin_clz(0xFFFFFFFFu / t.row_bytes);
    if (seg_shift_override > 0 && (uint32_t)seg_shift_override < t.seg_shift) t.seg_shift = (uint32_t)seg_shift_override;   // tests: tiny segments
    return t;
}
// BIG: the descriptor of the 4 GiB-window segment that holds `row`, and the row's byte offset inside it.  The four groups
// of a wave mostly land in the same segment (a 10 GB table has three), so the per-descriptor serialisation the compiler
// emits ("waterfall") runs once or twice instead of once per distinct row.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_view(const TableView& t, int32_t row, uint32_t& row_off) {
    const uint32_t seg = (uint32_t)row >> t.seg_shift;
    row_off = ((uint32_t)row & ((1u << t.seg_shift) - 1u)) * t.row_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(t.base + ((size_t)seg << t.seg_shift) * (t.row_bytes / 4), 0, (int)(uint32_t)(t.row_bytes << t.seg_shift), 0x00020000);
}

// Lane j of a 16-lane group owns elements {64c + 16m + j : m = 0..3} of chunk c (kept as v[c].{x,y,z,w}): every
// memory instruction of a group then touches 64 CONTIGUOUS bytes of the row, which is the shape the memory-side
// float atomics want (one 64-B request per group instead of four) and costs the loads nothing (HBM-bound).
template <int DCH, int AUX, bool BIG>
__device__ __forceinline__ void row_load(Row<DCH>& r, const TableView& t, int32_t row, int lane) {
    if (BIG) {
        uint32_t ro;
        const __amdgpu_buffer_rsrc_t rs = row_view(t, row, ro);
        const uint32_t o = ro + (uint32_t)lane * 4u;
#pragma unroll
        for (int c = 0; c < DCH; c++) {
            r.v[c].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + c * 256u), 0, AUX));
            r.v[c].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + c * 256u + 64u), 0, AUX));
            r.v[c].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + c * 256u + 128u), 0, AUX));
            r.v[c].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + c * 256u + 192u), 0, AUX));
        }
        return;
    }
    const uint32_t off = (uint32_t)row * t.row_bytes + (uint32_t)lane * 4u;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        r.v[c].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rsrc, (int)(off + c * 256u), 0, AUX));
        r.v[c].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rsrc, (int)(off + c * 256u + 64u), 0, AUX));
        r.v[c].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rsrc, (int)(off + c * 256u + 128u), 0, AUX));
        r.v[c].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rsrc, (int)(off + c * 256u + 192u), 0, AUX));
    }
}
// A batch slot without a row (a filler of a partial batch, a row whose lock was not won) loads the row `filler_row`: an index
// whose byte offset lies beyond the table descriptor's range, which the hardware answers with zeros WITHOUT touching memory
// (loading the centre's row instead cost cfg5 4 % and K = 20 at D = 256 12 %).  It is a ROW index, chosen once per launch, so the
// loads keep the plain address arithmetic of a real row: selecting an out-of-range OFFSET per load instruction, or branching
// between the two forms, made the common full batch 8-50 % slower.  Tables of 4 GiB and more (segment descriptors) keep the
// centre's row as filler.
template <int DCH, int AUX, bool BIG>
__device__ __forceinline__ void row_store(const Row<DCH>& r, const TableView& t, int32_t row, int lane) {
    if (BIG) {
        uint32_t ro;
        const __amdgpu_buffer_rsrc_t rs = row_view(t, row, ro);
        const uint32_t o = ro + (uint32_t)lane * 4u;
#pragma unroll
        for (int c = 0; c < DCH; c++) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].x), rs, (int)(o + c * 256u), 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].y), rs, (int)(o + c * 256u + 64u), 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].z), rs, (int)(o + c * 256u + 128u), 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].w), rs, (int)(o + c * 256u + 192u), 0, AUX);
        }
        return;
    }
    const uint32_t off = (uint32_t)row * t.row_bytes + (uint32_t)lane * 4u;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].x), t.rsrc, (int)(off + c * 256u), 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].y), t.rsrc, (int)(off + c * 256u + 64u), 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].z), t.rsrc, (int)(off + c * 256u + 128u), 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.v[c].w), t.rsrc, (int)(off + c * 256u + 192u), 0, AUX);
    }
}
// row += g * x, element-wise float atomics at the memory side (64 contiguous bytes per group and instruction)
template <int DCH>
__device__ __forceinline__ void row_atomic_axpy(const TableView& t, int32_t row, int lane, float g, const Row<DCH>& x) {
    float* p = t.base + (size_t)row * (t.row_bytes / 4) + lane;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        atomicAdd(p + c * 64 + 0, g * x.v[c].x);
        atomicAdd(p + c * 64 + 16, g * x.v[c].y);
        atomicAdd(p + c * 64 + 32, g * x.v[c].z);
        atomicAdd(p + c * 64 + 48, g * x.v[c].w);
    }
}
template <int DCH>
__device__ __forceinline__ float row_dot(const Row<DCH>& a, const Row<DCH>& b) {
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        acc = fmaf(a.v[c].x, b.v[c].x, acc);
        acc = fmaf(a.v[c].y, b.v[c].y, acc);
        acc = fmaf(a.v[c].z, b.v[c].z, acc);
        acc = fmaf(a.v[c].w, b.v[c].w, acc);
    }
    return group16_sum(acc);
}
// y += g * x
template <int DCH>
__device__ __forceinline__ void row_axpy(Row<DCH>& y, float g, const Row<DCH>& x) {
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        y.v[c].x = fmaf(g, x.v[c].x, y.v[c].x);
        y.v[c].y = fmaf(g, x.v[c].y, y.v[c].y);
        y.v[c].z = fmaf(g, x.v[c].z, y.v[c].z);
        y.v[c].w = fmaf(g, x.v[c].w, y.v[c].w);
    }
}
template <int DCH>
__device__ __forceinline__ void row_zero(Row<DCH>& y) {
#pragma unroll
    for (int c = 0; c < DCH; c++) y.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float sgns_g(float f, float label, float alpha, const float* s_exp) {
    if (f > (float)MAX_EXP) return (label - 1.0f) * alpha;
    if (f < -(float)MAX_EXP) return (label - 0.0f) * alpha;
    int idx = (int)((f + (float)MAX_EXP) * (float)(EXP_TABLE_SIZE / MAX_EXP / 2));
    idx = min(max(idx, 0), EXP_TABLE_SIZE - 1);
    return (label - s_exp[idx]) * alpha;
}

__device__ __forceinline__ uint64_t shfl16_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl((int)lo, src, 16);
    hi = (uint32_t)__shfl((int)hi, src, 16);
    return ((uint64_t)hi << 32) | lo;
}

// one (target row, label 0) update against l1; sequential form used when a pair drew the same row twice
template <int DCH, int POL, bool BIG>
__device__ __forceinline__ void neg_update_serial(const Row<DCH>& l1, Row<DCH>& neu, const TableView& syn1neg, int32_t tg,
                                                  int lane, float alpha, const float* s_exp) {
    Row<DCH> r;
    row_load<DCH, Policy<POL>::LOAD_AUX, BIG>(r, syn1neg, tg, lane);
    float f = row_dot(l1, r);
    float g = sgns_g(f, 0.0f, alpha, s_exp);
    row_axpy(neu, g, r);
    row_axpy(r, g, l1);
    row_store<DCH, Policy<POL>::STORE_AUX, BIG>(r, syn1neg, tg, lane);
}

// Hierarchical softmax, Hogwild: every pair walks its centre's Huffman path from the root, so an inner node of subtree
// weight w takes a fraction w/total of ALL pairs' updates — the root all of them.  As memory-side atomics those
// serialise on a handful of rows (measured on cfg3: 12 ns per 64-B request, 38 s per step).  The hs_n_hot nodes
// nearest the root (the highest rows: weights ascend with the row index) therefore collect their updates in per-block
// LDS accumulators; the worker that makes an accumulator's hs_drain-th addition takes its content out (an exchange
// per element, so concurrent additions are never lost) and adds it to the row in memory.  Rows are still READ from
// memory: a block sees its own parked updates at most hs_drain additions late.
extern __shared__ float s_dyn[];
template <int DCH>
__device__ __forceinline__ void hot_add(float* s_hot, int* s_cnt, int slot, int drain, const TableView& t, int32_t row, int lane,
                                        float g, const Row<DCH>& x) {
    float* a = s_hot + slot * (DCH * 64) + lane;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        atomicAdd(a + c * 64 + 0, g * x.v[c].x);
        atomicAdd(a + c * 64 + 16, g * x.v[c].y);
        atomicAdd(a + c * 64 + 32, g * x.v[c].z);
        atomicAdd(a + c * 64 + 48, g * x.v[c].w);
    }
    int n = 0;
    if (lane == 0) n = atomicAdd(&s_cnt[slot], 1) + 1;
    n = __shfl(n, 0, 16);
    if (n % drain == 0) {
        float* gp = t.base + (size_t)row * (t.row_bytes / 4) + lane;
#pragma unroll
        for (int c = 0; c < DCH; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float v = atomicExch(a + c * 64 + m * 16, 0.f);
                if (v != 0.f) atomicAdd(gp + c * 64 + m * 16, v);
            }
    }
}

// end of the kernel, every thread of the block: what is still parked in LDS goes to memory (rows are contiguous)
__device__ __forceinline__ void hot_drain_block(const float* s_hot, int n_floats, float* first_row) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_floats; i += blockDim.x) {
        const float v = s_hot[i];
        if (v != 0.f) atomicAdd(first_row + i, v);
    }
}

__device__ __forceinline__ int32_t walk_tok(bool in_regs, const int32_t* sen, int idx, int32_t tk0, int32_t tk1, int32_t tk2, int32_t tk3) {
    if (!in_regs) return sen[idx];
    const int r = idx >> 4;
    const int32_t v = r == 0 ? tk0 : (r == 1 ? tk1 : (r == 2 ? tk2 : tk3));
    return __shfl(v, idx & 15, 16);
}

template <int DCH, int POL, bool BIG, bool HS, bool PART>
__global__ void __launch_bounds__(256, (DCH <= 2 && !BIG) ? (HS ? DGE_HS_WAVES : 4) : 1)
k_sgns_train(TrainParams p) {
    using P = Policy<POL>;
    __shared__ float s_exp[EXP_TABLE_SIZE];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    __syncthreads();

    const int lane = threadIdx.x & 15;
    const int64_t worker = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    constexpr bool HOT = HS && P::ATOMIC;         // inner nodes near the root combine their updates in LDS (hot_add)
    float* s_hot = HOT ? s_dyn : nullptr;
    int* s_hot_cnt = HOT ? (int*)(s_dyn + (size_t)p.hs_n_hot * DCH * 64) : nullptr;
    if (HOT) {
        for (int i = threadIdx.x; i < p.hs_n_hot * (DCH * 64 + 1); i += blockDim.x) s_dyn[i] = 0.f;   // +0.0f == int 0
        __syncthreads();
    }
    if (!HOT && worker >= p.n_workers) return;    // (the HOT kernel keeps every thread for its final block-wide drain)

    const TableView syn0 = make_view(p.syn0, p.V, p.stride, p.big_seg_shift);
    const TableView syn1neg = make_view(p.syn1neg, p.V, p.stride, p.big_seg_shift);
    const TableView syn1 = make_view(HS ? p.syn1 : p.syn1neg, p.V, p.stride, p.big_seg_shift);
    int64_t hs_o = 0; int hs_n = 0; uint64_t hs_bits = 0;   // Huffman path of the open centre

    // lane j turns the pair's LCG state s into the state after j+1 draws: s*mA + cA
    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= lane; j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }

    const int L = p.L, W = p.W, K = p.K;
    const bool toks_in_regs = L <= 64;
    unsigned long long my_pairs = 0, my_words = 0;
    // PART (block schedule): only centres in partition part_tgt and contexts in partition part_ctx are visited — found
    // through two bit masks over the walk's tokens, so a block costs what its own pairs cost.  Every pair draws from
    // its own stream (seeded from the centre's stream and the context position): the draws of a pair do not depend on
    // which other pairs of the centre this block trains.
    uint64_t ctx_mask = 0, tgt_mask = 0, pair_mask = 0, s_centre = 0;
    int nx_len = 0; int64_t nx_wb = 0; int32_t nx0 = -1, nx1 = -1, nx2 = -1, nx3 = -1;

    // ---- per-worker state: walk w, centre i, next context c (contexts are c..c_hi without i)
    int64_t w = (HOT && worker >= p.n_workers) ? p.n_rows - p.n_workers : worker - p.n_workers;   // surplus workers find no walk
    if (PART) walk_fetch(p, w + p.n_workers, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);
    int len = 0, i = 0, c = 1, c_hi = 0;
    int32_t tk0 = -1, tk1 = -1, tk2 = -1, tk3 = -1;       // the walk's tokens: lane j holds tokens j, j+16, j+32, j+48
    const int32_t* sen = p.sen;
    int32_t word = 0;
    float alpha = 0.f;
    uint64_t s = 0;
    int64_t gbase = 0;
    Row<DCH> h, dh;                                       // syn1neg[word] and (ATOMIC) its accumulated update
    bool h_dirty = false;

#define DGE_TOK(idx) walk_tok(toks_in_regs, sen, (idx), tk0, tk1, tk2, tk3)
    // close the open centre: publish what it accumulated on syn1neg[word]
#define DGE_CLOSE_CENTRE()                                                                                         \
    do {                                                                                                           \
        if (h_dirty) {                                                                                             \
            h_dirty = false;                                                                                       \
            if (P::ATOMIC) row_atomic_axpy(syn1neg, word, lane, 1.0f, dh);                                         \
            else row_store<DCH, P::STORE_AUX, BIG>(h, syn1neg, word, lane);                                           \
        }                                                                                                          \
    } while (0)

    for (;;) {
        // ------------------------------------------------------------------ advance to the next (centre, context) pair
        bool new_centre = false, alive = true;
        while (c > c_hi) {
            DGE_CLOSE_CENTRE();
            if (PART) i = first_bit_from(tgt_mask, i + 1, len); else i++;
            while (i >= len) {                             // next walk of this worker (empty walks are skipped)
                w += p.n_workers;
                if (w >= p.n_rows) { alive = false; break; }
                int64_t wb_next = 0;
                if (PART) {                                // prefetched while the previous walk was trained
                    len = nx_len; wb_next = nx_wb; tk0 = nx0; tk1 = nx1; tk2 = nx2; tk3 = nx3;
                    walk_fetch(p, w + p.n_workers, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);
                } else len = (int)p.len[w];
                i = 0;
                if (len > 0) {
                    if (!PART || p.part_ctx == p.part_tgt) my_words += (unsigned long long)len;      // (block schedule: once per batch, in episode 0)
                    sen = p.sen + w * L;
                    if (!PART && toks_in_regs) {
                        tk0 = lane < L ? sen[lane] : -1;
                        tk1 = lane + 16 < L ? sen[lane + 16] : -1;
                        tk2 = lane + 32 < L ? sen[lane + 32] : -1;
                        tk3 = lane + 48 < L ? sen[lane + 48] : -1;
                    }
                    if (PART) {
                        ctx_mask = part_token_mask(tk0, tk1, tk2, tk3, p.part_n, p.part_ctx);
                        tgt_mask = part_token_mask(tk0, tk1, tk2, tk3, p.part_n, p.part_tgt);
                        i = first_bit_from(tgt_mask, 0, len);
                    }
                    // learning rate from the exact number of in-vocabulary tokens that precede this walk
                    const int64_t wbw = PART ? wb_next : p.wb[w];
                    const int64_t done = p.words_done_base + (p.words_scale == 1.0 ? wbw : (int64_t)((double)wbw * p.words_scale));
                    alpha = (float)((double)p.alpha0 * (1.0 - (double)done / (double)(p.all_words + 1)));
                    if (alpha < p.min_alpha) alpha = p.min_alpha;
                    gbase = (p.gidx_base + w) * (int64_t)L;
                }
            }
            if (!alive) break;
            // open centre i: DL4J's window draw, radius W - b
            word = DGE_TOK(i);
            s = dge_mix64(p.seed + (uint64_t)(gbase + i));
            s = s * DGE_W2V_MULT + 11;
            const int radius = W - (int)(s % (uint64_t)W);
            c = max(0, i - radius);
            c_hi = min(len - 1, i + radius);
            if (c_hi == i) c_hi--;
            if (c == i) c++;
            new_centre = true;
            if (PART) {
                s_centre = s;
                pair_mask = ctx_mask & ~(1ull << i) & (c < 64 ? (~0ull << c) : 0ull);
                if (c_hi < 63) pair_mask &= (1ull << (c_hi + 1)) - 1ull;
                c = first_bit_from(pair_mask, 0, c_hi + 1);
            }
            if (HS) { hs_o = p.hs_off[word]; hs_n = (int)(p.hs_off[word + 1] - hs_o); hs_bits = p.hs_codes[word]; }
        }
        if (!alive) break;
        const int32_t last = DGE_TOK(c);
        if (PART) s = dge_mix64(s_centre + (uint64_t)c);

        // ------------------------------------------------------------------ one pair: l1 = syn0[last], target rows in syn1neg
        Row<DCH> l1, neu;
        row_load<DCH, P::LOAD_AUX, BIG>(l1, syn0, last, lane);
        if (new_centre) {
            row_load<DCH, P::LOAD_AUX, BIG>(h, syn1neg, word, lane);
            if (P::ATOMIC) row_zero(dh);
        }
        row_zero(neu);
        if (HS) {
            // word2vec.c "HIERARCHICAL SOFTMAX", ahead of the negatives: the inner nodes on the centre's Huffman path,
            // label 1 - code.  Outside (-6, 6) the step is skipped (not saturated, unlike the negative-sampling branch).
            // The rows of one path are distinct and l1 does not change within the pair, so a batch in flight is the
            // sequential result.
            for (int kd = 0; kd < hs_n; kd += 16) {
                const int kc = min(16, hs_n - kd);
                const int32_t t = lane < kc ? p.hs_points[hs_o + kd + lane] : -1;
                for (int base = 0; base < kc; base += NEG_BATCH) {
                    int32_t tg[NEG_BATCH];
                    Row<DCH> rr[NEG_BATCH];
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++) {
                        int32_t v = __shfl(t, (base + q) & 15, 16);
                        tg[q] = (base + q < kc) ? v : -1;
                    }
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++) row_load<DCH, P::LOAD_AUX, BIG>(rr[q], syn1, tg[q] >= 0 ? tg[q] : (BIG ? 0 : p.filler_row), lane);
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++)
                        if (tg[q] >= 0) {
                            const float f = row_dot(l1, rr[q]);
                            if (f > -(float)MAX_EXP && f < (float)MAX_EXP) {
                                const int idx = (int)((f + (float)MAX_EXP) * (float)(EXP_TABLE_SIZE / MAX_EXP / 2));
                                const float code = (float)((hs_bits >> (kd + base + q)) & 1ULL);
                                const float g = (1.0f - code - s_exp[idx]) * alpha;
                                row_axpy(neu, g, rr[q]);
                                if (P::ATOMIC) {
                                    if (tg[q] >= p.hs_hot0) hot_add<DCH>(s_hot, s_hot_cnt, tg[q] - p.hs_hot0, p.hs_drain, syn1, tg[q], lane, g, l1);
                                    else row_atomic_axpy(syn1, tg[q], lane, g, l1);
                                } else {
                                    row_axpy(rr[q], g, l1);
                                    row_store<DCH, P::STORE_AUX, BIG>(rr[q], syn1, tg[q], lane);
                                }
                            }
                        }
                }
            }
        }
        {   // d == 0: target = word, label 1 (word2vec order: positive first)
            float f = row_dot(l1, h);
            float g = sgns_g(f, 1.0f, alpha, s_exp);
            row_axpy(neu, g, h);
            row_axpy(h, g, l1);
            if (P::ATOMIC) row_axpy(dh, g, l1);
            h_dirty = true;
        }
        for (int kd = 0; kd < K; kd += 16) {
            const int kc = min(16, K - kd);
            // lane j draws negative kd+j
            const uint64_t sl = s * mA + cA;
            int32_t t = -1;
            if (lane < kc) {
                t = p.table[(sl >> 16) % (uint64_t)p.T];
                if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                if (PART) t = part_row(t, p.part_n, p.part_tgt, p.V);
                if (t == word) t = -1;
            }
            s = shfl16_u64(sl, kc - 1);
            for (int base = 0; base < kc; base += NEG_BATCH) {
                int32_t tg[NEG_BATCH];
#pragma unroll
                for (int q = 0; q < NEG_BATCH; q++) {
                    int32_t v = __shfl(t, (base + q) & 15, 16);
                    tg[q] = (base + q < kc) ? v : -1;
                }
                bool dup = false;
                if (!P::ATOMIC) {
#pragma unroll
                    for (int q = 1; q < NEG_BATCH; q++)
#pragma unroll
                        for (int r = 0; r < q; r++) dup |= (tg[q] >= 0 && tg[q] == tg[r]);
                }
                if (!dup) {
                    // all rows of the batch in flight together: loads are unconditional (a skipped slot reads the
                    // centre's own row, always valid), only the arithmetic and the store are guarded
                    Row<DCH> rr[NEG_BATCH];
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++) row_load<DCH, P::LOAD_AUX, BIG>(rr[q], syn1neg, tg[q] >= 0 ? tg[q] : (BIG ? word : p.filler_row), lane);
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++)
                        if (tg[q] >= 0) {
                            float f = row_dot(l1, rr[q]);
                            float g = sgns_g(f, 0.0f, alpha, s_exp);
                            row_axpy(neu, g, rr[q]);
                            if (P::ATOMIC) {
                                row_atomic_axpy(syn1neg, tg[q], lane, g, l1);
                            } else {
                                row_axpy(rr[q], g, l1);
                                row_store<DCH, P::STORE_AUX, BIG>(rr[q], syn1neg, tg[q], lane);
                            }
                        }
                } else {
#pragma unroll 1
                    for (int q = 0; q < NEG_BATCH; q++)
                        if (tg[q] >= 0) neg_update_serial<DCH, POL, BIG>(l1, neu, syn1neg, tg[q], lane, alpha, s_exp);
                }
            }
        }
        if (P::ATOMIC) {
            row_atomic_axpy(syn0, last, lane, 1.0f, neu);
        } else {
#pragma unroll
            for (int q = 0; q < DCH; q++) {
                l1.v[q].x += neu.v[q].x; l1.v[q].y += neu.v[q].y; l1.v[q].z += neu.v[q].z; l1.v[q].w += neu.v[q].w;
            }
            row_store<DCH, P::STORE_AUX, BIG>(l1, syn0, last, lane);
        }
        my_pairs++;
        if (PART) {
            pair_mask &= pair_mask - 1ull;
            c = first_bit_from(pair_mask, 0, c_hi + 1);
        } else {
            c++;
            if (c == i) c++;
        }
    }
    DGE_CLOSE_CENTRE();
#undef DGE_TOK
#undef DGE_CLOSE_CENTRE
    if (lane == 0) {
        if (my_pairs) atomicAdd(&p.counters[0], my_pairs);
        if (my_words) atomicAdd(&p.counters[1], my_words);
    }
    if (HOT) hot_drain_block(s_hot, p.hs_n_hot * DCH * 64, p.syn1 + (size_t)p.hs_hot0 * (DCH * 64));
}

// ------------------------------------------------------------------------------------------ all-locked Hogwild trainer
// Policies 5/6: every row update of BOTH tables is a read-modify-write under that row's commit lock, rows move as 16 bytes
// per lane (lane j owns elements 64c+4j..64c+4j+3: one dwordx4 per chunk, a whole 256-B chunk per group and
// instruction).  Measured on cfg3 (ablations in DESIGN.md §5.1): the write-through stores of the 4-byte-per-lane layout
// that the float atomics need cost more than everything else in the pair; with 16-byte stores and 2 small atomic
// requests per row (take / drop the lock) the pair is bounded by its HBM traffic again.
// Lock order: the pair's syn0 row first (together with the first chunk of syn1neg try-locks; if it is not won,
// everything won in that round is dropped again and the round is repeated), then syn1neg rows in try-lock rounds that
// never wait while holding a syn1neg lock: no hold-and-wait cycle exists.
// STRICT (policy 6): a row is committed with one returning atomic per 128-B line before its lock drops — no update is ever
// lost (dge_selftest_locked_rows).  Relaxed (policy 5): the wave only drains its own stores (vmcnt) before dropping the
// lock; a re-lock from another XCD can overtake the write-through, which loses a row update with measured probability
// <= 4e-7 at >= 65k rows (0 of 2.4e6 at 1M rows) and up to 1.5 % of the worst row's updates on a 1024-row hot set
// hammered by 12k workers — Hogwild noise, below what unsynchronised float read-modify-writes lose (policy 1).
// Commit of a row before its lock drops: after the row's write-through stores, ONE returning float atomic (+0.0f) per
// 128-B line of the row.  A line's store and the atomic that follows it travel the same channel in order and the atomic
// is performed at the memory side, so its return implies the line's data is there; the wave then waits for the returns
// (row_commit_wait) and only then clears the lock word.  Lane l probes line l of the row.
__device__ __forceinline__ float row_probe_lines(const TableView& t, int32_t row, int lane, int n_lines) {
    float old = 0.f;
    if (lane < n_lines) old = __hip_atomic_fetch_add(t.base + (size_t)row * (t.row_bytes / 4) + lane * 32, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return old;
}
template <int DCH, int AUX, bool BIG>
__device__ __forceinline__ void rowA_load(Row<DCH>& r, const TableView& t, int32_t row, int lane) {
    if (BIG) {
        uint32_t ro;
        const __amdgpu_buffer_rsrc_t rs = row_view(t, row, ro);
#pragma unroll
        for (int c = 0; c < DCH; c++) {
            const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + (uint32_t)lane * 16u + c * 256u), 0, AUX));
            r.v[c] = make_float4(f.x, f.y, f.z, f.w);
        }
        return;
    }
    const uint32_t off = (uint32_t)row * t.row_bytes + (uint32_t)lane * 16u;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        // NOTE (hipcc 7.2): bit-casting the ELEMENTS of the loaded <4 x i32> lets the optimiser narrow the load to one
        // dword (wrong data in y/z/w); casting the whole vector keeps the dwordx4.
        const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(t.rsrc, (int)(off + c * 256u), 0, AUX));
        r.v[c] = make_float4(f.x, f.y, f.z, f.w);
    }
}
template <int DCH, int AUX, bool BIG>
__device__ __forceinline__ void rowA_store(const Row<DCH>& r, const TableView& t, int32_t row, int lane) {
    if (BIG) {
        uint32_t ro;
        const __amdgpu_buffer_rsrc_t rs = row_view(t, row, ro);
#pragma unroll
        for (int c = 0; c < DCH; c++) {
            v4f f;
            f.x = r.v[c].x; f.y = r.v[c].y; f.z = r.v[c].z; f.w = r.v[c].w;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, f), rs, (int)(ro + (uint32_t)lane * 16u + c * 256u), 0, AUX);
        }
        return;
    }
    const uint32_t off = (uint32_t)row * t.row_bytes + (uint32_t)lane * 16u;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        v4f f;
        f.x = r.v[c].x; f.y = r.v[c].y; f.z = r.v[c].z; f.w = r.v[c].w;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, f), t.rsrc, (int)(off + c * 256u), 0, AUX);
    }
}

// Policy 7 (HOTMIX): rows the vocabulary's head are wanted by many workers at once — under row locks they make the
// kernel spin (cfg5: 5e5 edges/s).  Those rows are never locked: they are read with agent-scope loads and updated with
// memory-side float atomics, like policy 2; all other rows keep the lock protocol.  A row is always updated one way or
// the other, never both, so neither side can overwrite the other's update.
// The atomics want 64 contiguous bytes per group instruction (lane j' -> element 64c + 16m + j'), the registers hold the
// 16-byte layout (lane j -> elements 64c + 4j .. 4j+3): element 16m + j' sits in lane 4m + j'/4, component j' % 4.
template <int DCH>
__device__ __forceinline__ void rowA_atomic_axpy(const TableView& t, int32_t row, int lane, float g, const Row<DCH>& x) {
    float* p = t.base + (size_t)row * (t.row_bytes / 4) + lane;
    const int hi = lane >> 2, comp = lane & 3;
#pragma unroll
    for (int c = 0; c < DCH; c++)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int src = 4 * m + hi;
            const float x0 = __shfl(x.v[c].x, src, 16), x1 = __shfl(x.v[c].y, src, 16), x2 = __shfl(x.v[c].z, src, 16), x3 = __shfl(x.v[c].w, src, 16);
            const float v = comp == 0 ? x0 : (comp == 1 ? x1 : (comp == 2 ? x2 : x3));
            atomicAdd(p + c * 64 + 16 * m, g * v);
        }
}
// the centre's delta parked in LDS (index 64q + 16*component + lane holds element 64q + 4*lane + component)
template <int DCH>
__device__ __forceinline__ void ldsA_atomic_add(const TableView& t, int32_t row, int lane, const float* d_base) {
    float* p = t.base + (size_t)row * (t.row_bytes / 4) + lane;
    const int hi = lane >> 2, comp = lane & 3;
#pragma unroll
    for (int c = 0; c < DCH; c++)
#pragma unroll
        for (int m = 0; m < 4; m++) atomicAdd(p + c * 64 + 16 * m, d_base[c * 64 + comp * 16 + 4 * m + hi]);
}

template <int DCH, bool STRICT, bool BIG, bool HOTMIX = false>
__device__ __forceinline__ void flushA_blocking(const TableView& syn1neg, int* locks, int32_t row, const float* d, int lane, int32_t hot_rows = 0) {
    if (HOTMIX && row < hot_rows) { ldsA_atomic_add<DCH>(syn1neg, row, lane, d - lane); return; }
    for (;;) {
        const bool won = lane == 0 ? row_trylock(locks, row) : false;
        const bool got = __shfl((int)won, 0, 16) != 0;
        if (got) {
            Row<DCH> cur;
            rowA_load<DCH, 16, BIG>(cur, syn1neg, got ? row : 0, lane);
#pragma unroll
            for (int q = 0; q < DCH; q++) {
                cur.v[q].x += d[q * 64]; cur.v[q].y += d[q * 64 + 16]; cur.v[q].z += d[q * 64 + 32]; cur.v[q].w += d[q * 64 + 48];
            }
            rowA_store<DCH, 16, BIG>(cur, syn1neg, row, lane);
            row_commit_wait(STRICT ? row_probe_lines(syn1neg, row, lane, DCH * 2) : 0.f);
            if (won) row_unlock<STRICT>(locks, row);
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

#define LK_NEG_LANES 13      /* lanes 0..12 draw negatives, lane 13 = pending centre flush, lane 14 = the pair's syn0 row */
#ifndef LK_CHUNK
#define LK_CHUNK 10          /* negatives per lock round: a multiple of NEG_BATCH, so no batch of a full chunk loads filler rows */
#endif
// 3 waves per SIMD is the measured optimum for D <= 128: 4 (128 VGPRs) spills 88 B per lane and runs 20 % slower, 2 runs 12 % slower
template <int DCH, bool STRICT, bool BIG, bool HOTMIX, bool PART>
__global__ void __launch_bounds__(256, (DCH <= 2 && !BIG) ? (HOTMIX ? (PART ? 2 : DGE_HOTMIX_WAVES) : (DCH == 1 ? 4 : DGE_LOCKED_WAVES)) : ((HOTMIX && DCH <= 4) ? 2 : 1))
k_sgns_train_locked(TrainParams p) {
    __shared__ float s_exp[EXP_TABLE_SIZE];
    __shared__ float s_dh[16 * 2 * DCH * 64];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    __syncthreads();

    const int lane = threadIdx.x & 15;
    const int wk = threadIdx.x >> 4;
    const int64_t worker = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (worker >= p.n_workers) return;

    const TableView syn0 = make_view(p.syn0, p.V, p.stride, p.big_seg_shift);
    const TableView syn1neg = make_view(p.syn1neg, p.V, p.stride, p.big_seg_shift);
    int* const locks1 = p.locks;
    int* const locks0 = p.locks + p.V + 1;
    const int32_t hot_rows = HOTMIX ? p.hot_rows : 0;

    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= lane; j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }

    const int L = p.L, W = p.W, K = p.K;
    const bool toks_in_regs = L <= 64;
    unsigned long long my_pairs = 0, my_words = 0;

    int64_t w = worker - p.n_workers;
    int len = 0, i = 0, c = 1, c_hi = 0;
    int32_t tk0 = -1, tk1 = -1, tk2 = -1, tk3 = -1;
    const int32_t* sen = p.sen;
    int32_t word = 0;
    float alpha = 0.f;
    uint64_t s = 0;
    int64_t gbase = 0;
    Row<DCH> h;
    bool h_dirty = false;
    int32_t pend_row = -1;
    int cur_buf = 0;
    float* const my_dh = s_dh + (size_t)wk * 2 * DCH * 64;
    bool retry_pair = false;      // the pair's syn0 row was busy: same pair again on the next trip through the loop
    int32_t t_first = -1;         // this lane's slot of the pair's first chunk (kept across a retry: the draw is not repeated)
    int32_t last = 0;
    uint64_t ctx_mask = 0, tgt_mask = 0, pair_mask = 0, s_centre = 0;     // PART: see k_sgns_train
    int nx_len = 0; int64_t nx_wb = 0; int32_t nx0 = -1, nx1 = -1, nx2 = -1, nx3 = -1;
    if (PART) walk_fetch(p, w + p.n_workers, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);

#define LK_TOK(idx) walk_tok(toks_in_regs, sen, (idx), tk0, tk1, tk2, tk3)
    // positive target (label 1): the centre's row lives in registers for all its contexts, its accumulated delta in LDS
#define LK_POSITIVE()                                                                                                  \
    do {                                                                                                               \
        const float f_ = row_dot(l1, h);                                                                               \
        const float g_ = sgns_g(f_, 1.0f, alpha, s_exp);                                                               \
        row_axpy(neu, g_, h);                                                                                          \
        row_axpy(h, g_, l1);                                                                                           \
        float* d_ = my_dh + cur_buf * DCH * 64 + lane;                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < DCH; q_++) {                                                           \
            d_[q_ * 64] = fmaf(g_, l1.v[q_].x, d_[q_ * 64]); d_[q_ * 64 + 16] = fmaf(g_, l1.v[q_].y, d_[q_ * 64 + 16]); \
            d_[q_ * 64 + 32] = fmaf(g_, l1.v[q_].z, d_[q_ * 64 + 32]); d_[q_ * 64 + 48] = fmaf(g_, l1.v[q_].w, d_[q_ * 64 + 48]); \
        }                                                                                                              \
        h_dirty = true;                                                                                                \
    } while (0)
#define LK_CLOSE_CENTRE()                                                                                              \
    do {                                                                                                               \
        if (h_dirty) {                                                                                                 \
            h_dirty = false;                                                                                           \
            if (pend_row >= 0) flushA_blocking<DCH, STRICT, BIG, HOTMIX>(syn1neg, locks1, pend_row, my_dh + (cur_buf ^ 1) * DCH * 64 + lane, lane, hot_rows); \
            pend_row = word;                                                                                           \
            cur_buf ^= 1;                                                                                              \
        }                                                                                                              \
    } while (0)

    for (;;) {
        bool new_centre = false, alive = true;
        while (!retry_pair && c > c_hi) {
            LK_CLOSE_CENTRE();
            if (PART) i = first_bit_from(tgt_mask, i + 1, len); else i++;
            while (i >= len) {
                w += p.n_workers;
                if (w >= p.n_rows) { alive = false; break; }
                int64_t wb_next = 0;
                if (PART) {                                // prefetched while the previous walk was trained
                    len = nx_len; wb_next = nx_wb; tk0 = nx0; tk1 = nx1; tk2 = nx2; tk3 = nx3;
                    walk_fetch(p, w + p.n_workers, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);
                } else len = (int)p.len[w];
                i = 0;
                if (len > 0) {
                    if (!PART || p.part_ctx == p.part_tgt) my_words += (unsigned long long)len;      // (block schedule: once per batch, in episode 0)
                    sen = p.sen + w * L;
                    if (!PART && toks_in_regs) {
                        tk0 = lane < L ? sen[lane] : -1;
                        tk1 = lane + 16 < L ? sen[lane + 16] : -1;
                        tk2 = lane + 32 < L ? sen[lane + 32] : -1;
                        tk3 = lane + 48 < L ? sen[lane + 48] : -1;
                    }
                    if (PART) {
                        ctx_mask = part_token_mask(tk0, tk1, tk2, tk3, p.part_n, p.part_ctx);
                        tgt_mask = part_token_mask(tk0, tk1, tk2, tk3, p.part_n, p.part_tgt);
                        i = first_bit_from(tgt_mask, 0, len);
                    }
                    const int64_t wbw = PART ? wb_next : p.wb[w];
                    const int64_t done = p.words_done_base + (p.words_scale == 1.0 ? wbw : (int64_t)((double)wbw * p.words_scale));
                    alpha = (float)((double)p.alpha0 * (1.0 - (double)done / (double)(p.all_words + 1)));
                    if (alpha < p.min_alpha) alpha = p.min_alpha;
                    gbase = (p.gidx_base + w) * (int64_t)L;
                }
            }
            if (!alive) break;
            word = LK_TOK(i);
            s = dge_mix64(p.seed + (uint64_t)(gbase + i));
            s = s * DGE_W2V_MULT + 11;
            const int radius = W - (int)(s % (uint64_t)W);
            c = max(0, i - radius);
            c_hi = min(len - 1, i + radius);
            if (c_hi == i) c_hi--;
            if (c == i) c++;
            new_centre = true;
            if (PART) {
                s_centre = s;
                pair_mask = ctx_mask & ~(1ull << i) & (c < 64 ? (~0ull << c) : 0ull);
                if (c_hi < 63) pair_mask &= (1ull << (c_hi + 1)) - 1ull;
                c = first_bit_from(pair_mask, 0, c_hi + 1);
            }
        }
        if (!alive) break;
        if (!retry_pair) {
            last = LK_TOK(c);
            if (PART) s = dge_mix64(s_centre + (uint64_t)c);
        }

        Row<DCH> l1, neu;
        if (new_centre) {
            // the previous centre's delta is still parked in LDS; when it belongs to THIS row (the same token twice in a walk)
            // it goes out first, so that one worker alone reads exactly what the sequential loop would
            if (pend_row == word) {
                flushA_blocking<DCH, STRICT, BIG, HOTMIX>(syn1neg, locks1, pend_row, my_dh + (cur_buf ^ 1) * DCH * 64 + lane, lane, hot_rows);
                pend_row = -1;
            }
            rowA_load<DCH, 16, BIG>(h, syn1neg, word, lane);       // unlocked read: stale by at most the OTHER workers' updates in flight
            float* d = my_dh + cur_buf * DCH * 64 + lane;
#pragma unroll
            for (int q = 0; q < DCH; q++) { d[q * 64] = 0.f; d[q * 64 + 16] = 0.f; d[q * 64 + 32] = 0.f; d[q * 64 + 48] = 0.f; }
        }
        row_zero(neu);
        bool have_l1 = false, abort_pair = false;
        const bool l1_only = retry_pair;
        int kd = 0;
        do {    // chunks of up to 13 negatives (at least one pass so that the syn0 row is locked and loaded even when K == 0)
            const int kc = min(LK_CHUNK, K - kd);
            int32_t t = -1;
            if (retry_pair && kd == 0) {
                t = t_first;
            } else {
                const uint64_t sl = s * mA + cA;
                if (lane < kc) {
                    t = p.table[(sl >> 16) % (uint64_t)p.T];
                    if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                    if (PART) t = part_row(t, p.part_n, p.part_tgt, p.V);
                    if (t == word) t = -1;
                }
                if (kc > 0) s = shfl16_u64(sl, kc - 1);
                if (kd == 0) t_first = t;
            }
            if (lane == 13) t = pend_row;
            if (lane == 14) t = last;
            // One lock round per CHUNK: every still-untrained row of the chunk (and the pending centre flush, and the pair's
            // syn0 row) is asked for at once; the rows that were won are then trained NEG_BATCH at a time — loads of a batch
            // in flight together, no wait between batches — and one commit wait ends the round before the locks drop.  With
            // K <= NEG_BATCH this is one batch per round; with K = 20 it is two lock/commit round trips per pair instead of five.
            unsigned pend13 = (unsigned)(__ballot(lane < kc && t >= 0) >> (threadIdx.x & 48)) & 0x1FFFu;
            bool flush_pending = pend_row >= 0;
            while (pend13 || !have_l1) {
                // a pair that already lost the race for its syn0 row once asks for that row ALONE until it has it:
                // otherwise the many waiting workers of a hot row keep grabbing (and dropping) the syn1neg rows the
                // row's current holder needs, and the holder starves (seen as a hang on a 3-row vocabulary)
                const bool others_ok = have_l1 || !l1_only;
                // (a negative that drew the row whose flush is still pending lets the flush go first: word2vec order)
                const bool want = (others_ok && lane < kc && ((pend13 >> lane) & 1u) && !(flush_pending && t == pend_row)) ||
                                  (others_ok && lane == 13 && flush_pending) || (lane == 14 && !have_l1);
                const bool lockfree = HOTMIX && want && (t < hot_rows || (lane == 14 && p.syn0_free));     // a head row: no lock, atomics
                const bool won = (want && !lockfree) ? row_trylock(lane == 14 ? locks0 : locks1, t) : false;
                const unsigned long long bal = __ballot(won || lockfree);
                const unsigned gotl = (unsigned)(bal >> (threadIdx.x & 48)) & 0xFFFFu;
                if (!have_l1 && !((gotl >> 14) & 1u)) {
                    // the pair's syn0 row is busy (possibly held by another group of THIS wave, which can only drop it
                    // once this group stops looping): drop whatever this round won and leave the pair for the next
                    // trip through the outer loop — no waiting while holding, no spinning under divergence
                    if (won) row_unlock<STRICT>(lane == 14 ? locks0 : locks1, t);
                    abort_pair = true;
                    break;
                }
                const bool got_l1 = !have_l1;
                const unsigned got13 = gotl & 0x1FFFu & pend13;
                const bool gotf = flush_pending && ((gotl >> 13) & 1u);
                Row<DCH> fr;
                float my_hot_g = 0.f;                      // HOTMIX: lane j keeps the step of the chunk's j-th row when that is a head row
                if (got_l1) rowA_load<DCH, 16, BIG>(l1, syn0, (gotl >> 14) & 1u ? last : 0, lane);
                if (flush_pending) rowA_load<DCH, 16, BIG>(fr, syn1neg, gotf ? pend_row : (BIG ? word : p.filler_row), lane);
                have_l1 = true;
                bool do_pos = got_l1;                      // the positive target comes first (word2vec order), once l1 is here
                float acc = 0.f;                           // STRICT: the commit probes' returns
                for (int base = 0; base < kc; base += NEG_BATCH) {
                    const unsigned got = (got13 >> base) & ((1u << NEG_BATCH) - 1u);
                    if (!got) continue;
                    int32_t tg[NEG_BATCH];
                    Row<DCH> rr[NEG_BATCH];
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++) tg[q] = __shfl(t, (base + q) & 15, 16);
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++) rowA_load<DCH, 16, BIG>(rr[q], syn1neg, ((got >> q) & 1u) ? tg[q] : (BIG ? word : p.filler_row), lane);
                    if (do_pos) { do_pos = false; LK_POSITIVE(); }     // behind the batch's loads: they are in flight meanwhile
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++)
                        if ((got >> q) & 1u) {
                            float f = row_dot(l1, rr[q]);
                            float g = sgns_g(f, 0.0f, alpha, s_exp);
                            row_axpy(neu, g, rr[q]);
                            if (HOTMIX && tg[q] < hot_rows) {
                                if (lane == base + q) my_hot_g = g;   // the atomics go out after this round's locks have dropped (below)
                            } else {
                                row_axpy(rr[q], g, l1);
                                rowA_store<DCH, 16, BIG>(rr[q], syn1neg, tg[q], lane);
                            }
                        }
                    if (STRICT) {   // every stored row is committed line by line (lane = 4*slot + line for DCH 2) before the locks drop
                        const int n_lines = DCH * 2;
#pragma unroll
                        for (int rep = 0; rep < (NEG_BATCH * DCH * 2 + 15) / 16; rep++) {
                            const int idx = lane + rep * 16, q = idx / n_lines, ln = idx - q * n_lines;
                            int32_t row = -1;
#pragma unroll
                            for (int qq = 0; qq < NEG_BATCH; qq++) if (qq == q && ((got >> qq) & 1u)) row = tg[qq];
                            if (row >= 0) acc += __hip_atomic_fetch_add(syn1neg.base + (size_t)row * (syn1neg.row_bytes / 4) + ln * 32, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                if (do_pos) LK_POSITIVE();                 // (no row of the chunk was won in this round, or K == 0)
                const bool hot_flush = HOTMIX && gotf && pend_row < hot_rows;
                if (gotf && !hot_flush) {
                    const float* d = my_dh + (cur_buf ^ 1) * DCH * 64 + lane;
#pragma unroll
                    for (int q = 0; q < DCH; q++) {
                        fr.v[q].x += d[q * 64]; fr.v[q].y += d[q * 64 + 16]; fr.v[q].z += d[q * 64 + 32]; fr.v[q].w += d[q * 64 + 48];
                    }
                    rowA_store<DCH, 16, BIG>(fr, syn1neg, pend_row, lane);
                    if (STRICT) acc += row_probe_lines(syn1neg, pend_row, lane, DCH * 2);
                }
                row_commit_wait(acc);
                if (won && lane != 14) row_unlock<STRICT>(locks1, t);
                if (HOTMIX) {
                    // head rows: memory-side atomics, issued behind the commit so that the wait above (which drains every
                    // outstanding memory operation of the wave) never sits on them while row locks are held
                    for (int j = 0; j < kc; j++) {
                        const int32_t tj = __shfl(t, j, 16);
                        const float gj = __shfl(my_hot_g, j, 16);
                        if (((got13 >> j) & 1u) && tj < hot_rows) rowA_atomic_axpy<DCH>(syn1neg, tj, lane, gj, l1);
                    }
                    if (hot_flush) ldsA_atomic_add<DCH>(syn1neg, pend_row, lane, my_dh + (cur_buf ^ 1) * DCH * 64);
                }
                pend13 &= ~got13;
                if (gotf) { flush_pending = false; pend_row = -1; if (lane == 13) t = -1; }
                if (pend13) __builtin_amdgcn_s_sleep(2);
            }
            kd += LK_CHUNK;
        } while (kd < K && !abort_pair);
        if (abort_pair) { retry_pair = true; __builtin_amdgcn_s_sleep(8); continue; }
        retry_pair = false;

#pragma unroll
        for (int q = 0; q < DCH; q++) {
            l1.v[q].x += neu.v[q].x; l1.v[q].y += neu.v[q].y; l1.v[q].z += neu.v[q].z; l1.v[q].w += neu.v[q].w;
        }
        if (HOTMIX && (last < hot_rows || p.syn0_free)) {
            rowA_atomic_axpy<DCH>(syn0, last, lane, 1.0f, neu);
        } else {
            rowA_store<DCH, 16, BIG>(l1, syn0, last, lane);
            row_commit_wait(STRICT ? row_probe_lines(syn0, last, lane, DCH * 2) : 0.f);
            if (lane == 14) row_unlock<STRICT>(locks0, last);
        }
        my_pairs++;
        if (PART) {
            pair_mask &= pair_mask - 1ull;
            c = first_bit_from(pair_mask, 0, c_hi + 1);
        } else {
            c++;
            if (c == i) c++;
        }
    }
    LK_CLOSE_CENTRE();
    if (pend_row >= 0) flushA_blocking<DCH, STRICT, BIG, HOTMIX>(syn1neg, locks1, pend_row, my_dh + (cur_buf ^ 1) * DCH * 64 + lane, lane, hot_rows);
#undef LK_TOK
#undef LK_POSITIVE
#undef LK_CLOSE_CENTRE
    if (lane == 0) {
        if (my_pairs) atomicAdd(&p.counters[0], my_pairs);
        if (my_words) atomicAdd(&p.counters[1], my_words);
    }
}

// ------------------------------------------------------------------------------------------ lock protocol self-test
// Conservation check of the commit-lock protocol used by k_sgns_train_locked, with the same primitives
// (row_trylock / rowA_load sc1 / rowA_store sc1 / workgroup release fence / row_unlock): every worker repeatedly picks
// NEG_BATCH pseudo-random rows, wins their locks in try-lock rounds and adds 1.0 to every element of each row it won.
// If exclusion, read freshness or write visibility failed anywhere on the chip, some increment would be lost:
// at the end every element of row r must equal the exact number of increments of row r (counted with integer atomics).
template <int DCH, int LAUX, int SAUX, int FENCE>
__global__ void __launch_bounds__(256)
k_selftest_locked_rows(float* table, int* locks, unsigned long long* counts, int32_t n_rows, int stride, int64_t n_workers,
                       int iters, uint64_t seed) {
    const int lane = threadIdx.x & 15;
    const int64_t worker = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (worker >= n_workers) return;
    const TableView tv = make_view(table, n_rows, stride);
    for (int it = 0; it < iters; it++) {
        int32_t t = -1;
        if (lane < NEG_BATCH) t = (int32_t)(dge_mix64(seed + (uint64_t)((worker * iters + it) * 16 + lane)) % (uint64_t)n_rows);
        int32_t tg[NEG_BATCH];
#pragma unroll
        for (int q = 0; q < NEG_BATCH; q++) tg[q] = __shfl(t, q, 16);
        unsigned pending = (1u << NEG_BATCH) - 1u;
        while (pending) {
            const bool want = lane < NEG_BATCH && ((pending >> lane) & 1u);
            const bool won = want ? row_trylock(locks, t) : false;
            const unsigned long long bal = __ballot(won);
            const unsigned got = (unsigned)(bal >> (threadIdx.x & 48)) & ((1u << NEG_BATCH) - 1u) & pending;
            if (FENCE & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            Row<DCH> rr[NEG_BATCH];
#pragma unroll
            for (int q = 0; q < NEG_BATCH; q++) rowA_load<DCH, LAUX, false>(rr[q], tv, ((got >> q) & 1u) ? tg[q] : 0, lane);
#pragma unroll
            for (int q = 0; q < NEG_BATCH; q++)
                if ((got >> q) & 1u) {
#pragma unroll
                    for (int c = 0; c < DCH; c++) { rr[q].v[c].x += 1.f; rr[q].v[c].y += 1.f; rr[q].v[c].z += 1.f; rr[q].v[c].w += 1.f; }
                    rowA_store<DCH, SAUX, false>(rr[q], tv, tg[q], lane);
                }
            if (FENCE & 4) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < NEG_BATCH; q++) if ((got >> q) & 1u) acc += row_probe_lines(tv, tg[q], lane, stride / 32);
                asm volatile("" :: "v"(acc));
            }
            if (FENCE & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (won) { row_unlock<(FENCE & 4) != 0>(locks, t); atomicAdd(&counts[t], 1ULL); }
            pending &= ~got;
            if (pending) __builtin_amdgcn_s_sleep(2);
        }
    }
}

extern "C" int dge_selftest_locked_rows(int device, int32_t n_rows, int64_t n_workers, int32_t iters, uint64_t seed, int32_t commit,
                                        int64_t* total_increments, double* max_abs_error) {
    if (n_rows <= 0 || n_workers <= 0 || iters <= 0 || !total_increments || !max_abs_error) DGE_FAIL(DGE_ERR_ARG, "dge_selftest_locked_rows: bad argument");
    int rc = dge_require_device(device);
    if (rc) return rc;
    const int stride = 128;
    float* d_tab = nullptr; int* d_locks = nullptr; unsigned long long* d_cnt = nullptr;
    if ((rc = dge_dev_alloc(&d_tab, (size_t)n_rows * stride))) return rc;
    if ((rc = dge_dev_alloc(&d_locks, (size_t)n_rows))) return rc;
    if ((rc = dge_dev_alloc(&d_cnt, (size_t)n_rows))) return rc;
    DGE_HIP(hipMemset(d_tab, 0, (size_t)n_rows * stride * sizeof(float)));
    DGE_HIP(hipMemset(d_locks, 0, (size_t)n_rows * sizeof(int)));
    DGE_HIP(hipMemset(d_cnt, 0, (size_t)n_rows * sizeof(unsigned long long)));
    unsigned blocks = (unsigned)((n_workers * 16 + 255) / 256);
#define ST_LAUNCH(L, S, F) hipLaunchKernelGGL((k_selftest_locked_rows<2, L, S, F>), dim3(blocks), dim3(256), 0, 0, d_tab, d_locks, d_cnt, n_rows, stride, n_workers, iters, seed)
    switch (commit) {
        case 0: ST_LAUNCH(16, 16, 0); break;      // relaxed commit of policy 5: sc1 both sides, the wave drains its stores
        case 1: ST_LAUNCH(16, 16, 4); break;      // strict commit of policy 6: + one returning atomic per stored 128-B line
        case 2: ST_LAUNCH(16, 16, 2); break;      // agent-scope release fence (buffer_wbl2): also lossless, 19x slower in the trainer
        default: DGE_FAIL(DGE_ERR_ARG, "dge_selftest_locked_rows: commit must be 0, 1 or 2");
    }
#undef ST_LAUNCH
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipDeviceSynchronize());
    std::vector<float> tab((size_t)n_rows * stride); std::vector<unsigned long long> cnt((size_t)n_rows); std::vector<int> lk((size_t)n_rows);
    DGE_HIP(hipMemcpy(tab.data(), d_tab, tab.size() * sizeof(float), hipMemcpyDeviceToHost));
    DGE_HIP(hipMemcpy(cnt.data(), d_cnt, cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    DGE_HIP(hipMemcpy(lk.data(), d_locks, lk.size() * sizeof(int), hipMemcpyDeviceToHost));
    dge_dev_free(d_tab); dge_dev_free(d_locks); dge_dev_free(d_cnt);
    double worst = 0.0; int64_t total = 0;
    for (int32_t r = 0; r < n_rows; r++) {
        total += (int64_t)cnt[(size_t)r];
        if (lk[(size_t)r] != 0) worst = 1e30;                       // a lock was left held
        for (int c = 0; c < stride; c++) worst = std::max(worst, fabs((double)tab[(size_t)r * stride + c] - (double)cnt[(size_t)r]));
    }
    *total_increments = total; *max_abs_error = worst;
    return DGE_OK;
}

// hot_add / hot_drain_block in isolation: every worker adds 1.0 to every element of pseudo-random hot rows `iters` times;
// afterwards each row must hold exactly the number of additions it received (integers < 2^24 are exact in float).
__global__ void __launch_bounds__(256) k_selftest_hot_add(float* rows, unsigned long long* hits, int n_hot, int drain, int iters, uint64_t seed, int64_t n_workers) {
    const int lane = threadIdx.x & 15;
    const int64_t worker = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    float* s_hot = s_dyn;
    int* s_cnt = (int*)(s_dyn + (size_t)n_hot * 64);
    for (int i = threadIdx.x; i < n_hot * 65; i += blockDim.x) s_dyn[i] = 0.f;
    __syncthreads();
    const TableView t = make_view(rows, n_hot, 64);
    Row<1> one; one.v[0] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (worker < n_workers) {
        uint64_t s = dge_mix64(seed + (uint64_t)worker);
        for (int it = 0; it < iters; it++) {
            s = s * DGE_W2V_MULT + 11;
            // skewed like a Huffman path: slot k with probability ~2^-(k+1)
            int slot = min(n_hot - 1, (int)__builtin_ctzll((s >> 20) | (1ull << 40)));
            slot = n_hot - 1 - slot;
            hot_add<1>(s_hot, s_cnt, slot, drain, t, slot, lane, 1.0f, one);
            if (lane == 0) atomicAdd(&hits[slot], 1ULL);
        }
    }
    hot_drain_block(s_hot, n_hot * 64, rows);
}

extern "C" int dge_selftest_hot_add(int device, int32_t n_hot, int64_t n_workers, int32_t iters, int32_t drain, uint64_t seed,
                                    int64_t* total_additions, double* max_abs_error) {
    if (n_hot <= 0 || n_hot > 118 || n_workers <= 0 || iters <= 0 || drain <= 0 || !total_additions || !max_abs_error)
        DGE_FAIL(DGE_ERR_ARG, "dge_selftest_hot_add: bad argument");
    int rc = dge_require_device(device);
    if (rc) return rc;
    dge_tmp<float> d_rows; dge_tmp<unsigned long long> d_hits;
    if ((rc = d_rows.alloc((size_t)n_hot * 64))) return rc;
    if ((rc = d_hits.alloc((size_t)n_hot))) return rc;
    DGE_HIP(hipMemset(d_rows.p, 0, (size_t)n_hot * 64 * sizeof(float)));
    DGE_HIP(hipMemset(d_hits.p, 0, (size_t)n_hot * sizeof(unsigned long long)));
    const unsigned blocks = (unsigned)((n_workers * 16 + 255) / 256);
    hipLaunchKernelGGL(k_selftest_hot_add, dim3(blocks), dim3(256), (size_t)n_hot * 65 * 4, 0, d_rows.p, d_hits.p, n_hot, drain, iters, seed, n_workers);
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipDeviceSynchronize());
    std::vector<float> rows((size_t)n_hot * 64); std::vector<unsigned long long> hits((size_t)n_hot);
    DGE_HIP(hipMemcpy(rows.data(), d_rows.p, rows.size() * sizeof(float), hipMemcpyDeviceToHost));
    DGE_HIP(hipMemcpy(hits.data(), d_hits.p, hits.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int64_t total = 0; double worst = 0;
    for (int r = 0; r < n_hot; r++) {
        total += (int64_t)hits[(size_t)r];
        for (int e = 0; e < 64; e++) worst = std::max(worst, fabs((double)rows[(size_t)r * 64 + e] - (double)hits[(size_t)r]));
    }
    *total_additions = total; *max_abs_error = worst;
    return DGE_OK;
}

// ------------------------------------------------------------------------------------------ delta exchange
__global__ void k_delta_export(const float* __restrict__ cur, const float* __restrict__ snap, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = cur[i] - snap[i];
}
__global__ void k_delta_import(float* __restrict__ cur, float* __restrict__ snap, const float* __restrict__ in, float scale, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = fmaf(scale, in[i], snap[i]);
        cur[i] = v; snap[i] = v;
    }
}

// ------------------------------------------------------------------------------------------ host side
static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

extern "C" int dge_count_tokens(const dge_walks* w, int64_t row0, int64_t n_rows, int32_t n_vertices, int64_t* d_counts) {
    if (!w || !d_counts || row0 < 0 || n_rows < 0 || row0 + n_rows > w->n || n_vertices <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_count_tokens: bad argument");
    DGE_HIP(hipSetDevice(w->device));
    int64_t n = n_rows * w->L;
    if (n == 0) return DGE_OK;
    unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_count_tokens, dim3(blocks), dim3(256), 0, 0, w->d + row0 * w->L, n, n_vertices, (unsigned long long*)d_counts);
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipStreamSynchronize(0));
    return DGE_OK;
}

static void model_release(dge_model* m) {
    dge_dev_free(m->d_syn0); dge_dev_free(m->d_syn1neg); dge_dev_free(m->d_snap); dge_dev_free(m->d_vocab_ids);
    dge_dev_free(m->d_syn1); dge_dev_free(m->d_hs_off); dge_dev_free(m->d_hs_points); dge_dev_free(m->d_hs_codes);
    dge_dev_free(m->d_counts); dge_dev_free(m->d_remap); dge_dev_free(m->d_table); dge_dev_free(m->d_exp);
    dge_dev_free(m->d_sen); dge_dev_free(m->d_len); dge_dev_free(m->d_wb); dge_dev_free(m->d_scan_tmp); dge_dev_free(m->d_counters); dge_dev_free(m->d_locks);
    for (auto& e : m->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
}

extern "C" void dge_model_free(dge_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    model_release(m);
    delete m;
}

extern "C" int dge_model_set_stream(dge_model* m, void* hip_stream) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_set_stream: null model");
    DGE_HIP(hipSetDevice(m->device));
    DGE_HIP(hipStreamSynchronize(m->stream));
    if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
    m->stream = (hipStream_t)hip_stream;
    m->own_stream = false;
    return DGE_OK;
}

static bool dim_supported(int dch) { return dch == 1 || dch == 2 || dch == 3 || dch == 4 || dch == 6 || dch == 8; }

extern "C" int dge_model_create(int device, const dge_train_config* cfg, const int64_t* d_counts, dge_model** out) {
    if (!out || !cfg || !d_counts) DGE_FAIL(DGE_ERR_ARG, "dge_model_create: null argument");
    *out = nullptr;
    if (cfg->dim <= 0 || cfg->window <= 0 || cfg->negative < 0 || cfg->n_vertices <= 0 || cfg->epochs < 0 || cfg->workers < 0)
        DGE_FAIL(DGE_ERR_ARG, "dge_model_create: dim/window/n_vertices must be positive, negative/epochs/workers non-negative");
    int dch = (cfg->dim + 63) / 64;
    if (!dim_supported(dch)) dch = dch <= 6 ? 6 : 8;
    if (cfg->dim > 512) DGE_FAIL(DGE_ERR_ARG, "dge_model_create: dim %d > 512 is not supported", cfg->dim);
    if (cfg->update_policy < 0 || cfg->update_policy == 4 || cfg->update_policy > 7) DGE_FAIL(DGE_ERR_ARG, "dge_model_create: unknown update_policy %d", cfg->update_policy);
    if (cfg->use_hs && cfg->update_policy != 0 && cfg->update_policy != 2 && cfg->update_policy != 3)
        DGE_FAIL(DGE_ERR_ARG, "dge_model_create: use_hs runs under update_policy 0 (auto), 2 or 3, not %d", cfg->update_policy);
    int rc = dge_require_device(device);
    if (rc) return rc;
    DGE_HIP(hipDeviceSynchronize());      // d_counts may have been produced on the caller's streams (count kernel, all-reduce)
    dge_model* m = new dge_model();
    m->device = device;
    m->cfg = *cfg;
    if (m->cfg.table_size <= 0) m->cfg.table_size = 100000000LL;
    if (m->cfg.table_size >= 0x7fffffffLL) { delete m; DGE_FAIL(DGE_ERR_ARG, "dge_model_create: table_size must be < 2^31"); }
    m->D = cfg->dim; m->stride = dch * 64; m->NV = cfg->n_vertices; m->T = m->cfg.table_size;
    hipError_t he = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete m; DGE_FAIL(DGE_ERR_DEVICE, "hipStreamCreate failed"); }
    m->own_stream = true;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->n_cus = prop.multiProcessorCount; }
    hipStream_t st = m->stream;
    const int32_t NV = m->NV;

#define MC(expr) do { int rc__ = (expr); if (rc__) { model_release(m); delete m; return rc__; } } while (0)
#define MH(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { dge_set_error("HIP error %s at %s:%d", hipGetErrorName(e__), __FILE__, __LINE__); model_release(m); delete m; return DGE_ERR_DEVICE; } } while (0)

    // --- vocabulary: stable descending sort on count (ids ascending inside a tie), keep count >= min_count
    dge_tmp<int32_t> d_ids, d_ids_sorted; dge_tmp<int64_t> d_cnt_sorted; dge_tmp<unsigned long long> d_kept; dge_tmp<char> d_tmp;
    MC(d_ids.alloc((size_t)NV)); MC(d_ids_sorted.alloc((size_t)NV)); MC(d_cnt_sorted.alloc((size_t)NV));
    MC(d_kept.alloc(2));
    hipLaunchKernelGGL(k_iota_i32, dim3(grid_for(NV, 256)), dim3(256), 0, st, d_ids.p, (int64_t)NV);
    size_t tmp_bytes = 0;
    MH(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, d_counts, d_cnt_sorted.p, d_ids.p, d_ids_sorted.p, NV, 0, 64, st));
    MC(d_tmp.alloc(tmp_bytes));
    MH(hipcub::DeviceRadixSort::SortPairsDescending((void*)d_tmp.p, tmp_bytes, d_counts, d_cnt_sorted.p, d_ids.p, d_ids_sorted.p, NV, 0, 64, st));
    MH(hipMemsetAsync(d_kept.p, 0, 2 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_count_kept, dim3(std::min<unsigned>(grid_for(NV, 256), 2048u)), dim3(256), 0, st, d_cnt_sorted.p, (int64_t)NV,
                       (int64_t)cfg->min_count, d_kept.p);
    unsigned long long kept = 0;
    MH(hipMemcpyAsync(&kept, d_kept.p, sizeof(kept), hipMemcpyDeviceToHost, st));
    MH(hipStreamSynchronize(st));
    const int64_t V = (int64_t)kept;
    m->V = V;
    MC(dge_dev_alloc(&m->d_vocab_ids, (size_t)V)); MC(dge_dev_alloc(&m->d_counts, (size_t)V)); MC(dge_dev_alloc(&m->d_remap, (size_t)NV));
    if (V) {
        MH(hipMemcpyAsync(m->d_vocab_ids, d_ids_sorted.p, V * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        MH(hipMemcpyAsync(m->d_counts, d_cnt_sorted.p, V * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    }
    MH(hipMemsetAsync(m->d_remap, 0xFF, (size_t)NV * sizeof(int32_t), st));
    if (V) hipLaunchKernelGGL(k_scatter_remap, dim3(grid_for(V, 256)), dim3(256), 0, st, m->d_vocab_ids, V, m->d_remap);
    m->h_counts.resize((size_t)V); m->h_vocab_ids.resize((size_t)V);
    if (V) {
        MH(hipMemcpyAsync(m->h_counts.data(), m->d_counts, V * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        MH(hipMemcpyAsync(m->h_vocab_ids.data(), m->d_vocab_ids, V * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    MH(hipStreamSynchronize(st));
    int64_t tw = 0;
    for (int64_t i = 0; i < V; i++) tw += m->h_counts[(size_t)i];
    m->total_words = tw;

    // --- unigram^0.75 cumulative (word2vec.c InitUnigramTable's running d1; a serial double sum by definition)
    MC(dge_dev_alloc(&m->d_table, (size_t)m->T));
    if (V > 0) {
        std::vector<double> cum((size_t)V);
        double twp = 0.0; const double power = 0.75;
        for (int64_t i = 0; i < V; i++) twp += pow((double)m->h_counts[(size_t)i], power);
        {
            double s2 = 0.0;
            for (int64_t i = 0; i < V; i++) { double q = pow((double)m->h_counts[(size_t)i], power) / twp; s2 += q * q; }
            m->neg_collision = s2;
            // Head of the vocabulary for the mixed policy (7).  A try-lock fails when another worker holds the row: per pair
            // ~5 syn1neg rows drawn with q_i (unigram^0.75) and one syn0 row that occurs with p_i (unigram), held for the whole
            // pair.  Expected failures per attempt with W workers ~ W * 5 * (sum q_i^2 + sum p_i^2) over the LOCKED rows; the
            // head [0, H) is taken out until that is below 0.1.  (cfg3: 0.14 with H = 0 — left alone, see train_rows; cfg5: 3.3 M
            // rows, H ~ 1e4.)
            const double W0 = (double)((int64_t)m->n_cus * 3 * 16);
            double tail = 0.0; int64_t H = V;
            while (H > 0) {
                const double c = (double)m->h_counts[(size_t)(H - 1)];
                const double q = pow(c, power) / twp, pp = c / (double)tw;
                if (W0 * 5.0 * (tail + q * q + pp * pp) >= 0.1) break;
                tail += q * q + pp * pp; H--;
            }
            m->hot_rows_auto = H;
            // A second, sharper reason to keep a row out of the lock protocol: the pair holds its syn0 row's lock for its whole
            // duration, so the pairs whose context is row i run one after the other — p_i * pairs of them, while the launch as a
            // whole lasts pairs / W pair-times.  A row with W * p_i near 1 therefore becomes the critical path of the launch
            // (measured: ONE vertex with 1e-4 of all tokens in an otherwise flat 1 M-row vocabulary — W * p = 1.2 — slows the lock
            // kernel by 20-25 %; the bench graph's busiest row has 0.36).  Rows beyond 0.5 go to the atomics side.
            int64_t Hs = 0;
            while (Hs < V && W0 * (double)m->h_counts[(size_t)Hs] / (double)tw > 0.5) Hs++;
            m->hot_rows_serial = Hs;
        }
        double d1 = 0.0;
        for (int64_t i = 0; i < V; i++) { d1 = (i == 0) ? pow((double)m->h_counts[0], power) / twp : d1 + pow((double)m->h_counts[(size_t)i], power) / twp; cum[(size_t)i] = d1; }
        dge_tmp<double> d_cum; dge_tmp<int32_t> d_g, d_m; dge_tmp<char> d_tmp2;
        MC(d_cum.alloc((size_t)V)); MC(d_g.alloc((size_t)m->T)); MC(d_m.alloc((size_t)m->T));
        MH(hipMemcpyAsync(d_cum.p, cum.data(), V * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_table_chase, dim3(grid_for(m->T, 256)), dim3(256), 0, st, d_cum.p, V, m->T, d_g.p);
        tmp_bytes = 0;
        MH(hipcub::DeviceScan::ExclusiveScan(nullptr, tmp_bytes, d_g.p, d_m.p, hipcub::Min(), (int32_t)0, m->T, st));
        MC(d_tmp2.alloc(tmp_bytes));
        MH(hipcub::DeviceScan::ExclusiveScan((void*)d_tmp2.p, tmp_bytes, d_g.p, d_m.p, hipcub::Min(), (int32_t)0, m->T, st));
        hipLaunchKernelGGL(k_table_fill, dim3(grid_for(m->T, 256)), dim3(256), 0, st, d_m.p, V, m->T, m->d_table);
        MH(hipStreamSynchronize(st));
    } else {
        MH(hipMemsetAsync(m->d_table, 0, (size_t)m->T * sizeof(int32_t), st));
    }

    // --- sigmoid LUT (word2vec.c expTable) and weights
    {
        float e[EXP_TABLE_SIZE];
        for (int i = 0; i < EXP_TABLE_SIZE; i++) {
            // C semantics of word2vec.c: the argument is a float expression, exp() itself is the DOUBLE function
            float x = (float)exp((double)((i / (float)EXP_TABLE_SIZE * 2 - 1) * MAX_EXP));
            e[i] = x / (x + 1);
        }
        MC(dge_dev_alloc(&m->d_exp, EXP_TABLE_SIZE));
        MH(hipMemcpyAsync(m->d_exp, e, sizeof(e), hipMemcpyHostToDevice, st));
        MH(hipStreamSynchronize(st));
    }
    size_t tab = (size_t)V * (size_t)m->stride;
    MC(dge_dev_alloc(&m->d_syn0, tab + 64)); MC(dge_dev_alloc(&m->d_syn1neg, tab + 64));
    MH(hipMemsetAsync(m->d_syn1neg, 0, (tab + 64) * sizeof(float), st));
    if (V) hipLaunchKernelGGL(k_init_syn0, dim3(grid_for(V, 256)), dim3(256), 0, st, m->d_syn0, V, m->D, m->stride, cfg->seed);
    if (cfg->use_hs) {
        // inner-node table (V rows allocated, V-1 used: the tables stay the same size for the delta exchange) and paths
        const int longest = dge_huffman_paths(m->h_counts.data(), V, m->h_hs_off, m->h_hs_points, m->h_hs_codes);
        MC(dge_dev_alloc(&m->d_syn1, tab + 64));
        MH(hipMemsetAsync(m->d_syn1, 0, (tab + 64) * sizeof(float), st));
        if (longest > 40) { model_release(m); delete m; DGE_FAIL(DGE_ERR_ARG, "dge_model_create: a Huffman code of %d bits exceeds word2vec's MAX_CODE_LENGTH 40", longest); }
        MC(dge_dev_alloc(&m->d_hs_off, (size_t)V + 1)); MC(dge_dev_alloc(&m->d_hs_points, m->h_hs_points.size())); MC(dge_dev_alloc(&m->d_hs_codes, (size_t)V));
        MH(hipMemcpyAsync(m->d_hs_off, m->h_hs_off.data(), ((size_t)V + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        if (!m->h_hs_points.empty()) MH(hipMemcpyAsync(m->d_hs_points, m->h_hs_points.data(), m->h_hs_points.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (V) MH(hipMemcpyAsync(m->d_hs_codes, m->h_hs_codes.data(), (size_t)V * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    }
    MC(dge_dev_alloc(&m->d_locks, 2 * ((size_t)V + 1)));      // [0,V]: syn1neg rows, [V+1,2V+1]: syn0 rows
    MH(hipMemsetAsync(m->d_locks, 0, 2 * ((size_t)V + 1) * sizeof(int), st));
    MC(dge_dev_alloc(&m->d_counters, 2));
    MH(hipMemsetAsync(m->d_counters, 0, 2 * sizeof(unsigned long long), st));
    MH(hipStreamSynchronize(st));
    MH(hipGetLastError());
#undef MC
#undef MH
    *out = m;
    return DGE_OK;
}

static int ensure_work(dge_model* m, int64_t n_rows, int32_t L) {
    if (n_rows <= m->cap_rows && L <= m->cap_L) return DGE_OK;
    DGE_HIP(hipStreamSynchronize(m->stream));
    dge_dev_free(m->d_sen); dge_dev_free(m->d_len); dge_dev_free(m->d_wb); dge_dev_free(m->d_scan_tmp);
    m->d_sen = nullptr; m->d_len = nullptr; m->d_wb = nullptr; m->d_scan_tmp = nullptr;
    int64_t nr = std::max(n_rows, m->cap_rows); int32_t nl = std::max(L, m->cap_L);
    int rc;
    if ((rc = dge_dev_alloc(&m->d_sen, (size_t)(nr * nl)))) return rc;
    if ((rc = dge_dev_alloc(&m->d_len, (size_t)nr))) return rc;
    if ((rc = dge_dev_alloc(&m->d_wb, (size_t)nr))) return rc;
    size_t bytes = 0;
    DGE_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, m->d_len, m->d_wb, nr, m->stream));
    DGE_HIP(hipMalloc(&m->d_scan_tmp, bytes ? bytes : 1));
    m->scan_tmp_bytes = bytes;
    m->cap_rows = nr; m->cap_L = nl;
    return DGE_OK;
}

template <int DCH, bool BIG>
static void launch_train_b(const TrainParams& p, int pol, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st) {
    switch (pol) {
        case 0: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 1: hipLaunchKernelGGL((k_sgns_train<DCH, 1, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 10: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, true, false>), dim3(blocks), dim3(threads), 0, st, p); break;    // + hierarchical softmax
        case 12: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, true, false>), dim3(blocks), dim3(threads), shmem, st, p); break;
        case 5: hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 6: hipLaunchKernelGGL((k_sgns_train_locked<DCH, true, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 7: hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, true, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        // block schedule of the multi-GPU path (dge_model_set_partition): in-order, atomics, commit locks
        case 20: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, false, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 22: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, false, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 25: hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, false, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 27: hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, true, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        default: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
    }
}
template <int DCH>
static void launch_train(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st) {
    if (big) launch_train_b<DCH, true>(p, pol, blocks, threads, shmem, st);
    else launch_train_b<DCH, false>(p, pol, blocks, threads, shmem, st);
}

static int train_rows(dge_model* m, const int32_t* d_rows, int64_t n_rows, int32_t L, int64_t walk_index_base, int32_t epoch,
                      int64_t words_before, double words_scale, int64_t total_walks) {
    if (n_rows == 0 || m->V == 0) return DGE_OK;
    int rc = ensure_work(m, n_rows, L);
    if (rc) return rc;
    hipStream_t st = m->stream;
    hipLaunchKernelGGL(k_remap_compact, dim3(grid_for(n_rows, 256)), dim3(256), 0, st, d_rows, n_rows, L, m->d_remap, m->NV, m->d_sen, m->d_len);
    size_t bytes = m->scan_tmp_bytes;
    DGE_HIP(hipcub::DeviceScan::ExclusiveSum(m->d_scan_tmp, bytes, m->d_len, m->d_wb, n_rows, st));

    TrainParams p;
    p.sen = m->d_sen; p.len = m->d_len; p.wb = m->d_wb;
    p.syn0 = m->d_syn0; p.syn1neg = m->d_syn1neg; p.table = m->d_table; p.exp_table = m->d_exp;
    p.n_rows = n_rows; p.L = L; p.W = m->cfg.window; p.K = m->cfg.negative; p.stride = m->stride;
    p.V = m->V; p.T = m->T; p.seed = m->cfg.seed;
    p.gidx_base = (int64_t)epoch * total_walks + walk_index_base;
    p.words_done_base = (int64_t)epoch * m->total_words + words_before;
    p.all_words = (int64_t)std::max(m->cfg.epochs, 1) * m->total_words;
    p.words_scale = words_scale;
    p.alpha0 = m->cfg.alpha; p.min_alpha = m->cfg.min_alpha;
    p.counters = m->d_counters;
    p.locks = m->d_locks;
    p.syn1 = m->d_syn1; p.hs_off = m->d_hs_off; p.hs_points = m->d_hs_points; p.hs_codes = m->d_hs_codes;
    p.hs_hot0 = 0x7fffffff; p.hs_n_hot = 0; p.hs_drain = 1; p.hot_rows = 0;
    p.part_n = m->part_n; p.part_ctx = m->part_ctx; p.part_tgt = m->part_tgt; p.syn0_free = 0;
    p.big_seg_shift = 0;
    p.filler_row = (int32_t)(0xFFFFFFF0u / ((uint32_t)m->stride * 4u)) - 1;      // offset + the largest in-row displacement stays below 2^32
    if (const char* e = getenv("DGE_BIG_SEG_SHIFT")) p.big_seg_shift = atoi(e);          // tests: several segments on a small table
    const bool part = m->part_n > 1;
    const bool hs = m->cfg.use_hs != 0;
    if (part && hs) DGE_FAIL(DGE_ERR_STATE, "the block schedule (dge_model_set_partition) cannot carry the hierarchical-softmax term: a Huffman path crosses every partition");

    int64_t workers;
    if (m->cfg.workers == 0) {
        // fill the device: 4 blocks of 16 workers per CU, but never more concurrent walks than half the vocabulary
        // (Hogwild's premise is sparse collisions: measured, a 2.3k-row table keeps 0.99 cosine to the in-order
        // result up to ~1k workers and loses it beyond; the reference ran 8 workers on <= 6.4k rows)
        const bool auto_locked = !hs && m->cfg.update_policy == 0 && m->V >= 262144 && (double)((int64_t)m->n_cus * 3 * 16) * 5.0 * m->neg_collision < 0.25;
        const bool auto_mixed = !hs && m->cfg.update_policy == 0 && m->V >= 262144 && ((!auto_locked && m->hot_rows_auto <= m->V / 8) || (auto_locked && m->hot_rows_serial > 0));
        const int blocks_per_cu = (m->cfg.update_policy == 5 || m->cfg.update_policy == 6 || m->cfg.update_policy == 7 || auto_locked || auto_mixed) ? ((m->cfg.update_policy == 7 || auto_mixed) ? (m->stride <= 128 ? DGE_HOTMIX_WAVES : 2) : (m->stride == 64 ? 4 : DGE_LOCKED_WAVES)) : 4;   // (rows of one chunk leave room for a 4th wave per SIMD in the lock kernel; a 5th under atomics gains nothing: cfg2 7.6e8 either way)   // what the kernel's VGPR budget keeps resident
        workers = (int64_t)m->n_cus * blocks_per_cu * 16;
        workers = std::min(workers, std::max<int64_t>(64, m->V / 2));
        workers = std::min(workers, (n_rows + 15) / 16 * 16);
    } else workers = m->cfg.workers;
    p.n_workers = workers;
    // update policy (see Policy<>, k_sgns_train_locked and dge_train_config.update_policy)
    int pol = m->cfg.update_policy;
    if (pol == 0) {
        // auto.  The commit-lock kernel is the fast one while lock attempts rarely fail: a try fails when the row is among
        // the ~5 rows another worker holds, i.e. with probability ~ workers * 5 * sum_i q_i^2 (q = unigram^0.75 sampling
        // probabilities).  cfg3 (uniform-ish, 1M rows, 12k workers): 0.07 -> locked, 8.9e8 edges/s.  A Zipf-popular
        // vocabulary (cfg5) gives >> 1: the same kernel spins on its hot rows (measured 5e5 edges/s) while memory-side
        // atomics are indifferent to the skew (5.9e7 = their byte rate) -> atomics.
        const double fail = (double)((int64_t)m->n_cus * 3 * 16) * 5.0 * m->neg_collision;
        // In between (a skewed head over a long tail — cfg5, and what real trip data looks like) the head rows alone are
        // taken out of the lock protocol: policy 7.
        pol = workers == 1 ? 100 : ((!hs && m->V >= 262144 && fail < 0.25) ? (m->hot_rows_serial > 0 ? 7 : 5) : ((!hs && m->V >= 262144 && m->hot_rows_auto <= m->V / 8) ? 7 : 2));
    }
    if (pol == 7) {
        const double fail_all = (double)((int64_t)m->n_cus * 3 * 16) * 5.0 * m->neg_collision;
        // a flat vocabulary with a few busy rows: only those; a skewed one: the whole head
        p.hot_rows = (int32_t)std::min<int64_t>((m->cfg.update_policy == 0 && fail_all < 0.25) ? m->hot_rows_serial : std::max(m->hot_rows_auto, m->hot_rows_serial), m->V);
        if (const char* e = getenv("DGE_HOT_ROWS")) { long long v = atoll(e); if (v >= 0) p.hot_rows = (int32_t)std::min<int64_t>(v, m->V); }     // tuning/ablation knob
    }
    if (pol == 100 || (workers == 1 && pol != 5 && pol != 6 && pol != 7 && pol != 2 && pol != 1)) pol = 0;   // in-order: plain accesses
    if (pol == 3) pol = 0;
    if (part && L > 64) DGE_FAIL(DGE_ERR_ARG, "the block schedule keeps a walk's tokens in registers: walks of up to 64 tokens, not %d", L);
    if (part) {
        // One block of the multi-GPU schedule: the live rows are V/part_n per table, so lock attempts collide part_n times
        // as often as on the whole table (measured on cfg3 with bench.py --sim-ranks, profiles/r01_block_schedule_sim.txt).
        p.hot_rows = 0;                         // (the head/tail split of policy 7 is not used inside a block)
        if (pol == 0) pol = 20;
        else if (m->cfg.update_policy == 0) {
            const double per_worker = 5.0 * m->neg_collision * (double)m->part_n;
            const int64_t w_max = per_worker > 0 ? (int64_t)(0.37 / per_worker) / 256 * 256 : workers;
            if (m->V >= 262144 && w_max >= workers) pol = 25;                  // cfg3: up to 4 ranks
            else if (m->V >= 262144 && w_max >= 4096) {
                // more ranks: also take the pair's syn0 row out of the lock protocol (it is held for the whole pair: at
                // 8 ranks 10 % of the live syn0 rows are locked at any time and every tenth pair is aborted and retried);
                // its update goes out as atomics behind the last unlock.  8 192 workers (2 resident blocks a CU): 4.6e8 edges/s
                // per rank against 3.7e8 with the syn0 locks and 3.6e8 with atomics everywhere.
                pol = 27; p.hot_rows = 0; p.syn0_free = 1;
                if (m->cfg.workers == 0) { workers = std::min<int64_t>(workers, (int64_t)m->n_cus * 2 * 16); p.n_workers = workers; }
            } else pol = 22;
        }
        else if (pol == 2) pol = 22;
        else if (pol == 5) pol = 25;
        else if (pol == 7) { pol = 27; p.hot_rows = 0; p.syn0_free = 1; }      // locks on syn1neg only (see the auto rule above)
        else DGE_FAIL(DGE_ERR_ARG, "the block schedule runs under update_policy 0 (auto), 2, 3, 5 or 7, not %d", m->cfg.update_policy);
    }
    size_t shmem = 0;
    if (hs) {
        pol = pol == 0 ? 10 : 12;               // dge_model_create admitted policies 0/2/3 only
        if (pol == 12) {
            // LDS accumulators for the inner nodes nearest the root: 30 KB a block (4 blocks a CU stay resident)
            const int64_t row_b = (int64_t)m->stride * 4 + 4;
            p.hs_n_hot = (int32_t)std::min<int64_t>(std::max<int64_t>(m->V - 1, 0), 30720 / row_b);
            p.hs_hot0 = (int32_t)(std::max<int64_t>(m->V - 1, 0) - p.hs_n_hot);
            p.hs_drain = 64;
            if (const char* e = getenv("DGE_HS_DRAIN")) { int v = atoi(e); if (v >= 1) p.hs_drain = v; }      // tuning/ablation knob
            shmem = (size_t)p.hs_n_hot * (size_t)row_b;
        }
    }
    unsigned threads = workers == 1 ? 64u : 256u;
    unsigned blocks = (unsigned)((workers * 16 + threads - 1) / threads);

    // per-segment descriptors (TableView) for tables of 4 GiB and more; DGE_FORCE_BIG=1 selects that code path on small
    // tables too so that the parity tests can cover it
    const bool big = (uint64_t)m->V * (uint64_t)m->stride * 4ull >= 0xFFFFFFFFull || getenv("DGE_FORCE_BIG") != nullptr;
    EventPair ev; ev.kind = 0;
    DGE_HIP(hipEventCreate(&ev.a)); DGE_HIP(hipEventCreate(&ev.b));
    DGE_HIP(hipEventRecord(ev.a, st));
    switch (m->stride / 64) {
        case 1: launch_train<1>(p, pol, big, blocks, threads, shmem, st); break;
        case 2: launch_train<2>(p, pol, big, blocks, threads, shmem, st); break;
        case 3: launch_train<3>(p, pol, big, blocks, threads, shmem, st); break;
        case 4: launch_train<4>(p, pol, big, blocks, threads, shmem, st); break;
        case 6: launch_train<6>(p, pol, big, blocks, threads, shmem, st); break;
        default: launch_train<8>(p, pol, big, blocks, threads, shmem, st); break;
    }
    DGE_HIP(hipEventRecord(ev.b, st));
    m->pending.push_back(ev);
    m->launches++;
    m->last_policy = pol >= 20 ? pol - 20 : (pol >= 10 ? pol - 10 : pol); m->last_workers = workers; m->last_hot_rows = p.hot_rows;
    DGE_HIP(hipGetLastError());
    return DGE_OK;
}

extern "C" int dge_model_train(dge_model* m, const dge_walks* w, int64_t row0, int64_t n_rows, int64_t walk_index_base, int32_t epoch,
                               int64_t words_before, double words_scale, int64_t total_walks) {
    if (!m || !w || row0 < 0 || n_rows < 0 || row0 + n_rows > w->n || epoch < 0 || words_before < 0 || !(words_scale > 0))
        DGE_FAIL(DGE_ERR_ARG, "dge_model_train: bad argument");
    if (w->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_train: corpus and model live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    if (total_walks <= 0) total_walks = w->n;
    return train_rows(m, w->d + row0 * w->L, n_rows, w->L, walk_index_base, epoch, words_before, words_scale, total_walks);
}

extern "C" int dge_model_walk_and_train(dge_model* m, const dge_graph* g, dge_walks* w, int64_t row0, int64_t n_rows, int64_t walk_seed,
                                        int64_t walk_index_base, int32_t epoch, int64_t words_before, double words_scale, int64_t total_walks) {
    if (!m || !g || !w || row0 < 0 || n_rows < 0 || row0 + n_rows > w->n) DGE_FAIL(DGE_ERR_ARG, "dge_model_walk_and_train: bad argument");
    if (!g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_model_walk_and_train: alias tables not built");
    if (w->device != m->device || g->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_walk_and_train: handles live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    EventPair ev; ev.kind = 1;
    DGE_HIP(hipEventCreate(&ev.a)); DGE_HIP(hipEventCreate(&ev.b));
    DGE_HIP(hipEventRecord(ev.a, m->stream));
    int rc = dge_launch_walks_strided(g, m->stream, w->d + row0 * w->L, n_rows, w->L, walk_seed, walk_index_base, nullptr);
    if (rc) return rc;
    DGE_HIP(hipEventRecord(ev.b, m->stream));
    m->pending.push_back(ev);
    if (total_walks <= 0) total_walks = w->n;
    return train_rows(m, w->d + row0 * w->L, n_rows, w->L, walk_index_base, epoch, words_before, words_scale, total_walks);
}

extern "C" int dge_train_sgns_device(const dge_walks* w, const dge_train_config* cfg, dge_model** out) {
    if (!w || !cfg || !out) DGE_FAIL(DGE_ERR_ARG, "dge_train_sgns_device: null argument");
    *out = nullptr;
    if (cfg->n_vertices <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_train_sgns_device: n_vertices must be positive");
    DGE_HIP(hipSetDevice(w->device));
    int64_t* d_counts = nullptr;
    int rc = dge_dev_alloc(&d_counts, (size_t)cfg->n_vertices);
    if (rc) return rc;
    DGE_HIP(hipMemset(d_counts, 0, (size_t)cfg->n_vertices * sizeof(int64_t)));
    rc = dge_count_tokens(w, 0, w->n, cfg->n_vertices, d_counts);
    dge_model* m = nullptr;
    if (!rc) rc = dge_model_create(w->device, cfg, d_counts, &m);
    dge_dev_free(d_counts);
    if (rc) return rc;
    for (int ep = 0; ep < cfg->epochs && !rc; ep++) rc = dge_model_train(m, w, 0, w->n, 0, ep, 0, 1.0, w->n);
    if (!rc) { hipError_t e = hipStreamSynchronize(m->stream); if (e != hipSuccess) { dge_set_error("training failed: %s", hipGetErrorName(e)); rc = DGE_ERR_DEVICE; } }
    if (rc) { dge_model_free(m); return rc; }
    *out = m;
    return DGE_OK;
}

extern "C" int dge_train_sgns(int device, const int32_t* walks, int64_t n_walks, int32_t max_len, const dge_train_config* cfg, dge_model** out) {
    dge_walks* w = nullptr;
    int rc = dge_walks_from_host(device, walks, n_walks, max_len, &w);
    if (rc) return rc;
    rc = dge_train_sgns_device(w, cfg, out);
    dge_walks_free(w);
    return rc;
}

static int sync_tables_to_host(dge_model* m, bool want_syn0, bool want_syn1, bool want_hs = false) {
    DGE_HIP(hipSetDevice(m->device));
    DGE_HIP(hipStreamSynchronize(m->stream));
    size_t tab = (size_t)m->V * (size_t)m->stride;
    std::vector<float> tmp(tab ? tab : 1);
    for (int which = 0; which < 3; which++) {
        if ((which == 0 && !want_syn0) || (which == 1 && !want_syn1) || (which == 2 && !want_hs)) continue;
        const float* src = which == 0 ? m->d_syn0 : (which == 1 ? m->d_syn1neg : m->d_syn1);
        if (tab) DGE_HIP(hipMemcpy(tmp.data(), src, tab * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<float>& dst = which == 0 ? m->h_syn0 : (which == 1 ? m->h_syn1neg : m->h_syn1);
        const int64_t rows = which == 2 ? std::max<int64_t>(m->V - 1, 0) : m->V;
        dst.resize((size_t)rows * (size_t)m->D + 1);
        for (int64_t r = 0; r < rows; r++) memcpy(dst.data() + r * m->D, tmp.data() + r * m->stride, (size_t)m->D * sizeof(float));
    }
    return DGE_OK;
}

extern "C" int dge_model_vectors(dge_model* m, const float** syn0, const int32_t** vocab_ids, int64_t* V, int32_t* dim) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_vectors: null model");
    int rc = sync_tables_to_host(m, true, false);
    if (rc) return rc;
    if (syn0) *syn0 = m->h_syn0.data();
    if (vocab_ids) *vocab_ids = m->h_vocab_ids.data();
    if (V) *V = m->V;
    if (dim) *dim = m->D;
    return DGE_OK;
}
extern "C" int dge_model_syn1neg(dge_model* m, const float** syn1neg) {
    if (!m || !syn1neg) DGE_FAIL(DGE_ERR_ARG, "dge_model_syn1neg: null argument");
    int rc = sync_tables_to_host(m, false, true);
    if (rc) return rc;
    *syn1neg = m->h_syn1neg.data();
    return DGE_OK;
}
extern "C" int dge_model_syn1(dge_model* m, const float** syn1, int64_t* rows) {
    if (!m || !syn1) DGE_FAIL(DGE_ERR_ARG, "dge_model_syn1: null argument");
    if (!m->d_syn1) DGE_FAIL(DGE_ERR_STATE, "dge_model_syn1: the model was created without use_hs");
    int rc = sync_tables_to_host(m, false, false, true);
    if (rc) return rc;
    *syn1 = m->h_syn1.data();
    if (rows) *rows = std::max<int64_t>(m->V - 1, 0);
    return DGE_OK;
}
extern "C" int dge_model_huffman(dge_model* m, const int64_t** offsets, const int32_t** points, const uint64_t** codes) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_huffman: null model");
    if (!m->d_syn1) DGE_FAIL(DGE_ERR_STATE, "dge_model_huffman: the model was created without use_hs");
    if (offsets) *offsets = m->h_hs_off.data();
    if (points) *points = m->h_hs_points.data();
    if (codes) *codes = m->h_hs_codes.data();
    return DGE_OK;
}
extern "C" int dge_model_counts(dge_model* m, const int64_t** counts) {
    if (!m || !counts) DGE_FAIL(DGE_ERR_ARG, "dge_model_counts: null argument");
    *counts = m->h_counts.data();
    return DGE_OK;
}
extern "C" int dge_model_table(dge_model* m, const int32_t** table, int64_t* table_size) {
    if (!m || !table) DGE_FAIL(DGE_ERR_ARG, "dge_model_table: null argument");
    DGE_HIP(hipSetDevice(m->device));
    m->h_table.resize((size_t)m->T);
    DGE_HIP(hipMemcpy(m->h_table.data(), m->d_table, (size_t)m->T * sizeof(int32_t), hipMemcpyDeviceToHost));
    *table = m->h_table.data();
    if (table_size) *table_size = m->T;
    return DGE_OK;
}

static int drain_events(dge_model* m) {
    if (m->pending.empty()) return DGE_OK;
    DGE_HIP(hipSetDevice(m->device));
    DGE_HIP(hipStreamSynchronize(m->stream));
    for (auto& e : m->pending) {
        float ms = 0.f;
        DGE_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        if (e.kind == 0) m->kernel_ms += ms; else m->walk_ms += ms;
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    m->pending.clear();
    return DGE_OK;
}

extern "C" int dge_model_stats(const dge_model* mc, dge_train_stats* out) {
    dge_model* m = const_cast<dge_model*>(mc);
    if (!m || !out) DGE_FAIL(DGE_ERR_ARG, "dge_model_stats: null argument");
    int rc = drain_events(m);
    if (rc) return rc;
    unsigned long long c[2] = {0, 0};
    DGE_HIP(hipMemcpy(c, m->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    out->pairs = (int64_t)c[0]; out->words = (int64_t)c[1];
    out->kernel_ms = m->kernel_ms; out->walk_kernel_ms = m->walk_ms; out->launches = m->launches;
    return DGE_OK;
}

extern "C" int dge_model_schedule(const dge_model* m, int32_t* update_policy, int64_t* workers, int32_t* hot_rows) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_schedule: null model");
    if (m->last_policy < 0) DGE_FAIL(DGE_ERR_STATE, "dge_model_schedule: nothing has been trained yet");
    if (update_policy) *update_policy = m->last_policy;
    if (workers) *workers = m->last_workers;
    if (hot_rows) *hot_rows = m->last_hot_rows;
    return DGE_OK;
}

extern "C" int dge_model_reset_stats(dge_model* m) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_reset_stats: null model");
    int rc = drain_events(m);
    if (rc) return rc;
    DGE_HIP(hipMemset(m->d_counters, 0, 2 * sizeof(unsigned long long)));
    m->kernel_ms = 0; m->walk_ms = 0; m->launches = 0;
    return DGE_OK;
}

extern "C" int dge_write_vec(dge_model* m, const char* const* names, const char* path, int header) {
    if (!m || !path) DGE_FAIL(DGE_ERR_ARG, "dge_write_vec: null argument");
    int rc = sync_tables_to_host(m, true, false);
    if (rc) return rc;
    FILE* f = fopen(path, "w");
    if (!f) DGE_FAIL(DGE_ERR_IO, "dge_write_vec: cannot open %s", path);
    if (header) fprintf(f, "%lld %d\n", (long long)m->V, m->D);
    for (int64_t r = 0; r < m->V; r++) {
        int32_t id = m->h_vocab_ids[(size_t)r];
        if (names && names[id]) fputs(names[id], f); else fprintf(f, "%d", id);
        const float* v = m->h_syn0.data() + r * m->D;
        for (int j = 0; j < m->D; j++) fprintf(f, " %.9g", (double)v[j]);
        fputc('\n', f);
    }
    if (fclose(f) != 0) DGE_FAIL(DGE_ERR_IO, "dge_write_vec: write to %s failed", path);
    return DGE_OK;
}

// ------------------------------------------------------------------------------------------ multi-GPU block schedule
// N ranks, rows split by row % N.  In episode e rank g trains the block (contexts in partition g, centres and negatives
// in partition (g+e) % N) of the SAME global batch of walks: the N blocks of an episode touch disjoint rows of both
// tables, after N episodes every pair has been trained exactly once, and nothing is ever averaged or summed — the
// result is the single-GPU result with the pairs in another order.  syn0 partition g never leaves rank g during
// training; after each episode the ranks exchange the syn1neg partitions they just trained (an all-gather of packed rows).
__global__ void k_partition_pack(const float* __restrict__ table, float* __restrict__ buf, int64_t V, int32_t stride, int32_t n, int32_t part, int64_t rows_padded) {
    const int64_t total = rows_padded * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / stride * n + part;
        buf[i] = r < V ? table[r * stride + i % stride] : 0.f;
    }
}
__global__ void k_partition_unpack(float* __restrict__ table, const float* __restrict__ buf, int64_t V, int32_t stride, int32_t n, int32_t part, int64_t rows_padded) {
    const int64_t total = rows_padded * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / stride * n + part;
        if (r < V) table[r * stride + i % stride] = buf[i];
    }
}

extern "C" int dge_model_set_partition(dge_model* m, int32_t n_parts, int32_t ctx_part, int32_t tgt_part) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_set_partition: null model");
    if (n_parts <= 1) { m->part_n = 1; m->part_ctx = 0; m->part_tgt = 0; return DGE_OK; }
    if (ctx_part < 0 || ctx_part >= n_parts || tgt_part < 0 || tgt_part >= n_parts) DGE_FAIL(DGE_ERR_ARG, "dge_model_set_partition: partition out of range");
    if (m->V < n_parts) DGE_FAIL(DGE_ERR_ARG, "dge_model_set_partition: %d partitions for %lld vocabulary rows", n_parts, (long long)m->V);
    m->part_n = n_parts; m->part_ctx = ctx_part; m->part_tgt = tgt_part;
    return DGE_OK;
}

extern "C" int dge_model_partition_floats(const dge_model* m, int32_t n_parts, int64_t* n_floats) {
    if (!m || !n_floats || n_parts <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_model_partition_floats: bad argument");
    *n_floats = (m->V + n_parts - 1) / n_parts * (int64_t)m->stride;
    return DGE_OK;
}

static int partition_copy(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf, bool pack) {
    if (!m || !d_buf || n_parts <= 0 || part < 0 || part >= n_parts || (table != 0 && table != 1)) DGE_FAIL(DGE_ERR_ARG, "dge_model_%s_partition: bad argument", pack ? "export" : "import");
    DGE_HIP(hipSetDevice(m->device));
    if (!pack) DGE_HIP(hipDeviceSynchronize());          // d_buf comes from the caller's collective, on the caller's stream
    float* tab = table == 0 ? m->d_syn0 : m->d_syn1neg;
    const int64_t rows = (m->V + n_parts - 1) / n_parts;
    if (rows > 0) {
        if (pack) hipLaunchKernelGGL(k_partition_pack, dim3(2048), dim3(256), 0, m->stream, tab, d_buf, m->V, m->stride, n_parts, part, rows);
        else hipLaunchKernelGGL(k_partition_unpack, dim3(2048), dim3(256), 0, m->stream, tab, d_buf, m->V, m->stride, n_parts, part, rows);
    }
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipStreamSynchronize(m->stream));
    return DGE_OK;
}
extern "C" int dge_model_export_partition(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf) { return partition_copy(m, table, n_parts, part, d_buf, true); }
extern "C" int dge_model_import_partition(dge_model* m, int table, int32_t n_parts, int32_t part, const float* d_buf) { return partition_copy(m, table, n_parts, part, const_cast<float*>(d_buf), false); }

// ------------------------------------------------------------------------------------------ multi-GPU exchange
extern "C" int dge_model_sync_size(const dge_model* m, int64_t* n_floats) {
    if (!m || !n_floats) DGE_FAIL(DGE_ERR_ARG, "dge_model_sync_size: null argument");
    *n_floats = (m->d_syn1 ? 3 : 2) * m->V * (int64_t)m->stride;
    return DGE_OK;
}

extern "C" int dge_model_snapshot(dge_model* m) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_snapshot: null model");
    DGE_HIP(hipSetDevice(m->device));
    size_t tab = (size_t)m->V * (size_t)m->stride;
    if (!m->d_snap) { int rc = dge_dev_alloc(&m->d_snap, (m->d_syn1 ? 3 : 2) * tab + 64); if (rc) return rc; }
    DGE_HIP(hipMemcpyAsync(m->d_snap, m->d_syn0, tab * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
    DGE_HIP(hipMemcpyAsync(m->d_snap + tab, m->d_syn1neg, tab * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
    if (m->d_syn1) DGE_HIP(hipMemcpyAsync(m->d_snap + 2 * tab, m->d_syn1, tab * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
    DGE_HIP(hipStreamSynchronize(m->stream));
    return DGE_OK;
}

extern "C" int dge_model_export_delta(dge_model* m, float* d_buf) {
    if (!m || !d_buf) DGE_FAIL(DGE_ERR_ARG, "dge_model_export_delta: null argument");
    if (!m->d_snap) DGE_FAIL(DGE_ERR_STATE, "dge_model_export_delta: call dge_model_snapshot before training the shard");
    DGE_HIP(hipSetDevice(m->device));
    int64_t tab = m->V * (int64_t)m->stride;
    if (tab) {
        hipLaunchKernelGGL(k_delta_export, dim3(2048), dim3(256), 0, m->stream, m->d_syn0, m->d_snap, d_buf, tab);
        hipLaunchKernelGGL(k_delta_export, dim3(2048), dim3(256), 0, m->stream, m->d_syn1neg, m->d_snap + tab, d_buf + tab, tab);
        if (m->d_syn1) hipLaunchKernelGGL(k_delta_export, dim3(2048), dim3(256), 0, m->stream, m->d_syn1, m->d_snap + 2 * tab, d_buf + 2 * tab, tab);
    }
    DGE_HIP(hipStreamSynchronize(m->stream));
    DGE_HIP(hipGetLastError());
    return DGE_OK;
}

extern "C" int dge_model_import_delta(dge_model* m, const float* d_buf, float scale) {
    if (!m || !d_buf) DGE_FAIL(DGE_ERR_ARG, "dge_model_import_delta: null argument");
    if (!m->d_snap) DGE_FAIL(DGE_ERR_STATE, "dge_model_import_delta: no snapshot");
    DGE_HIP(hipSetDevice(m->device));
    DGE_HIP(hipDeviceSynchronize());      // d_buf comes from the caller's collective, on the caller's stream
    int64_t tab = m->V * (int64_t)m->stride;
    if (tab) {
        hipLaunchKernelGGL(k_delta_import, dim3(2048), dim3(256), 0, m->stream, m->d_syn0, m->d_snap, d_buf, scale, tab);
        hipLaunchKernelGGL(k_delta_import, dim3(2048), dim3(256), 0, m->stream, m->d_syn1neg, m->d_snap + tab, d_buf + tab, scale, tab);
        if (m->d_syn1) hipLaunchKernelGGL(k_delta_import, dim3(2048), dim3(256), 0, m->stream, m->d_syn1, m->d_snap + 2 * tab, d_buf + 2 * tab, scale, tab);
    }
    DGE_HIP(hipStreamSynchronize(m->stream));
    DGE_HIP(hipGetLastError());
    return DGE_OK;
}

// ------------------------------------------------------------------------------------------ native RCCL exchange
// For hosts without torch.distributed (the Java/JNI form): the same delta exchange with RCCL called directly.  librccl
// is dlopen()ed on first use, so a process that already carries a RCCL (PyTorch bundles one) is never handed a second
// copy at load time.
#include <dlfcn.h>
struct dge_comm {
    void* nccl = nullptr;       // ncclComm_t
    int rank = 0, nranks = 1, device = 0;
    float* d_buf = nullptr; int64_t buf_floats = 0;
};
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, dge_unique_id, int) = nullptr;     // ncclUniqueId is a 128-byte struct passed by value
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.lib) return DGE_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) DGE_FAIL(DGE_ERR_DEVICE, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, dge_unique_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy) DGE_FAIL(DGE_ERR_DEVICE, "librccl lacks an expected symbol");
    g_rccl.lib = h;
    return DGE_OK;
}
int rccl_fail(int rc, const char* what) {
    DGE_FAIL(DGE_ERR_DEVICE, "RCCL %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}
}  // namespace

extern "C" int dge_comm_unique_id(dge_unique_id* out) {
    if (!out) DGE_FAIL(DGE_ERR_ARG, "dge_comm_unique_id: null output");
    int rc = rccl_load();
    if (rc) return rc;
    int n = g_rccl.GetUniqueId(out);
    return n ? rccl_fail(n, "ncclGetUniqueId") : DGE_OK;
}

extern "C" int dge_comm_create(dge_comm** out, const dge_unique_id* id, int rank, int nranks, int device) {
    if (!out || !id || nranks <= 0 || rank < 0 || rank >= nranks) DGE_FAIL(DGE_ERR_ARG, "dge_comm_create: bad argument");
    *out = nullptr;
    int rc = dge_require_device(device);
    if (rc) return rc;
    if ((rc = rccl_load())) return rc;
    dge_comm* c = new dge_comm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    int n = g_rccl.CommInitRank(&c->nccl, nranks, *id, rank);
    if (n) { delete c; return rccl_fail(n, "ncclCommInitRank"); }
    *out = c;
    return DGE_OK;
}

extern "C" void dge_comm_free(dge_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    dge_dev_free(c->d_buf);
    delete c;
}

// delta = tables - snapshot; all-reduce(sum) over the communicator; tables = snapshot + delta_sum / nranks; re-snapshot
extern "C" int dge_model_allreduce_deltas(dge_model* m, dge_comm* c) {
    if (!m || !c) DGE_FAIL(DGE_ERR_ARG, "dge_model_allreduce_deltas: null argument");
    if (c->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_allreduce_deltas: communicator and model live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    int64_t nfl = 0;
    int rc = dge_model_sync_size(m, &nfl);
    if (rc) return rc;
    if (c->buf_floats < nfl) { dge_dev_free(c->d_buf); c->d_buf = nullptr; if ((rc = dge_dev_alloc(&c->d_buf, (size_t)nfl + 64))) return rc; c->buf_floats = nfl; }
    if ((rc = dge_model_export_delta(m, c->d_buf))) return rc;
    int n = g_rccl.AllReduce(c->d_buf, c->d_buf, (size_t)nfl, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->nccl, m->stream);
    if (n) return rccl_fail(n, "ncclAllReduce");
    DGE_HIP(hipStreamSynchronize(m->stream));
    return dge_model_import_delta(m, c->d_buf, 1.0f / (float)c->nranks);
}

// block schedule, the exchange after episode `episode` (or, with table 0 and episode 0, the final gather of syn0): rank g
// publishes partition (g + episode) % nranks of `table` and takes the partitions the other ranks publish
extern "C" int dge_model_exchange_partitions(dge_model* m, dge_comm* c, int table, int32_t episode) {
    if (!m || !c || (table != 0 && table != 1) || episode < 0) DGE_FAIL(DGE_ERR_ARG, "dge_model_exchange_partitions: bad argument");
    if (c->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_exchange_partitions: communicator and model live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    int64_t pf = 0;
    int rc = dge_model_partition_floats(m, c->nranks, &pf);
    if (rc) return rc;
    const int64_t need = pf * ((int64_t)c->nranks + 1);
    if (c->buf_floats < need) { dge_dev_free(c->d_buf); c->d_buf = nullptr; if ((rc = dge_dev_alloc(&c->d_buf, (size_t)need + 64))) return rc; c->buf_floats = need; }
    float* mine = c->d_buf; float* all = c->d_buf + pf;
    if ((rc = dge_model_export_partition(m, table, c->nranks, (c->rank + episode) % c->nranks, mine))) return rc;
    int n = g_rccl.AllGather(mine, all, (size_t)pf, /*ncclFloat32*/ 7, c->nccl, m->stream);
    if (n) return rccl_fail(n, "ncclAllGather");
    DGE_HIP(hipStreamSynchronize(m->stream));
    for (int r = 0; r < c->nranks; r++)
        if (r != c->rank && (rc = dge_model_import_partition(m, table, c->nranks, (r + episode) % c->nranks, all + (int64_t)r * pf))) return rc;
    return DGE_OK;
}

