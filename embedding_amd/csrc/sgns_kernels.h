// sgns_kernels.h — device code of the skip-gram negative-sampling trainer (gfx950): row movers, the two trainer kernels and
// their launch switch.  Included by sgns.hip (host side, self-tests) and by sgns_train_dch.hip, which is compiled once per
// row width (DCH = 64-float chunks per row) so that the 144 kernel instantiations build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dge_algos.h"
#include "dge_internal.h"

#define EXP_TABLE_SIZE 1000
#define MAX_EXP 6
#define HS_REP 16            /* most copies an inner node has during a launch of k_sgns_train_hsw (the root's) */
#define HS_REP_NODES 64      /* at most this many inner nodes have copies */
#define HS_REP_ROWS ((HS_REP - 1) * HS_REP_NODES)      /* spare rows behind syn1 for them */
#define NEG_BATCH 5
#ifndef DGE_LOCKED_WAVES
#define DGE_LOCKED_WAVES 3
#endif
#ifndef DGE_HOTMIX_WAVES
#define DGE_HOTMIX_WAVES 3
#endif
#ifndef DGE_HS_WAVES
#define DGE_HS_WAVES 3
#endif

// ------------------------------------------------------------------------------------------ trainer
struct TrainParams {
    const int32_t* sen; const int64_t* len; const int64_t* wb;
    float* syn0; float* syn1neg;
    int32_t acc_rows, acc_drain;  // update_policy 7: the hottest rows [0, acc_rows) combine their syn1neg updates in the atomics wave's LDS accumulators, `acc_drain` updates a flush
    const uint4* ctab;        // word2vec's unigram^0.75 table in rank-block form (neg_table_row): 16 B per 96 slots
    // the same table in RUN form (neg_row_by_runs) when the vocabulary has few distinct counts: n_runs runs of equally frequent rows, and the n_exc slots
    // at which the closed form is off by one (found by comparing it with the table slot by slot when the model is created); n_runs == 0: not available
    const double* run_base; const uint32_t* run_row; const uint32_t* exc_slot; const int32_t* exc_row;
    int32_t n_runs, n_exc;
    double T_inv;
    const float* exp_table;
    int64_t n_rows; int32_t L, W, K, stride;
    int32_t D;                // the rows' meaningful floats (stride = D rounded up to 64: the rest is zero padding and stays zero)
    int64_t V, T;
    uint64_t T_magic, W_magic;    // floor((2^64 - 1) / T), floor((2^64 - 1) / W): dge_fast_mod (the item generators of sgns_sorted.hip are instruction bound)
    uint32_t N_magic;             // floor((2^32 - 1) / part_n): dge_fast_div32
    uint64_t seed;
    int64_t gidx_base;        // (epoch*total_walks + walk_index_base): RNG stream key of row 0
    int64_t words_done_base;  // epoch*total_words + words_before
    int64_t all_words;        // epochs*total_words
    double words_scale;
    float alpha0, min_alpha;
    int64_t n_workers;
    int32_t hs_wave;          // hierarchical softmax under atomics: every atomic of a workgroup goes through its atomics wave (12 workers a workgroup)
    int32_t hs_cold;          // hierarchical softmax under atomics: inner nodes [0, hs_cold) are updated by plain read-modify-write
    unsigned long long* counters;
    unsigned long long* next_walk;   // lock kernels: walks are handed out in order, one at a time (zero before every launch): worker w starts on walk w, the
                                     // k-th request gets walk n_workers + k.  nullptr: worker w takes walks w, w + n_workers, ...
    int* locks;               // one commit-lock word per syn1neg row (all zero between launches)
    float* syn1;              // hierarchical softmax: inner-node rows, Huffman paths (null when off)
    const int64_t* hs_off; const int32_t* hs_points; const uint64_t* hs_codes;
    int32_t hs_hot0, hs_n_hot; // inner nodes [hs_hot0, hs_hot0 + hs_n_hot) — the ones nearest the root — combine in LDS
    int32_t hs_drain;         // an LDS accumulator is drained to memory every hs_drain additions
    // k_sgns_train_hsw: the busiest inner nodes [hs_rep0, hs_rep0 + hs_rep_n) — each on a tenth of all paths and more — live in up to HS_REP copies during a launch: copy 0
    // is the row itself, copy c = 1 .. HS_REP-1 is row V + (c-1) * hs_rep_n + node - hs_rep0 of syn1 (HS_REP_ROWS spare rows behind the table: dge_model_create); a reader
    // adds the copies up, a writer's atomics go to ITS copy
    int32_t hs_rep0, hs_rep_n;
    int32_t hs_rep_thr[HS_REP];   // node >= hs_rep_thr[k]: the node has more than k copies (k = 1 .. HS_REP-1; ascending weights: a node's copies grow with its number)
    int32_t hot_rows;         // policy 7: vocabulary rows [0, hot_rows) — the most frequent — are never locked, they take atomics
    // multi-GPU block schedule (dge_model_set_partition): only pairs whose context row is in partition part_ctx and whose
    // centre row is in partition part_tgt (row % part_n) are trained; negatives are moved into partition part_tgt
    int32_t part_n, part_ctx, part_tgt;
    int32_t filler_row;       // a row index whose offset is outside every table descriptor (see row_load): loads of it cost no traffic
    int32_t big_seg_shift;    // BIG: 0, or (tests) a smaller segment size than the 4 GiB window allows
    int32_t syn0_free;        // HOTMIX kernels: the pair's syn0 row is never locked either (read agent-scope, updated with atomics)
    // The lock kernels' watchdog (round 5): a worker that is still WAITING for a row lock wd_ticks (100 MHz s_memrealtime ticks) after its wave started gives up — it
    // releases what it holds, counts itself in counters[3] and leaves its walks untrained; dge_model_stats reports the launch as failed.  Looked at on the waiting
    // paths only (a busy pair row, a round that left rows unwon, a blocking flush): the pair that wins its locks never reads the clock.  0 = off.
    uint64_t wd_ticks;
};
// (the deadline lives in LDS, written once per workgroup: the watchdog costs the lock kernels no register)
__device__ __forceinline__ bool lk_timed_out(const unsigned long long* s_deadline) { return (unsigned long long)wall_clock64() > *s_deadline; }

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

// The unigram^0.75 table of word2vec.c (InitUnigramTable) is a step function: table[a + 1] - table[a] is 0 or 1 (k_table_fill).  So it is kept
// as rank blocks — per 96 slots one 16-byte record {row of the block's first slot, 96 step bits} — and a look-up is ONE 16-byte load plus three
// popcounts: row(a) = first + #steps in (96b, a].  1e8 slots are 16.7 MB instead of 400 MB, cache resident, 8.0e10 instead of 5.5e10 look-ups/s alone;
// inside the trainer it is worth nothing measurable (profiles/r03_placement.txt) — it is kept for the 383 MB.  Same rows as table[a], bit for bit.
#define DGE_CTAB_SLOTS 96u
__device__ __forceinline__ int32_t neg_table_row(const uint4* __restrict__ ctab, uint64_t slot) {
    const uint32_t a = (uint32_t)slot, b = a / DGE_CTAB_SLOTS, j = a - b * DGE_CTAB_SLOTS;
    const uint4 r = ctab[b];
    // steps at positions 1..j of the block (bit 0 of every block is clear)
    const uint32_t m0 = j >= 31u ? 0xFFFFFFFFu : ((2u << j) - 1u);
    const uint32_t m1 = j < 32u ? 0u : (j >= 63u ? 0xFFFFFFFFu : ((2u << (j - 32u)) - 1u));
    const uint32_t m2 = j < 64u ? 0u : (j >= 95u ? 0xFFFFFFFFu : ((2u << (j - 64u)) - 1u));
    return (int32_t)(r.x + (uint32_t)__popc(r.y & m0) + (uint32_t)__popc(r.z & m1) + (uint32_t)__popc(r.w & m2));
}

// The table without the table.  Rows are ordered by count, so rows of equal count form runs, and inside a run word2vec's cumulative d1 grows by the
// same increment: the slot -> row map is a few hundred straight segments.  Run r = rows [row[r], row[r + 1]) with cumulative base[r] before its first
// row (the increment is (base[r + 1] - base[r]) / rows of the run); table[a] = #{rows whose cumulative value is below (a - 1) / T} wherever no slot holds two rows' ends (elsewhere, and
// where the arithmetic here rounds differently from the serial sum, the slot is in the exception list: the model's creation compares this function with
// the real table slot by slot and keeps the run form only if at most DGE_RUN_EXC slots differ).  The arrays live in LDS: a look-up is 8 compares there
// and a handful of f64 operations instead of a request to memory — the lock kernel runs against a request rate and its 5 look-ups a pair cost 7 % of a
// launch (profiles/r03_shape_sweep.txt).  base[] is padded with +inf to DGE_RUN_MAX entries.  The runs cover the vocabulary's tail, rows [row[0], V);
// a slot of the head rows in front of it returns -2: the caller reads the table for it.
#define DGE_RUN_MAX 2048
#define DGE_RUN_EXC 64
template <typename PD, typename PU, typename PI>
__device__ __forceinline__ int32_t neg_row_by_runs(uint32_t a, const PD base, const PU row, const PU exc_slot, const PI exc_row, int n_exc, double T_inv, int64_t V) {
    if (a == 0u) return -2;
    const double x = (double)(a - 1u) * T_inv;
    if (!(base[0] < x)) return -2;                  // the head rows in front of the first run (and slots 0, 1): the table itself answers
    int lo = 0;
#pragma unroll
    for (int st = DGE_RUN_MAX / 2; st >= 1; st >>= 1) if (base[lo + st] < x) lo += st;          // the last boundary below x (base[0] < x)
    const int64_t n = (int64_t)row[lo + 1] - (int64_t)row[lo];                                  // (behind the last run: base = +inf, row = V: n = 0)
    int64_t k = 0;
    if (n > 0) {
        // rows of this run with cumulative value below x: the m-th has base + m (next base - base) / n, so m < q
        const double q = (x - base[lo]) * (double)n / (base[lo + 1] - base[lo]);
        k = (int64_t)ceil(q) - 1;
        k = k < 0 ? 0 : (k > n ? n : k);
    }
    int32_t r = (int32_t)min((int64_t)row[lo] + k, V - 1);
    if (n_exc > 0) {
        int e = 0;
#pragma unroll
        for (int st = DGE_RUN_EXC / 2; st >= 1; st >>= 1) if (e + st < n_exc && exc_slot[e + st] <= a) e += st;
        if (exc_slot[e] == a) r = exc_row[e];
    }
    return r;
}

// x % d for a divisor that is fixed for the launch: q = mulhi(x, floor((2^64 - 1) / d)) falls short of x / d by at most 2 — a dozen instructions
// instead of the ~70 of a 64-bit division, same remainder
__device__ __forceinline__ uint64_t dge_fast_mod(uint64_t x, uint64_t d, uint64_t magic) {
    uint64_t r = x - __umul64hi(x, magic) * d;
    if (r >= d) r -= d;
    if (r >= d) r -= d;
    return r;
}

// x / d for 0 <= x < 2^31 and a launch-constant divisor, the same way
__device__ __forceinline__ int32_t dge_fast_div32(int32_t x, int32_t d, uint32_t magic) {
    uint32_t q = __umulhi((uint32_t)x, magic);
    uint32_t r = (uint32_t)x - q * (uint32_t)d;
    if (r >= (uint32_t)d) { r -= (uint32_t)d; q++; }
    if (r >= (uint32_t)d) { q++; }
    return (int32_t)q;
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

// Sum over the 16 lanes of a group, in every lane: four DPP steps inside the row of 16 — swap inside pairs, swap the pairs of a quad, mirror the half row, mirror the
// row.  The same additions in the same tree as the xor butterfly (1, 2, 4, 8) it replaces (after a step every lane of a quad / half row holds the same bits: a + b ==
// b + a), so results do not move; but __shfl_xor compiles to ds_bpermute — a trip through the LDS crossbar per step, six dot products a pair on the pair's latency chain.
#define DGE_DPP_ADD(v, CTRL) ((v) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (CTRL), 0xF, 0xF, true)))
__device__ __forceinline__ float group16_sum(float p) {
    p = DGE_DPP_ADD(p, 0xB1);       // quad_perm [1, 0, 3, 2]
    p = DGE_DPP_ADD(p, 0x4E);       // quad_perm [2, 3, 0, 1]
    p = DGE_DPP_ADD(p, 0x141);      // row_half_mirror
    p = DGE_DPP_ADD(p, 0x140);      // row_mirror
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
    uint32_t valid;           // floats of a row that are not padding: atomics skip the rest (a D = 20 row is 2 requests of 64 B, not 4)
    uint32_t seg_shift;       // BIG: rows per segment = 1 << seg_shift (the largest power of two whose rows fit a 4 GiB window)
    bool big;
};
__device__ __forceinline__ TableView make_view(float* base, int64_t rows, int stride, int seg_shift_override = 0) {
    TableView t;
    t.base = base;
    t.row_bytes = (uint32_t)stride * 4u;
    t.valid = (uint32_t)stride;
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
        const uint32_t e = (uint32_t)(c * 64 + lane);          // this lane's first element of the chunk; padding takes no atomics
        if (e < t.valid) atomicAdd(p + c * 64 + 0, g * x.v[c].x);
        if (e + 16 < t.valid) atomicAdd(p + c * 64 + 16, g * x.v[c].y);
        if (e + 32 < t.valid) atomicAdd(p + c * 64 + 32, g * x.v[c].z);
        if (e + 48 < t.valid) atomicAdd(p + c * 64 + 48, g * x.v[c].w);
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
                if ((uint32_t)(c * 64 + m * 16 + lane) >= t.valid) continue;
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

// ---- The atomics wave (mixed lock kernels: the head rows' atomics; k_sgns_train with hierarchical softmax: every atomic).
// A wave waits for ITS OWN outstanding memory operations whenever it waits for anything (vmcnt counts loads, stores and atomics alike on
// gfx9), so a worker that issues head-row atomics sits on their completion at its next try-lock — and the memory-side atomic unit, saturated
// by the head, answers slowly: on cfg5 the atomic bytes (2.3 s at 1.24 TB/s) and the plain bytes (2.5 s) ADDED UP to the launch's 4.7 s,
// although the memory system serves both at once (scripts/micro/atomic_overlap.hip: 3.64 ms together, 3.36 + 1.02 alone).  So the last wave
// of every workgroup trains nothing: the 12 workers of the other three waves post (vector, rows, steps) messages into LDS boxes, the atomics
// wave turns them into float atomics and never waits for their completion.  A message: LK_MB_HDR floats of header (count, then 16 rows as
// int bits, then 16 steps) followed by the vector in element order; two boxes per worker.  Box states: 0 = free, 1 = rows of syn1neg, 2 = of syn0,
// 3 = of syn1 (hierarchical softmax; k_sgns_train_hsw's copies of the busiest inner nodes are rows behind the table's V).
#define LK_MB_WORKERS 12
#define LK_MB_HDR 36
template <int DCH> struct LkBox { static constexpr int FLOATS = LK_MB_HDR + DCH * 64; };
__device__ __forceinline__ int lk_flag_load(int* f) { return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lk_flag_store(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// worker side: lane j contributes (row, step) when row >= 0; `vec` is the 16-byte register layout (lane j: elements 64c + 4j .. 4j + 3)
template <int DCH>
__device__ __forceinline__ void lk_post(float* boxes, int* flags, int wk, unsigned& n_posts, int kind, int32_t row, float step, const Row<DCH>& vec, int lane) {
    const int b = wk * 2 + (int)(n_posts & 1u);
    float* box = boxes + (size_t)b * LkBox<DCH>::FLOATS;
    for (;;) {                                              // (both boxes of this worker are still being read: wait for the older one)
        int f = 0;
        if (lane == 0) f = lk_flag_load(&flags[b]);
        if (__shfl(f, 0, 16) == 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    const unsigned have = (unsigned)(__ballot(row >= 0) >> (threadIdx.x & 48)) & 0xFFFFu;
    if (lane == 0) box[0] = __int_as_float((int)have);
    box[1 + lane] = __int_as_float(row);
    box[17 + lane] = step;
#pragma unroll
    for (int c = 0; c < DCH; c++) *(float4*)(box + LK_MB_HDR + c * 64 + 4 * lane) = vec.v[c];
    if (lane == 0) lk_flag_store(&flags[b], kind);         // release: the box's contents are written before the flag turns
    n_posts++;
}
// the same for the 4-byte-per-lane register layout of k_sgns_train's atomic policy (lane j: elements 64c + 16m + j, m = x, y, z, w)
template <int DCH>
__device__ __forceinline__ void lk_post4(float* boxes, int* flags, int wk, unsigned& n_posts, int kind, int32_t row, float step, const Row<DCH>& vec, int lane) {
    const int b = wk * 2 + (int)(n_posts & 1u);
    float* box = boxes + (size_t)b * LkBox<DCH>::FLOATS;
    for (;;) {
        int f = 0;
        if (lane == 0) f = lk_flag_load(&flags[b]);
        if (__shfl(f, 0, 16) == 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    const unsigned have = (unsigned)(__ballot(row >= 0) >> (threadIdx.x & 48)) & 0xFFFFu;
    if (lane == 0) box[0] = __int_as_float((int)have);
    box[1 + lane] = __int_as_float(row);
    box[17 + lane] = step;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        float* v = box + LK_MB_HDR + c * 64 + lane;
        v[0] = vec.v[c].x; v[16] = vec.v[c].y; v[32] = vec.v[c].z; v[48] = vec.v[c].w;
    }
    if (lane == 0) lk_flag_store(&flags[b], kind);
    n_posts++;
}
// atomics wave: all 64 lanes on ONE message at a time — an atomic instruction then covers 256 contiguous bytes of a row (a wave can have ~63
// memory instructions outstanding, and atomics on a saturated unit complete slowly: with one 16-lane group per message, 64 bytes an instruction,
// the wave itself would cap the workgroup's atomic rate).  Returns when every worker of the workgroup has left and every box is free.
// The hottest rows of a skewed vocabulary (update_policy 7; rows are ordered by count, so they are rows [0, n_acc)) do not go out update by update:
// the wave adds their syn1neg updates up in LDS accumulators of its own (acc: n_acc rows; it is the only wave that touches them, so plain LDS
// read-modify-writes) and sends a row's sum as ONE set of atomics after `drain` updates, and whatever is left when the workgroup ends.  On a power law
// with 20 negatives a pair the unigram^0.75 table sends 9 % of all draws to the 30 hottest rows (cfg5: the head's syn1neg atomics are a quarter of a
// launch, profiles/r03_zipf_ablation.txt).  Readers still read memory: they see this workgroup's parked updates of such a row at most `drain` late.
// OFF by default (dge_set_tuning DGE_TUNE_ACC_ROWS): cfg3_zipf gains 2-4 % with 16 rows whatever the drain (4 .. 64), cfg5 nothing, and with 16 updates a
// flush the hottest rows lag enough to move the trained scores (mean score of linked pairs 2.45 against 1.34, AUC unchanged; 1.37 with 4 a flush).
#define LK_ACC_ROWS(DCH) ((DCH) <= 2 ? 16 : 8) /* rows a bank */
// Round 5: two banks of LK_ACC_ROWS accumulators — bank 0 for syn1neg rows (messages of kind 1), bank 1 for syn0 rows (kind 2) — and a slot is the row's rank INSIDE THE
// BLOCK'S PARTITION (row / div: a block of an n-rank schedule only ever meets rows = part (mod n), LkAcc::div = n, ::part_* = the block's partitions; one GPU: div 1).
// One block of the multi-GPU schedule is where the banks pay: a partition's hottest row takes n rows' worth of negative draws and the owner of a busy vertex trains ALL
// its pairs as context, so per episode ONE row takes ~2e5 updates on cfg3_zipf at 8 ranks — and a row's memory-side atomics complete one after the other (~78 ns an
// update: 14 of the episode's 22 ms whatever the head size, the worker count or the syn0 rule, profiles/r05_skewed_knobs_cfg3_zipf.txt).  Summed up `drain` at a time
// in every workgroup, that chain is 1/drain as long.
struct LkAcc { float* acc; int* cnt; int n_tgt, n_ctx, drain, div, part_tgt, part_ctx; };
template <int DCH>
__device__ __forceinline__ void lk_acc_flush(float* acc, int slot, const TableView& t, int32_t row, int wl) {
    float* a = acc + slot * (DCH * 64) + wl;
    float* pr = t.base + (size_t)row * (t.row_bytes / 4) + wl;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        const float v = a[c * 64];
        a[c * 64] = 0.f;
        if ((uint32_t)(c * 64 + wl) < t.valid) atomicAdd(pr + c * 64, v);
    }
}
template <int DCH, int NBOX = LK_MB_WORKERS * 2>
__device__ __forceinline__ void lk_atomics_wave(float* boxes, int* flags, int* done, int n_workers_here, const TableView& syn0, const TableView& syn1neg, const TableView& syn1,
                                                const LkAcc A = LkAcc{nullptr, nullptr, 0, 0, 1, 1, 0, 0}) {
    const int wl = threadIdx.x & 63;
    static_assert(NBOX <= 64, "one flag per lane");
    for (;;) {
        bool any = false;
        // all flags in ONE look (lane b reads flag b; round 4: one LDS read per box and sweep was 56 dependent reads in the seven-wave workgroups, more than the
        // messages found cost), then the boxes that are full in turn
        const int my_flag = wl < NBOX ? lk_flag_load(&flags[wl]) : 0;
        for (unsigned long long full = __ballot(my_flag != 0); full; full &= full - 1ull) {
            const int b = __builtin_ctzll(full);
            const int f = __builtin_amdgcn_readlane(my_flag, b);
            any = true;
            const float* box = boxes + (size_t)b * LkBox<DCH>::FLOATS;
            const unsigned have = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(box[0]));
            const int32_t my_row = __float_as_int(box[1 + (wl & 15)]);
            const float my_step = box[17 + (wl & 15)];
            float v[DCH];
#pragma unroll
            for (int c = 0; c < DCH; c++) v[c] = box[LK_MB_HDR + c * 64 + wl];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                        // the box is in registers
            if (wl == 0) lk_flag_store(&flags[b], 0);
            const TableView& t = f == 2 ? syn0 : (f == 3 ? syn1 : syn1neg);
            for (unsigned left = have; left; left &= left - 1u) {
                const int j = __builtin_ctz(left);
                const int32_t row = __builtin_amdgcn_readlane(my_row, j);
                const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_step), j));
                const int n_here = f == 1 ? A.n_tgt : (f == 2 ? A.n_ctx : 0);
                if (row < n_here * A.div) {                 // one of the partition's hottest rows: into its accumulator
                    const int slot = (A.div == 1 ? row : row / A.div) + (f == 2 ? LK_ACC_ROWS(DCH) : 0);
                    float* a = A.acc + slot * (DCH * 64) + wl;
#pragma unroll
                    for (int c = 0; c < DCH; c++) a[c * 64] = fmaf(g, v[c], a[c * 64]);
                    const int n = __builtin_amdgcn_readfirstlane(A.cnt[slot]) + 1;
                    if (n >= A.drain) lk_acc_flush<DCH>(A.acc, slot, t, row, wl);
                    if (wl == 0) A.cnt[slot] = n >= A.drain ? 0 : n;
                    continue;
                }
                float* pr = t.base + (size_t)row * (t.row_bytes / 4) + wl;
#pragma unroll
                for (int c = 0; c < DCH; c++)
                    if ((uint32_t)(c * 64 + wl) < t.valid) atomicAdd(pr + c * 64, g * v[c]);
            }
        }
        if (!any) {
            const int d = __builtin_amdgcn_readfirstlane(__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (d >= n_workers_here) {                      // every worker has left: whatever it posted is visible now — one last look
                bool left_over = false;
                for (int b = 0; b < NBOX; b++) left_over |= __builtin_amdgcn_readfirstlane(lk_flag_load(&flags[b])) != 0;
                if (!left_over) {
                    for (int r = 0; r < A.n_tgt; r++)       // what is still parked goes out
                        if (__builtin_amdgcn_readfirstlane(A.cnt[r]) > 0) lk_acc_flush<DCH>(A.acc, r, syn1neg, r * A.div + A.part_tgt, wl);
                    for (int r = 0; r < A.n_ctx; r++)
                        if (__builtin_amdgcn_readfirstlane(A.cnt[LK_ACC_ROWS(DCH) + r]) > 0) lk_acc_flush<DCH>(A.acc, LK_ACC_ROWS(DCH) + r, syn0, r * A.div + A.part_ctx, wl);
                    return;
                }
            } else __builtin_amdgcn_s_sleep(2);
        }
    }
}

template <int DCH, int POL, bool BIG, bool HS, bool PART>
__global__ void __launch_bounds__(256, (DCH <= 2 && !BIG) ? (HS ? DGE_HS_WAVES : 4) : 1)
k_sgns_train(TrainParams p) {
    using P = Policy<POL>;
    constexpr bool HOT = HS && P::ATOMIC;         // inner nodes near the root combine their updates in LDS (hot_add)
    // ONE worker under the atomic policy runs the sequential schedule: its float atomics are fire-and-forget, so before it reads rows again it
    // waits for them (vmcnt counts them), and a row drawn twice within a batch is trained in turn — as for the plain policies
    const bool solo = P::ATOMIC && p.n_workers == 1;
#define DGE_SOLO_WAIT() do { if (solo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
    __shared__ float s_exp[EXP_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) float s_mb[HOT ? LK_MB_WORKERS * 2 * LkBox<DCH>::FLOATS : 4];     // the atomics wave's message boxes
    __shared__ int s_mb_flag[LK_MB_WORKERS * 2];
    __shared__ int s_mb_done;
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    if (threadIdx.x < LK_MB_WORKERS * 2) s_mb_flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_mb_done = 0;
    __syncthreads();

    const int lane = threadIdx.x & 15;
    const int wk = threadIdx.x >> 4;
    // Hierarchical softmax under atomics (p.hs_wave): a pair is ~27 rows of atomics in ~7 batches, and a worker that issues them itself waits
    // for each batch's atomics to complete before the next batch's rows arrive (a wave waits for ALL its outstanding memory operations) — on
    // cfg3 126 us a pair.  So the workgroup's fourth wave is the atomics wave (lk_atomics_wave), 12 workers post to it.
    const bool use_mb = HOT && p.hs_wave != 0;
    const bool atomics_wave = use_mb && wk >= LK_MB_WORKERS;
    const int64_t worker = use_mb ? (wk < LK_MB_WORKERS ? (int64_t)blockIdx.x * LK_MB_WORKERS + wk : p.n_workers) : ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    unsigned n_posts = 0;
    float* s_hot = HOT ? s_dyn : nullptr;
    int* s_hot_cnt = HOT ? (int*)(s_dyn + (size_t)p.hs_n_hot * DCH * 64) : nullptr;
    if (HOT) {
        for (int i = threadIdx.x; i < p.hs_n_hot * (DCH * 64 + 1); i += blockDim.x) s_dyn[i] = 0.f;   // +0.0f == int 0
        __syncthreads();
    }
    if (!HOT && worker >= p.n_workers) return;    // (the HOT kernel keeps every thread for its final block-wide drain)

    TableView syn0 = make_view(p.syn0, p.V, p.stride, p.big_seg_shift);
    TableView syn1neg = make_view(p.syn1neg, p.V, p.stride, p.big_seg_shift);
    TableView syn1 = make_view(HS ? p.syn1 : p.syn1neg, p.V, p.stride, p.big_seg_shift);
    syn0.valid = syn1neg.valid = syn1.valid = (uint32_t)p.D;
    if (atomics_wave) {       // (then on to the block-wide drain at the end, with worker = n_workers: no walk)
        const int64_t here = min((int64_t)LK_MB_WORKERS, p.n_workers - (int64_t)blockIdx.x * LK_MB_WORKERS);
        lk_atomics_wave<DCH>(s_mb, s_mb_flag, &s_mb_done, (int)max(here, (int64_t)0), syn0, syn1neg, syn1);
    }
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
    // PART with HS: EVERY centre is visited in every block — for the inner nodes of its path that lie in partition part_tgt (inner-node rows are split by
    // node % n like vocabulary rows; the syn1 partition travels the ring with the syn1neg partition of the same number); the negative-sampling terms of a
    // pair stay with the block of its centre's partition (is_tgt)
    uint64_t ctx_mask = 0, tgt_mask = 0, cen_mask = 0, pair_mask = 0, s_centre = 0;
    bool is_tgt = true;
    int nx_len = 0; int64_t nx_wb = 0; int32_t nx0 = -1, nx1 = -1, nx2 = -1, nx3 = -1;

    // ---- per-worker state: walk w, centre i, next context c (contexts are c..c_hi without i)
    // (the next walk: the worker's own number first, then — p.next_walk set — whatever the launch-wide counter hands out, see k_sgns_train_locked)
    int64_t w = -1, w_next = (HOT && worker >= p.n_workers) ? p.n_rows : worker;                  // surplus workers find no walk
    if (PART) walk_fetch(p, w_next, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);
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
            if (P::ATOMIC) { if (use_mb) lk_post4<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, lane == 0 ? word : -1, 1.0f, dh, lane); else row_atomic_axpy(syn1neg, word, lane, 1.0f, dh); } \
            else row_store<DCH, P::STORE_AUX, BIG>(h, syn1neg, word, lane);                                           \
        }                                                                                                          \
    } while (0)

    for (;;) {
        // ------------------------------------------------------------------ advance to the next (centre, context) pair
        bool new_centre = false, alive = true;
        while (c > c_hi) {
            DGE_CLOSE_CENTRE();
            if (PART) i = first_bit_from(cen_mask, i + 1, len); else i++;
            while (i >= len) {                             // next walk of this worker (empty walks are skipped)
                w = w_next;
                if (w >= p.n_rows) { alive = false; break; }
                if (p.next_walk) {
                    unsigned long long t = 0;
                    if (lane == 0) t = atomicAdd(p.next_walk, 1ull);
                    w_next = (int64_t)shfl16_u64(t, 0) + p.n_workers;
                } else w_next = w + p.n_workers;
                int64_t wb_next = 0;
                if (PART) {                                // prefetched while the previous walk was trained
                    len = nx_len; wb_next = nx_wb; tk0 = nx0; tk1 = nx1; tk2 = nx2; tk3 = nx3;
                    walk_fetch(p, w_next, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);
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
                        cen_mask = HS ? (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) : tgt_mask;
                        i = first_bit_from(cen_mask, 0, len);
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
            const int radius = W - (int)dge_fast_mod(s, (uint64_t)W, p.W_magic);
            c = max(0, i - radius);
            c_hi = min(len - 1, i + radius);
            if (c_hi == i) c_hi--;
            if (c == i) c++;
            new_centre = true;
            if (PART) {
                is_tgt = !HS || ((tgt_mask >> i) & 1ull) != 0;
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
        DGE_SOLO_WAIT();
        row_load<DCH, P::LOAD_AUX, BIG>(l1, syn0, last, lane);
        if (new_centre && (!PART || is_tgt)) {
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
                int32_t t = lane < kc ? p.hs_points[hs_o + kd + lane] : -1;
                if (PART && t >= 0 && t % p.part_n != p.part_tgt) t = -1;          // another block's inner node
                float mb_g = 0.f; bool mb_mine = false;      // atomics wave: this lane's node of the round takes atomics, with this step
                for (int base = 0; base < kc; base += NEG_BATCH) {
                    int32_t tg[NEG_BATCH];
                    Row<DCH> rr[NEG_BATCH];
#pragma unroll
                    for (int q = 0; q < NEG_BATCH; q++) {
                        int32_t v = __shfl(t, (base + q) & 15, 16);
                        tg[q] = (base + q < kc) ? v : -1;
                    }
                    if (PART) { bool any = false; _Pragma("unroll") for (int q = 0; q < NEG_BATCH; q++) any |= tg[q] >= 0; if (!any) continue; }
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
                                    else if (tg[q] < p.hs_cold) {
                                        // a cold inner node (on < 2e-5 of the paths: two workers meet on it within a read-modify-write
                                        // about once in 100-200 updates at the warm end of the class, far less below): the row is already here for the dot product, its update goes
                                        // back as a write-through store — 512 B at the plain rate instead of 512 B at the atomic rate
                                        row_axpy(rr[q], g, l1);
                                        row_store<DCH, 16, BIG>(rr[q], syn1, tg[q], lane);
                                    } else if (use_mb) { if (lane == base + q) { mb_g = g; mb_mine = true; } }
                                    else row_atomic_axpy(syn1, tg[q], lane, g, l1);
                                } else {
                                    row_axpy(rr[q], g, l1);
                                    row_store<DCH, P::STORE_AUX, BIG>(rr[q], syn1, tg[q], lane);
                                }
                            }
                        }
                }
                if (use_mb && ((unsigned)(__ballot(mb_mine) >> (threadIdx.x & 48)) & 0xFFFFu)) lk_post4<DCH>(s_mb, s_mb_flag, wk, n_posts, 3, mb_mine ? t : -1, mb_g, l1, lane);
            }
        }
        if (!PART || is_tgt) {   // d == 0: target = word, label 1 (word2vec order: positive first)
            float f = row_dot(l1, h);
            float g = sgns_g(f, 1.0f, alpha, s_exp);
            row_axpy(neu, g, h);
            row_axpy(h, g, l1);
            if (P::ATOMIC) row_axpy(dh, g, l1);
            h_dirty = true;
        }
        for (int kd = 0; kd < K && (!PART || is_tgt); kd += 16) {
            const int kc = min(16, K - kd);
            // lane j draws negative kd+j
            const uint64_t sl = s * mA + cA;
            int32_t t = -1;
            if (lane < kc) {
                t = neg_table_row(p.ctab, dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                if (PART) t = part_row(t, p.part_n, p.part_tgt, p.V);
                if (t == word) t = -1;
            }
            s = shfl16_u64(sl, kc - 1);
            float mb_g = 0.f; bool mb_mine = false;
            for (int base = 0; base < kc; base += NEG_BATCH) {
                int32_t tg[NEG_BATCH];
#pragma unroll
                for (int q = 0; q < NEG_BATCH; q++) {
                    int32_t v = __shfl(t, (base + q) & 15, 16);
                    tg[q] = (base + q < kc) ? v : -1;
                }
                bool dup = false;
                DGE_SOLO_WAIT();                           // (the previous batch's atomics)
                if (!P::ATOMIC || solo) {
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
                                if (use_mb) { if (lane == base + q) { mb_g = g; mb_mine = true; } }
                                else row_atomic_axpy(syn1neg, tg[q], lane, g, l1);
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
            if (use_mb && ((unsigned)(__ballot(mb_mine) >> (threadIdx.x & 48)) & 0xFFFFu)) lk_post4<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, mb_mine ? t : -1, mb_g, l1, lane);
        }
        if (P::ATOMIC) {
            if (use_mb) lk_post4<DCH>(s_mb, s_mb_flag, wk, n_posts, 2, lane == 0 ? last : -1, 1.0f, neu, lane);
            else row_atomic_axpy(syn0, last, lane, 1.0f, neu);
        } else {
#pragma unroll
            for (int q = 0; q < DCH; q++) {
                l1.v[q].x += neu.v[q].x; l1.v[q].y += neu.v[q].y; l1.v[q].z += neu.v[q].z; l1.v[q].w += neu.v[q].w;
            }
            row_store<DCH, P::STORE_AUX, BIG>(l1, syn0, last, lane);
        }
        if (!PART || is_tgt) my_pairs++;
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
#undef DGE_SOLO_WAIT
    if (lane == 0 && !atomics_wave) {
        if (my_pairs) atomicAdd(&p.counters[0], my_pairs);
        if (my_words) atomicAdd(&p.counters[1], my_words);
        if (use_mb && worker < p.n_workers) __hip_atomic_fetch_add(&s_mb_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // (behind this worker's last post)
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
            if ((uint32_t)(c * 64 + 16 * m + lane) < t.valid) atomicAdd(p + c * 64 + 16 * m, g * v);
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
        for (int m = 0; m < 4; m++)
            if ((uint32_t)(c * 64 + 16 * m + lane) < t.valid) atomicAdd(p + c * 64 + 16 * m, d_base[c * 64 + comp * 16 + 4 * m + hi]);
}

template <int DCH, bool STRICT, bool BIG, bool HOTMIX = false>
__device__ __forceinline__ bool flushA_blocking(const TableView& syn1neg, int* locks, int32_t row, const float* d, int lane, int32_t hot_rows = 0, const unsigned long long* s_deadline = nullptr) {
    if (HOTMIX && row < hot_rows) { ldsA_atomic_add<DCH>(syn1neg, row, lane, d - lane); return true; }
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
            return true;
        }
        if (s_deadline && lk_timed_out(s_deadline)) return false;        // (the watchdog: this delta stays unflushed, the launch is reported as failed)
        __builtin_amdgcn_s_sleep(2);
    }
}

#define LK_NEG_LANES 13      /* lanes 0..12 draw negatives, lane 13 = pending centre flush, lane 14 = the pair's syn0 row */
#ifndef LK_CHUNK
#define LK_CHUNK 10          /* negatives per lock round: a multiple of NEG_BATCH, so no batch of a full chunk loads filler rows */
#endif
// 3 waves per SIMD is the measured optimum for D <= 128: 4 (128 VGPRs) spills 88 B per lane and runs 20 % slower, 2 runs 12 % slower
template <int DCH, bool STRICT, bool BIG, bool HOTMIX, bool PART, bool WDOG = false>
__global__ void __launch_bounds__(256, (DCH <= 2 && !BIG) ? (HOTMIX ? (PART ? 2 : DGE_HOTMIX_WAVES) : (DCH == 1 ? 4 : DGE_LOCKED_WAVES)) : ((HOTMIX && DCH <= 4) ? 2 : 1))
k_sgns_train_locked(TrainParams p) {
    __shared__ float s_exp[EXP_TABLE_SIZE];
    __shared__ float s_dh[(HOTMIX ? LK_MB_WORKERS : 16) * 2 * DCH * 64];     // (a mixed workgroup has 12 workers: its fourth wave is the atomics wave)
    __shared__ __attribute__((aligned(16))) float s_mb[HOTMIX ? LK_MB_WORKERS * 2 * LkBox<DCH>::FLOATS : 4];     // the atomics wave's message boxes
    __shared__ int s_mb_flag[LK_MB_WORKERS * 2];
    __shared__ int s_mb_done;
    __shared__ unsigned long long s_deadline;              // the watchdog's (TrainParams::wd_ticks)
    // WDOG: its own instantiation, launched only where update_policy 5 / 6 was FORCED (p.wd_ticks != 0).  What auto picks the locks for — a flat vocabulary — cannot
    // make them wait, and the headline kernel must not pay for the check: the extra control flow costs 4 registers and 22 spilled scalars, 1.5 % of a cfg3 launch
    // (same box, A / B: 390.9 against 396.8 ms).  The mixed kernels never carry it: the rows that could make them wait are their head, which takes no lock.
    constexpr bool WD = WDOG && !HOTMIX;
    __shared__ float s_acc[HOTMIX ? 2 * LK_ACC_ROWS(DCH) * DCH * 64 : 4];    // the atomics wave's accumulators of the hottest rows, a bank per table (lk_atomics_wave)
    __shared__ int s_acc_cnt[2 * LK_ACC_ROWS(DCH)];
    // the negative-sampling table's run form (neg_row_by_runs), where the model has one (not in the mixed kernels: skewed vocabularies have none)
    __shared__ double s_run_base[HOTMIX ? 1 : DGE_RUN_MAX];
    __shared__ uint32_t s_run_row[HOTMIX ? 1 : DGE_RUN_MAX + 1];
    __shared__ uint32_t s_exc_slot[HOTMIX ? 1 : DGE_RUN_EXC];
    __shared__ int32_t s_exc_row[HOTMIX ? 1 : DGE_RUN_EXC];
    const bool use_runs = !HOTMIX && p.n_runs > 0;
    if (use_runs) {
        for (int i = threadIdx.x; i < DGE_RUN_MAX; i += blockDim.x) s_run_base[HOTMIX ? 0 : i] = p.run_base[i];
        for (int i = threadIdx.x; i < DGE_RUN_MAX + 1; i += blockDim.x) s_run_row[HOTMIX ? 0 : i] = p.run_row[i];
        for (int i = threadIdx.x; i < DGE_RUN_EXC; i += blockDim.x) { s_exc_slot[HOTMIX ? 0 : i] = p.exc_slot[i]; s_exc_row[HOTMIX ? 0 : i] = p.exc_row[i]; }
    }
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    if (threadIdx.x < LK_MB_WORKERS * 2) s_mb_flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_mb_done = 0; s_deadline = p.wd_ticks ? (unsigned long long)wall_clock64() + p.wd_ticks : ~0ull; }
    if (HOTMIX) {
        for (int i = threadIdx.x; i < 2 * LK_ACC_ROWS(DCH) * DCH * 64; i += blockDim.x) s_acc[i] = 0.f;
        if (threadIdx.x < 2 * LK_ACC_ROWS(DCH)) s_acc_cnt[threadIdx.x] = 0;
    }
    __syncthreads();

    const int lane = threadIdx.x & 15;
    const int wk = threadIdx.x >> 4;
    // HOTMIX with more than one worker: 12 workers a workgroup, the fourth wave is the atomics wave (one worker alone issues its atomics itself:
    // its next read of a head row must see them, as the sequential schedule does)
    const bool use_mb = HOTMIX && p.n_workers > 1;
    const int64_t worker = use_mb ? (int64_t)blockIdx.x * LK_MB_WORKERS + wk : ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;

    TableView syn0 = make_view(p.syn0, p.V, p.stride, p.big_seg_shift);
    TableView syn1neg = make_view(p.syn1neg, p.V, p.stride, p.big_seg_shift);
    syn0.valid = syn1neg.valid = (uint32_t)p.D;
    if (use_mb && wk >= LK_MB_WORKERS) {
        const int64_t here = min((int64_t)LK_MB_WORKERS, p.n_workers - (int64_t)blockIdx.x * LK_MB_WORKERS);
        // (accumulators only for rows of the head: the tail's rows never reach this wave.  A block's head holds hot_rows / part_n rows of each partition.)
        const int div = PART ? max(p.part_n, 1) : 1;
        const int n_acc = min(min(p.acc_rows, p.hot_rows / div), LK_ACC_ROWS(DCH));
        lk_atomics_wave<DCH>(s_mb, s_mb_flag, &s_mb_done, (int)max(here, (int64_t)0), syn0, syn1neg, syn1neg,
                             LkAcc{s_acc, s_acc_cnt, n_acc, PART ? n_acc : 0, max(p.acc_drain, 1), div, PART ? p.part_tgt : 0, PART ? p.part_ctx : 0});
        return;
    }
    if (worker >= p.n_workers) return;
    unsigned n_posts = 0;
    int* const locks1 = p.locks;
    int* const locks0 = p.locks + p.V + 1;
    const int32_t hot_rows = HOTMIX ? p.hot_rows : 0;
    // one block of the multi-GPU schedule: how often the lock protocol made this worker wait (dge_model_lock_stats; the block kernels run two waves a SIMD and have the
    // registers to spare — the one-GPU kernels do not count)
    unsigned n_aborted = 0, n_short = 0, n_rounds = 0;

    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= lane; j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }

    const int L = p.L, W = p.W, K = p.K;
    const bool toks_in_regs = L <= 64;
    unsigned long long my_pairs = 0, my_words = 0;

    // A worker's next walk: its own number first, then whatever the launch-wide counter hands out — the workers of a slower compute unit
    // (consecutive processes, and two models of one process, differ: profiles/r02_box_drift.txt) take fewer walks instead of finishing late.
    int64_t w = -1, w_next = worker;
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
    if (PART) walk_fetch(p, w_next, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);

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
    // the watchdog fired while this worker waited for a lock (nothing is held any more): it counts itself and becomes a worker whose walks have run out — the state
    // the main loop leaves through (no walk, no open centre, nothing parked)
#define LK_GIVE_UP()                                                                                                   \
    do {                                                                                                               \
        if (lane == 0) atomicAdd(&p.counters[3], 1ull);                                                                \
        w_next = p.n_rows; i = len; c = c_hi + 1; retry_pair = false; h_dirty = false; pend_row = -1;                  \
    } while (0)
#define LK_CLOSE_CENTRE()                                                                                              \
    do {                                                                                                               \
        if (h_dirty) {                                                                                                 \
            h_dirty = false;                                                                                           \
            if (pend_row >= 0 && !flushA_blocking<DCH, STRICT, BIG, HOTMIX>(syn1neg, locks1, pend_row, my_dh + (cur_buf ^ 1) * DCH * 64 + lane, lane, hot_rows, WD ? &s_deadline : nullptr)) LK_GIVE_UP(); \
            else { pend_row = word; cur_buf ^= 1; }                                                                    \
        }                                                                                                              \
    } while (0)

    for (;;) {
        bool new_centre = false, alive = true;
        while (!retry_pair && c > c_hi) {
            LK_CLOSE_CENTRE();
            if (PART) i = first_bit_from(tgt_mask, i + 1, len); else i++;
            while (i >= len) {
                w = w_next;
                if (w >= p.n_rows) { alive = false; break; }
                // (past the watchdog's deadline no walk is begun: the workers that never had to wait must not train the whole rest of the launch alone)
                if (WD && lk_timed_out(&s_deadline)) { if (lane == 0) atomicAdd(&p.counters[3], 1ull); alive = false; break; }
                if (p.next_walk) {
                    unsigned long long t = 0;
                    if (lane == 0) t = atomicAdd(p.next_walk, 1ull);
                    w_next = (int64_t)shfl16_u64(t, 0) + p.n_workers;
                } else w_next = w + p.n_workers;
                int64_t wb_next = 0;
                if (PART) {                                // prefetched while the previous walk was trained
                    len = nx_len; wb_next = nx_wb; tk0 = nx0; tk1 = nx1; tk2 = nx2; tk3 = nx3;
                    walk_fetch(p, w_next, L, lane, nx_len, nx_wb, nx0, nx1, nx2, nx3);
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
            const int radius = W - (int)dge_fast_mod(s, (uint64_t)W, p.W_magic);
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
                if (!flushA_blocking<DCH, STRICT, BIG, HOTMIX>(syn1neg, locks1, pend_row, my_dh + (cur_buf ^ 1) * DCH * 64 + lane, lane, hot_rows, WD ? &s_deadline : nullptr)) { LK_GIVE_UP(); continue; }
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
                    {
                        const uint64_t slot = dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic);
                        t = use_runs ? neg_row_by_runs((uint32_t)slot, s_run_base, s_run_row, s_exc_slot, s_exc_row, p.n_exc, p.T_inv, p.V) : -2;
                        if (t == -2) t = neg_table_row(p.ctab, slot);
                    }
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
                    if (use_mb) {
                        const bool mine = lane < kc && ((got13 >> lane) & 1u) && t < hot_rows;
                        if ((unsigned)(__ballot(mine) >> (threadIdx.x & 48)) & 0xFFFFu) lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, mine ? t : -1, my_hot_g, l1, lane);
                    } else
                    for (int j = 0; j < kc; j++) {
                        const int32_t tj = __shfl(t, j, 16);
                        const float gj = __shfl(my_hot_g, j, 16);
                        if (((got13 >> j) & 1u) && tj < hot_rows) rowA_atomic_axpy<DCH>(syn1neg, tj, lane, gj, l1);
                    }
                    if (hot_flush) {
                        if (use_mb) {                       // the parked delta of a head centre: element 64q + 4*lane + component sits at 64q + 16*component + lane
                            const float* d = my_dh + (cur_buf ^ 1) * DCH * 64 + lane;
                            Row<DCH> dv;
#pragma unroll
                            for (int q = 0; q < DCH; q++) { dv.v[q].x = d[q * 64]; dv.v[q].y = d[q * 64 + 16]; dv.v[q].z = d[q * 64 + 32]; dv.v[q].w = d[q * 64 + 48]; }
                            lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, lane == 0 ? pend_row : -1, 1.0f, dv, lane);
                        } else ldsA_atomic_add<DCH>(syn1neg, pend_row, lane, my_dh + (cur_buf ^ 1) * DCH * 64);
                    }
                }
                pend13 &= ~got13;
                if (gotf) { flush_pending = false; pend_row = -1; if (lane == 13) t = -1; }
                if (PART) { n_rounds++; if (pend13) n_short++; }
                if (pend13) {
                    if (WD && lk_timed_out(&s_deadline)) {       // the watchdog: the round's locks have dropped; what is still held is the pair's syn0 row
                        if (lane == 14 && !(HOTMIX && (last < hot_rows || p.syn0_free))) row_unlock<STRICT>(locks0, last);
                        LK_GIVE_UP();
                        kd = K;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            kd += LK_CHUNK;
        } while (kd < K && !abort_pair);
        if (WD && i >= len) continue;                      // (gave up inside a round)
        if (abort_pair) {
            if (PART) n_aborted++;
            if (WD && lk_timed_out(&s_deadline)) { LK_GIVE_UP(); continue; }      // (nothing is held here)
            retry_pair = true;
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        retry_pair = false;

#pragma unroll
        for (int q = 0; q < DCH; q++) {
            l1.v[q].x += neu.v[q].x; l1.v[q].y += neu.v[q].y; l1.v[q].z += neu.v[q].z; l1.v[q].w += neu.v[q].w;
        }
        if (HOTMIX && (last < hot_rows || p.syn0_free)) {
            if (use_mb) lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 2, lane == 0 ? last : -1, 1.0f, neu, lane);
            else rowA_atomic_axpy<DCH>(syn0, last, lane, 1.0f, neu);
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
    if (pend_row >= 0 && !flushA_blocking<DCH, STRICT, BIG, HOTMIX>(syn1neg, locks1, pend_row, my_dh + (cur_buf ^ 1) * DCH * 64 + lane, lane, hot_rows, WD ? &s_deadline : nullptr)) LK_GIVE_UP();
#undef LK_TOK
#undef LK_POSITIVE
#undef LK_CLOSE_CENTRE
#undef LK_GIVE_UP
    if (lane == 0) {
        if (my_pairs) atomicAdd(&p.counters[0], my_pairs);
        if (my_words) atomicAdd(&p.counters[1], my_words);
        if (PART) { atomicAdd(&p.counters[4], (unsigned long long)n_aborted); atomicAdd(&p.counters[5], (unsigned long long)n_short); atomicAdd(&p.counters[6], (unsigned long long)n_rounds); }
        if (use_mb) __hip_atomic_fetch_add(&s_mb_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // (behind this worker's last post)
    }
}

// ------------------------------------------------------------------------------------------ hierarchical softmax, a wave per centre (round 4)
// k_sgns_train<.., HS> trains pair after pair: every (centre, context) pair climbs the centre's whole Huffman path, ~20 inner nodes of 512 bytes each read
// AND updated per pair, the updates as memory-side float atomics — 10 KB of atomic traffic a pair at 1.13 TB/s: the bound of that kernel (0.49 of the
// roofline on cfg3, profiles/r01_cfg3_hs_pmc.csv).  But the path belongs to the CENTRE: all contexts of a centre (L = W = 24: 16 on average) meet the same
// nodes.  Here a WAVE takes a walk centre by centre, its four 16-lane groups share the centre's path (group g holds nodes g, g + 4, g + 8, ... in
// registers, HSW_NQ of them each: the row as it moves and the update it has gathered), the centre's contexts are trained four at a time (one per group for
// the negative-sampling half; for the tree half every group applies all four contexts to ITS nodes, context after context, and the four partial
// "neu" sums meet through cross-group shuffles), and a node's gathered update leaves ONCE PER CENTRE: into the workgroup's LDS accumulators near the
// root (hot_add), as a plain read-modify-write at the cold end of the tree, through the atomics wave in between.  Per pair the tree half then costs one
// sixteenth of its reads and of its atomics.  Within the wave the node rows see every update in word2vec's order (context after context); what other
// waves do to a node meanwhile is seen at the next centre: Hogwild staleness of one centre's pairs.  The negative-sampling half (positive target in
// registers per group, K negatives by atomics through the atomics wave, the context row by atomics) is that of k_sgns_train<.., 2, .., HS>, draw for draw.
// Rows move 16 bytes per lane.  Walks of up to 64 tokens, rows of up to 128 floats, more than one worker; everything else runs k_sgns_train.
#define HSW_NQ (DCH <= 2 ? 6 : 3) /* path nodes a group holds in registers: 4 x 6 = the 24 nodes nearest the root on rows of up to 128 floats, 4 x 3 = 12 on rows of up to 256 (round 5: the
                                    same 96 registers); deeper ones go pair by pair */
template <int DCH>
__device__ __forceinline__ void hot_addA(float* s_hot, int* s_cnt, int slot, int drain, const TableView& t, int32_t row, int lane, const Row<DCH>& x) {
    float* a = s_hot + slot * (DCH * 64) + 4 * lane;
#pragma unroll
    for (int c = 0; c < DCH; c++) {
        atomicAdd(a + c * 64 + 0, x.v[c].x); atomicAdd(a + c * 64 + 1, x.v[c].y); atomicAdd(a + c * 64 + 2, x.v[c].z); atomicAdd(a + c * 64 + 3, x.v[c].w);
    }
    int n = 0;
    if (lane == 0) n = atomicAdd(&s_cnt[slot], 1) + 1;
    n = __shfl(n, 0, 16);
    if (n % drain == 0) {
        // the accumulator is in element order, so the drain may take any lane mapping: lane j takes elements 64c + 16m + j — an atomic instruction then
        // covers 64 CONTIGUOUS bytes of the row, one request (with the adders' 16-byte layout it would be four quarter-filled ones: the hottest rows'
        // drains serialise at the memory side, 12 ns a request — cfg3: 2.27e8 edges/s with the strided drain)
        float* e = s_hot + slot * (DCH * 64) + lane;
        float* gp = t.base + (size_t)row * (t.row_bytes / 4) + lane;
#pragma unroll
        for (int c = 0; c < DCH; c++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if ((uint32_t)(c * 64 + 16 * m + lane) >= t.valid) continue;
                const float v = atomicExch(e + c * 64 + 16 * m, 0.f);
                if (v != 0.f) atomicAdd(gp + c * 64 + 16 * m, v);
            }
    }
}
template <int DCH>
__device__ __forceinline__ void row_add(Row<DCH>& y, const Row<DCH>& x) {
#pragma unroll
    for (int c = 0; c < DCH; c++) { y.v[c].x += x.v[c].x; y.v[c].y += x.v[c].y; y.v[c].z += x.v[c].z; y.v[c].w += x.v[c].w; }
}
// (a node's copies in proportion to its share of the paths: the root HS_REP, a node on half the paths half as many, ... — a reader pays a row read per copy)
__device__ __forceinline__ int hsw_n_copies(const TrainParams& p, int32_t nd) {
    int r = 1;
#pragma unroll
    for (int k = 1; k < HS_REP; k++) r += nd >= p.hs_rep_thr[k] ? 1 : 0;
    return r;
}
// adds the copies 1 .. n-1 of node nd to its row r (loaded by the caller), four rows in flight at a time (a copy index behind the last reads the last again: not added)
template <int DCH>
__device__ __forceinline__ void hsw_add_copies(Row<DCH>& r, int32_t nd, const TableView& syn1, const TrainParams& p, int lane) {
    const int nc = hsw_n_copies(p, nd);
    const int32_t j = (int32_t)p.V + nd - p.hs_rep0;
    for (int c = 1; c < nc; c += 4) {
        Row<DCH> t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rowA_load<DCH, 16, false>(t[u], syn1, (min(c + u, nc - 1) - 1) * p.hs_rep_n + j, lane);
#pragma unroll
        for (int u = 0; u < 4; u++) if (c + u < nc) row_add(r, t[u]);
    }
}
// NLOCK (flat vocabularies, where the commit locks work — auto_policy 5): the K negatives of a pair are read-modify-written under their rows' commit locks (the
// protocol of k_sgns_train_locked: try-lock rounds that never wait while holding, 16-byte write-through rows, relaxed commit) instead of going out as float atomics —
// 2.5 of the 3.6 KB of atomics a pair, which is what the kernel runs against; the centre's gathered update is then flushed under the row's lock as well (by ONE group,
// after the four groups' shares met through shuffles: a lock taken by one group of a wave must never be waited for by another), so that syn1neg is only ever
// updated under locks.  The context row (syn0) stays on atomics.
// NW = waves of a workgroup that train (one more issues the atomics): 3, two workgroups a compute unit — or 7 in ONE workgroup of 512 threads (with NLOCK, where the
// atomics wave has half the messages to serve): the LDS accumulators of the nodes next to the root are then shared by 7 waves instead of 3, so at the same number of
// additions parked device-wide (workgroups x hs_drain) each is drained half as often — the root's row takes every centre's update and its atomics complete one
// 64-byte request per ~12 ns: at 3.3e8 edges/s and a drain every 4 additions that row alone was busy half the time —, 7 of a compute unit's 8 waves train instead
// of 6.  (How many accumulators: sgns.hip — fewer than the small workgroups hold turned out better.)
template <int DCH, bool NLOCK, int NW, bool HEAD = false>      // HEAD (with NLOCK): the vocabulary's head [0, p.hot_rows) takes atomics, only the tail's negatives go under locks
__global__ void __launch_bounds__((NW + 1) * 64, (NW == 3 && DCH <= 2) ? 2 : 1)
k_sgns_train_hsw(TrainParams p) {
    const int32_t hot_rows = (NLOCK && HEAD) ? p.hot_rows : 0;
    constexpr int NBOX = NW * 4 * 2;                       // two message boxes per 16-lane group that trains
    __shared__ float s_exp[EXP_TABLE_SIZE];
    __shared__ __attribute__((aligned(16))) float s_mb[NBOX * LkBox<DCH>::FLOATS];
    __shared__ int s_mb_flag[NBOX];
    __shared__ int s_mb_done;
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    if (threadIdx.x < NBOX) s_mb_flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_mb_done = 0;
    float* s_hot = s_dyn;
    int* s_hot_cnt = (int*)(s_dyn + (size_t)p.hs_n_hot * DCH * 64);
    for (int i = threadIdx.x; i < p.hs_n_hot * (DCH * 64 + 1); i += blockDim.x) s_dyn[i] = 0.f;   // +0.0f == int 0
    __syncthreads();

    const int lane = threadIdx.x & 15, grp = (threadIdx.x >> 4) & 3, wl = threadIdx.x & 63;
    const int wk = threadIdx.x >> 4;                       // this group's message boxes (groups 0 .. 4 NW - 1 train, the last wave issues the atomics)
    const int wv = threadIdx.x >> 6;
    TableView syn0 = make_view(p.syn0, p.V, p.stride), syn1neg = make_view(p.syn1neg, p.V, p.stride), syn1 = make_view(p.syn1, p.V + HS_REP_ROWS, p.stride);
    syn0.valid = syn1neg.valid = syn1.valid = (uint32_t)p.D;
    const int64_t wave = (int64_t)blockIdx.x * NW + wv;    // p.n_workers = waves that train
    if (wv == NW) {
        const int64_t waves_here = min((int64_t)NW, p.n_workers - (int64_t)blockIdx.x * NW);
        lk_atomics_wave<DCH, NBOX>(s_mb, s_mb_flag, &s_mb_done, (int)max(waves_here, (int64_t)0) * 4, syn0, syn1neg, syn1);
    }
    // The busiest inner nodes (p.hs_rep_n of them, the root first among them) take an update from every centre — or every second, fourth … — and float atomics on
    // ONE row complete at 78 ns a row (scripts/micro/hot_row_spread.hip: 1.3e7 a second, against 2.2e7 centres a second here).  Parking their updates in LDS
    // accumulators (above; still what the pair-by-pair kernel does) trades that for staleness, and on a skewed tree the staleness costs quality (the Zipf community
    // graph's epoch: AUC 0.9541 with a drain every 4 additions, 0.9585 with every addition drained at half the speed).  So these few rows are kept in HS_REP copies for
    // the launch: a wave adds to ITS copy (wave % copies of the node; copy 0 is the row itself) and reads the sum of all copies — nothing is parked, nobody's update
    // waits for a drain, and no row takes more than a sixteenth of all centres' updates (the root's copies against speed, cfg3: 2 1.97e8, 4 2.98e8, 8 3.55e8, 16 3.70e8,
    // 24 3.61e8, 32 3.55e8 edges/s — fewer serialise, more cost the readers their row reads).  train_rows zeroes the copies before the launch and folds them into the rows behind it.
#define HSW_COPY_OF(nd_) (((int)(wave & (HS_REP - 1)) * hsw_n_copies(p, nd_)) >> 4)       /* this wave's copy of the node (0: the row itself); HS_REP = 16 */
    unsigned n_posts = 0;
    const int L = p.L, W = p.W, K = p.K;
    // lane j turns a state into the state j + 1 draws on; (mK, cK): K draws on
    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= lane; j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }
    uint64_t mK = 1, cK = 0;
    for (int j = 0; j < K; j++) { mK *= DGE_W2V_MULT; cK = cK * DGE_W2V_MULT + 11; }
    unsigned long long my_pairs = 0, my_words = 0;

    int64_t w_next = (wv == NW || wave >= p.n_workers) ? p.n_rows : wave;
    while (w_next < p.n_rows) {
        const int64_t w = w_next;
        if (p.next_walk) {
            unsigned long long t = 0;
            if (wl == 0) t = atomicAdd(p.next_walk, 1ull);
            w_next = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t)) + p.n_workers;
        } else w_next = w + p.n_workers;
        const int len = (int)p.len[w];
        if (len < 2) { my_words += (unsigned long long)max(len, 0); continue; }
        my_words += (unsigned long long)len;
        const int32_t tok = wl < len ? p.sen[w * L + wl] : -1;            // lane l of the wave holds token l (L <= 64)
        const int64_t wbw = p.wb[w];
        const int64_t done = p.words_done_base + (p.words_scale == 1.0 ? wbw : (int64_t)((double)wbw * p.words_scale));
        float alpha = (float)((double)p.alpha0 * (1.0 - (double)done / (double)(p.all_words + 1)));
        if (alpha < p.min_alpha) alpha = p.min_alpha;
        const int64_t gbase = (p.gidx_base + w) * (int64_t)L;

        for (int i = 0; i < len; i++) {
            // ---- open centre i: DL4J's window draw, the Huffman path, this group's nodes of it
            const int32_t word = __builtin_amdgcn_readlane(tok, i);
            uint64_t s = dge_mix64(p.seed + (uint64_t)(gbase + i));
            s = s * DGE_W2V_MULT + 11;
            const int radius = W - (int)dge_fast_mod(s, (uint64_t)W, p.W_magic);
            const int lo = max(0, i - radius), hi = min(len - 1, i + radius);
            const int n_ctx = hi - lo;                                    // positions lo .. hi without i
            if (n_ctx <= 0) continue;
            const int64_t hs_o = p.hs_off[word];
            const int P = (int)(p.hs_off[word + 1] - hs_o);
            const uint64_t hs_bits = p.hs_codes[word];
            int32_t node[HSW_NQ];
            Row<DCH> S[HSW_NQ], dS[HSW_NQ];
#pragma unroll
            for (int q = 0; q < HSW_NQ; q++) {
                const int k = 4 * q + grp;
                node[q] = k < P ? p.hs_points[hs_o + k] : -1;
                rowA_load<DCH, 16, false>(S[q], syn1, node[q] >= 0 ? node[q] : p.filler_row, lane);
                row_zero(dS[q]);
            }
            // (behind the six row loads, which go out together: the copies of the few busy nodes among them — on a balanced tree the path's first four, one a group)
            // (a node on a tenth of all paths lies at depth <= 5 of a Huffman tree — its weight is at most 2 / Fib(depth + 2) —: among a group's first two nodes)
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (node[q] >= p.hs_rep0) hsw_add_copies<DCH>(S[q], node[q], syn1, p, lane);
            Row<DCH> h, dh;                                               // the positive target syn1neg[word]: a copy per group, its gathered update
            rowA_load<DCH, 16, false>(h, syn1neg, word, lane);
            row_zero(dh);
            bool dh_dirty = false;

            for (int base = 0; base < n_ctx; base += 4) {
                const int n_here = min(4, n_ctx - base);
                const bool active = grp < n_here;
                int c = lo + base + grp;
                if (c >= i) c++;
                const int32_t last = __shfl(tok, active ? c : 0, 64);
                Row<DCH> l1, neu;
                rowA_load<DCH, 16, false>(l1, syn0, active ? last : p.filler_row, lane);
                row_zero(neu);
                // ---- negative-sampling half, this group's context: positive target first (word2vec order), then the K negatives of the pair's draws
                if (active) {
                    const float f = row_dot(l1, h);
                    const float g = sgns_g(f, 1.0f, alpha, s_exp);
                    row_axpy(neu, g, h); row_axpy(h, g, l1); row_axpy(dh, g, l1);
                    dh_dirty = true;
                }
                uint64_t sg = s;                                          // the centre's stream at this pair: grp pairs of K draws on
                for (int z = 0; z < grp; z++) sg = sg * mK + cK;
                for (int kd = 0; kd < K; kd += 16) {
                    const int kc = min(16, K - kd);
                    const uint64_t sl = sg * mA + cA;
                    int32_t t = -1;
                    if (active && lane < kc) {
                        t = neg_table_row(p.ctab, dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                        if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                        if (t == word) t = -1;
                    }
                    sg = shfl16_u64(sl, kc - 1);
                    // NLOCK: the negatives of the vocabulary's tail under their rows' commit locks; the head rows [0, p.hot_rows) — a skewed vocabulary's, which many
                    // workers want at once — take atomics like everything does without NLOCK (round 5: the head / tail split of update_policy 7 inside this kernel)
                    const int32_t t_at = NLOCK ? ((HEAD && t >= 0 && t < hot_rows) ? t : -1) : t;
                    if (NLOCK) {
                        // try-lock rounds over this pair's negatives: the rows won are trained NEG_BATCH at a time under their locks and released before the next
                        // round; nothing is ever waited for while a lock is held (the four groups of the wave loop independently)
                        unsigned pend = (unsigned)(__ballot(t >= 0 && t >= hot_rows) >> (threadIdx.x & 48)) & 0xFFFFu;
                        while (pend) {
                            const bool want = lane < kc && ((pend >> lane) & 1u);
                            const bool won = want ? row_trylock(p.locks, t) : false;
                            const unsigned got = (unsigned)(__ballot(won) >> (threadIdx.x & 48)) & 0xFFFFu & pend;
                            for (int b0 = 0; b0 < kc; b0 += NEG_BATCH) {
                                const unsigned gb = (got >> b0) & ((1u << NEG_BATCH) - 1u);
                                if (!gb) continue;
                                int32_t tg[NEG_BATCH];
                                Row<DCH> rr[NEG_BATCH];
#pragma unroll
                                for (int q = 0; q < NEG_BATCH; q++) tg[q] = __shfl(t, (b0 + q) & 15, 16);
#pragma unroll
                                for (int q = 0; q < NEG_BATCH; q++) rowA_load<DCH, 16, false>(rr[q], syn1neg, ((gb >> q) & 1u) ? tg[q] : p.filler_row, lane);
#pragma unroll
                                for (int q = 0; q < NEG_BATCH; q++)
                                    if ((gb >> q) & 1u) {
                                        const float f = row_dot(l1, rr[q]);
                                        const float g = sgns_g(f, 0.0f, alpha, s_exp);
                                        row_axpy(neu, g, rr[q]);
                                        row_axpy(rr[q], g, l1);
                                        rowA_store<DCH, 16, false>(rr[q], syn1neg, tg[q], lane);
                                    }
                            }
                            row_commit_wait(0.f);
                            if (won) row_unlock<false>(p.locks, t);
                            pend &= ~got;
                            if (pend) __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    if ((!NLOCK || HEAD) && ((unsigned)(__ballot(t_at >= 0) >> (threadIdx.x & 48)) & 0xFFFFu)) {
                    float mb_g = 0.f;
                    for (int b0 = 0; b0 < kc; b0 += NEG_BATCH) {
                        int32_t tg[NEG_BATCH];
                        Row<DCH> rr[NEG_BATCH];
#pragma unroll
                        for (int q = 0; q < NEG_BATCH; q++) { const int32_t v = __shfl(t_at, (b0 + q) & 15, 16); tg[q] = (b0 + q < kc) ? v : -1; }
                        if (NLOCK) { bool any = false; _Pragma("unroll") for (int q = 0; q < NEG_BATCH; q++) any |= tg[q] >= 0; if (!any) continue; }
#pragma unroll
                        for (int q = 0; q < NEG_BATCH; q++) rowA_load<DCH, 16, false>(rr[q], syn1neg, tg[q] >= 0 ? tg[q] : p.filler_row, lane);
#pragma unroll
                        for (int q = 0; q < NEG_BATCH; q++)
                            if (tg[q] >= 0) {
                                const float f = row_dot(l1, rr[q]);
                                const float g = sgns_g(f, 0.0f, alpha, s_exp);
                                row_axpy(neu, g, rr[q]);
                                if (lane == b0 + q) mb_g = g;
                            }
                    }
                    lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, t_at, mb_g, l1, lane);
                    }
                }
                for (int z = 0; z < n_here; z++) s = s * mK + cK;         // the centre's stream behind this round's pairs
                // ---- tree half: every group applies the round's contexts, one after the other, to ITS nodes of the path
                for (int j = 0; j < n_here; j++) {
                    Row<DCH> lj, part;
#pragma unroll
                    for (int cc = 0; cc < DCH; cc++) {
                        lj.v[cc].x = __shfl(l1.v[cc].x, j * 16 + lane, 64); lj.v[cc].y = __shfl(l1.v[cc].y, j * 16 + lane, 64);
                        lj.v[cc].z = __shfl(l1.v[cc].z, j * 16 + lane, 64); lj.v[cc].w = __shfl(l1.v[cc].w, j * 16 + lane, 64);
                    }
                    row_zero(part);
#pragma unroll
                    for (int q = 0; q < HSW_NQ; q++)
                        if (node[q] >= 0) {
                            const float f = row_dot(lj, S[q]);
                            if (f > -(float)MAX_EXP && f < (float)MAX_EXP) {      // word2vec.c: outside the table the step is skipped
                                const int idx = (int)((f + (float)MAX_EXP) * (float)(EXP_TABLE_SIZE / MAX_EXP / 2));
                                const float code = (float)((hs_bits >> (4 * q + grp)) & 1ULL);
                                const float g = (1.0f - code - s_exp[idx]) * alpha;
                                row_axpy(part, g, S[q]); row_axpy(S[q], g, lj); row_axpy(dS[q], g, lj);
                            }
                        }
                    for (int k = 4 * HSW_NQ + grp; k < P; k += 4) {        // beyond the 24 (12) nodes in registers: the deepest nodes of a long path, pair by pair —
                        const int32_t nd = p.hs_points[hs_o + k];          // cold ones as a rule, but a chain-like tree (very skewed counts) has busy nodes down there too:
                        Row<DCH> r;                                        // each node is updated the way its class is updated everywhere else
                        rowA_load<DCH, 16, false>(r, syn1, nd, lane);
                        const float f = row_dot(lj, r);
                        if (f > -(float)MAX_EXP && f < (float)MAX_EXP) {
                            const int idx = (int)((f + (float)MAX_EXP) * (float)(EXP_TABLE_SIZE / MAX_EXP / 2));
                            const float g = (1.0f - (float)((hs_bits >> k) & 1ULL) - s_exp[idx]) * alpha;
                            row_axpy(part, g, r);
                            if (nd >= p.hs_hot0) {
                                Row<DCH> gl;
                                row_zero(gl); row_axpy(gl, g, lj);
                                hot_addA<DCH>(s_hot, s_hot_cnt, nd - p.hs_hot0, p.hs_drain, syn1, nd, lane, gl);
                            } else if (nd < p.hs_cold) {
                                row_axpy(r, g, lj);
                                rowA_store<DCH, 16, false>(r, syn1, nd, lane);
                            } else rowA_atomic_axpy<DCH>(syn1, nd, lane, g, lj);
                        }
                    }
#pragma unroll
                    for (int cc = 0; cc < DCH; cc++) {                      // the four groups' shares of context j's neu
                        part.v[cc].x += __shfl_xor(part.v[cc].x, 16, 64); part.v[cc].y += __shfl_xor(part.v[cc].y, 16, 64);
                        part.v[cc].z += __shfl_xor(part.v[cc].z, 16, 64); part.v[cc].w += __shfl_xor(part.v[cc].w, 16, 64);
                        part.v[cc].x += __shfl_xor(part.v[cc].x, 32, 64); part.v[cc].y += __shfl_xor(part.v[cc].y, 32, 64);
                        part.v[cc].z += __shfl_xor(part.v[cc].z, 32, 64); part.v[cc].w += __shfl_xor(part.v[cc].w, 32, 64);
                    }
                    if (grp == j) row_add(neu, part);
                }
                if (active) lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 2, lane == 0 ? last : -1, 1.0f, neu, lane);      // syn0[last] += neu
            }
            my_pairs += (unsigned long long)n_ctx;
            // ---- close the centre: what it gathered leaves once
            if (NLOCK) {
                // the four groups' shares of the centre's update meet (a group that trained no context holds zeros); group 0 adds the sum to the row under its lock
#pragma unroll
                for (int cc = 0; cc < DCH; cc++) {
                    dh.v[cc].x += __shfl_xor(dh.v[cc].x, 16, 64); dh.v[cc].y += __shfl_xor(dh.v[cc].y, 16, 64);
                    dh.v[cc].z += __shfl_xor(dh.v[cc].z, 16, 64); dh.v[cc].w += __shfl_xor(dh.v[cc].w, 16, 64);
                    dh.v[cc].x += __shfl_xor(dh.v[cc].x, 32, 64); dh.v[cc].y += __shfl_xor(dh.v[cc].y, 32, 64);
                    dh.v[cc].z += __shfl_xor(dh.v[cc].z, 32, 64); dh.v[cc].w += __shfl_xor(dh.v[cc].w, 32, 64);
                }
                if (HEAD && grp == 0 && word < hot_rows) lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, lane == 0 ? word : -1, 1.0f, dh, lane);      // (a head row: never locked)
                else if (grp == 0)
                    for (;;) {
                        const bool won = lane == 0 ? row_trylock(p.locks, word) : false;
                        if (__shfl((int)won, 0, 16)) {
                            Row<DCH> cur;
                            rowA_load<DCH, 16, false>(cur, syn1neg, word, lane);
                            row_add(cur, dh);
                            rowA_store<DCH, 16, false>(cur, syn1neg, word, lane);
                            row_commit_wait(0.f);
                            if (won) row_unlock<false>(p.locks, word);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
            } else if (dh_dirty) lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 1, lane == 0 ? word : -1, 1.0f, dh, lane);
#pragma unroll
            for (int q = 0; q < HSW_NQ; q++)
                if (node[q] >= 0) {
                    if (q < 2 && node[q] >= p.hs_rep0) {
                        const int my_copy = HSW_COPY_OF(node[q]);
                        lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 3, lane != 0 ? -1 : (my_copy == 0 ? node[q] : (int32_t)p.V + (my_copy - 1) * p.hs_rep_n + (node[q] - p.hs_rep0)), 1.0f, dS[q], lane);
                    } else if (node[q] >= p.hs_hot0) hot_addA<DCH>(s_hot, s_hot_cnt, node[q] - p.hs_hot0, p.hs_drain, syn1, node[q], lane, dS[q]);
                    else if (node[q] < p.hs_cold) {
                        Row<DCH> r;
                        rowA_load<DCH, 16, false>(r, syn1, node[q], lane);
                        row_add(r, dS[q]);
                        rowA_store<DCH, 16, false>(r, syn1, node[q], lane);
                    } else lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, 3, lane == 0 ? node[q] : -1, 1.0f, dS[q], lane);
                }
        }
    }
    if (wv != NW) {
        if (wl == 0) {
            if (my_pairs) atomicAdd(&p.counters[0], my_pairs);
            if (my_words) atomicAdd(&p.counters[1], my_words);
        }
        if (lane == 0 && wave < p.n_workers) __hip_atomic_fetch_add(&s_mb_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // (behind this group's last post)
    }
    hot_drain_block(s_hot, p.hs_n_hot * DCH * 64, p.syn1 + (size_t)p.hs_hot0 * (DCH * 64));
}

// ------------------------------------------------------------------------------------------ small rows: 32 lanes a worker, a float a lane (round 5)
// The reference's own layer size is 20 (tract level, J/DeepWalk.java:62-66) on a vocabulary of 6 408 rows: 1.6 MB of tables, nothing for the HBM to do.  What such
// a launch runs against is the REQUEST rate of the memory-side atomics on a handful of lines: k_sgns_train's 16-lane groups move a 20-float row as 16 + 4 lanes — two
// load requests and two atomic requests a row — and scripts/micro/small_row_atomics.hip (profiles/r05_small_row_atomics.txt) measures 9.3e9 row updates/s for ONE
// request of 20 contiguous lanes against 5.1e9 for the 16 + 4 form at saturation (and the same factor on the one busiest row's chain).  So rows of 17 .. 32 floats under
// the atomics policy get this kernel: a worker is HALF A WAVE, lane j holds element j of every row it touches, a row is one 128-byte request each way.  The
// schedule is k_sgns_train<.., atomics>'s, draw for draw (same window draws, same negatives from the same per-pair stream, positive first, the centre's syn1neg row
// and its gathered update in registers for all its contexts); only the lane that holds an element differs, so one worker alone trains what that kernel trains
// up to the order of the 32 products inside a dot product.  No hierarchical softmax, no block schedule, walks of up to 64 tokens: everything else stays with k_sgns_train.
__device__ __forceinline__ float small_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sum over the 32 lanes of a half wave, in every lane: four DPP steps inside a row of 16 (quad swaps, half-row mirror, row mirror), then the two rows of the half wave
// meet through gfx950's v_permlane16_swap (odd rows of one operand <-> even rows of the other: with both operands = v the results are {row0, row0, row2, row2} and
// {row1, row1, row3, row3}).  All VALU, no trip through the LDS crossbar: a pair's six dot products are on its latency chain.
#define SMALL_DPP_ADD(v, CTRL) ((v) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (CTRL), 0xF, 0xF, true)))
__device__ __forceinline__ float group32_sum(float v) {
    v = SMALL_DPP_ADD(v, 0xB1);      // quad_perm [1, 0, 3, 2]
    v = SMALL_DPP_ADD(v, 0x4E);      // quad_perm [2, 3, 0, 1]
    v = SMALL_DPP_ADD(v, 0x141);     // row_half_mirror
    v = SMALL_DPP_ADD(v, 0x140);     // row_mirror
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(v), (unsigned)__float_as_int(v), false, false);
    return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
}
__device__ __forceinline__ uint64_t shfl32_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl((int)lo, src, 32);
    hi = (uint32_t)__shfl((int)hi, src, 32);
    return ((uint64_t)hi << 32) | lo;
}
#define SMALL_NEG_BATCH 8
template <int LG>      // lanes a worker (32)
__global__ void __launch_bounds__(256, 5)      // (five workgroups a compute unit: 10 240 workers resident, 96 registers)
k_sgns_train_small(TrainParams p) {
    static_assert(LG == 32, "half a wave a worker");
    __shared__ float s_exp[EXP_TABLE_SIZE];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    __syncthreads();
    const int lane = threadIdx.x & (LG - 1);
    const int64_t worker = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG;
    if (worker >= p.n_workers) return;
    const bool on = lane < p.D;                           // this lane holds an element
    // ONE worker (DGE_TUNE_SMALL_ROWS = 1 forces the kernel for it: parity tests) runs the sequential schedule, as in k_sgns_train: it waits for its own atomics before
    // it reads rows again, and trains a batch's negatives in turn
    const bool solo = p.n_workers == 1;
#define SMALL_SOLO_WAIT() do { if (solo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
    float* const syn0 = p.syn0 + lane;
    float* const syn1neg = p.syn1neg + lane;
    const int64_t stride = p.stride;

    // lane j (and j + 16) turns the pair's LCG state s into the state after (j & 15) + 1 draws: negatives are drawn sixteen at a time, as in k_sgns_train
    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= (lane & 15); j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }

    const int L = p.L, W = p.W, K = p.K;
    unsigned long long my_pairs = 0, my_words = 0;
    int64_t w = -1, w_next = worker;
    int len = 0, i = 0, c = 1, c_hi = 0;
    int32_t tk0 = -1, tk1 = -1;                           // the walk's tokens: lane j holds tokens j and j + 32
    int32_t word = 0;
    float alpha = 0.f;
    uint64_t s = 0;
    int64_t gbase = 0;
    float h = 0.f, dh = 0.f;                              // syn1neg[word] and its gathered update (this lane's element)
    bool h_dirty = false;
    bool pre_ok = false; float l1_pre = 0.f; int32_t t_pre = -1; uint64_t s_pre = 0;     // the centre's next pair, asked for one pair ahead
#define SMALL_TOK(idx) ((idx) < 32 ? __shfl(tk0, (idx), 32) : __shfl(tk1, (idx) - 32, 32))

    for (;;) {
        bool new_centre = false, alive = true;
        while (c > c_hi) {
            if (h_dirty) { h_dirty = false; if (on) atomicAdd(syn1neg + (int64_t)word * stride, dh); }
            i++;
            while (i >= len) {                             // next walk of this worker (empty walks are skipped)
                w = w_next;
                if (w >= p.n_rows) { alive = false; break; }
                if (p.next_walk) {
                    unsigned long long t = 0;
                    if (lane == 0) t = atomicAdd(p.next_walk, 1ull);
                    w_next = (int64_t)shfl32_u64(t, 0) + p.n_workers;
                } else w_next = w + p.n_workers;
                len = (int)p.len[w];
                i = 0;
                if (len > 0) {
                    my_words += (unsigned long long)len;
                    const int32_t* sen = p.sen + w * L;
                    tk0 = lane < L ? sen[lane] : -1;
                    tk1 = lane + 32 < L ? sen[lane + 32] : -1;
                    const int64_t wbw = p.wb[w];
                    const int64_t done = p.words_done_base + (p.words_scale == 1.0 ? wbw : (int64_t)((double)wbw * p.words_scale));
                    alpha = (float)((double)p.alpha0 * (1.0 - (double)done / (double)(p.all_words + 1)));
                    if (alpha < p.min_alpha) alpha = p.min_alpha;
                    gbase = (p.gidx_base + w) * (int64_t)L;
                }
            }
            if (!alive) break;
            word = SMALL_TOK(i);
            s = dge_mix64(p.seed + (uint64_t)(gbase + i));
            s = s * DGE_W2V_MULT + 11;
            const int radius = W - (int)dge_fast_mod(s, (uint64_t)W, p.W_magic);
            c = max(0, i - radius);
            c_hi = min(len - 1, i + radius);
            if (c_hi == i) c_hi--;
            if (c == i) c++;
            new_centre = true;
        }
        if (!alive) break;
        const int32_t last = SMALL_TOK(c);

        // ---- one pair: l1 = syn0[last], targets in syn1neg: the centre (label 1) first, then K negatives
        SMALL_SOLO_WAIT();
        // (the context row and the negatives' table look-ups of a centre's NEXT pair are asked for one pair ahead — see below; a new centre starts afresh)
        const bool have_pre = pre_ok && !new_centre;
        const float l1 = have_pre ? l1_pre : (on ? small_ld(syn0 + (int64_t)last * stride) : 0.f);
        if (new_centre) { h = on ? small_ld(syn1neg + (int64_t)word * stride) : 0.f; dh = 0.f; }
        float neu;
        {
            const float f = group32_sum(l1 * h);
            const float g = sgns_g(f, 1.0f, alpha, s_exp);
            neu = g * h;
            h = fmaf(g, l1, h);
            dh = fmaf(g, l1, dh);
            h_dirty = true;
        }
        for (int kd = 0; kd < K; kd += 16) {
            const int kc = min(16, K - kd);
            int32_t t = -1;
            if (have_pre) { t = t_pre; s = s_pre; }        // (K <= 16: one chunk)
            else {
                const uint64_t sl = s * mA + cA;
                if (lane < kc) {
                    t = neg_table_row(p.ctab, dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                    if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                    if (t == word) t = -1;
                }
                s = shfl32_u64(sl, kc - 1);
            }
            // One pair ahead (several workers, K <= 16): the next context row of this centre and the table look-ups of its negatives leave NOW, in front of this
            // pair's row loads — a pair is a latency chain (look-up -> rows -> atomics) and this takes two of its three trips off it.  The prefetched context row
            // misses this pair's own update when a token stands twice in a centre's window: Hogwild staleness of one pair, as between workers.
            pre_ok = false;
            if (!solo && K <= 16) {
                int c2 = c + 1;
                if (c2 == i) c2++;
                if (c2 <= c_hi) {
                    const int32_t last2 = SMALL_TOK(c2);
                    l1_pre = on ? small_ld(syn0 + (int64_t)last2 * stride) : 0.f;
                    const uint64_t sl2 = s * mA + cA;
                    t_pre = -1;
                    if (lane < K) {
                        t_pre = neg_table_row(p.ctab, dge_fast_mod(sl2 >> 16, (uint64_t)p.T, p.T_magic));
                        if (t_pre == 0 && p.V > 1) t_pre = (int32_t)(sl2 % (uint64_t)(p.V - 1)) + 1;
                        if (t_pre == word) t_pre = -1;
                    }
                    s_pre = shfl32_u64(sl2, K - 1);
                    pre_ok = true;
                }
            }
            for (int base = 0; base < kc; base += SMALL_NEG_BATCH) {
                int32_t tg[SMALL_NEG_BATCH];
                float rr[SMALL_NEG_BATCH];
#pragma unroll
                for (int q = 0; q < SMALL_NEG_BATCH; q++) {
                    const int32_t v = __shfl(t, (base + q) & 15, 32);
                    tg[q] = (base + q < kc) ? v : -1;
                }
                // all rows of the batch in flight together (a skipped slot loads nothing); one worker alone: row after row, each behind the last one's atomics
                if (!solo) {
#pragma unroll
                    for (int q = 0; q < SMALL_NEG_BATCH; q++) rr[q] = (on && tg[q] >= 0) ? small_ld(syn1neg + (int64_t)tg[q] * stride) : 0.f;
                }
                if (!solo) {                              // the batch's dot products side by side (independent chains), then the updates
                    float f[SMALL_NEG_BATCH];
#pragma unroll
                    for (int q = 0; q < SMALL_NEG_BATCH; q++) f[q] = group32_sum(l1 * rr[q]);
#pragma unroll
                    for (int q = 0; q < SMALL_NEG_BATCH; q++)
                        if (tg[q] >= 0) {
                            const float g = sgns_g(f[q], 0.0f, alpha, s_exp);
                            neu = fmaf(g, rr[q], neu);
                            if (on) atomicAdd(syn1neg + (int64_t)tg[q] * stride, g * l1);
                        }
                } else {
#pragma unroll 1
                    for (int q = 0; q < SMALL_NEG_BATCH; q++)
                        if (tg[q] >= 0) {
                            SMALL_SOLO_WAIT();
                            const float r1 = on ? small_ld(syn1neg + (int64_t)tg[q] * stride) : 0.f;
                            const float g = sgns_g(group32_sum(l1 * r1), 0.0f, alpha, s_exp);
                            neu = fmaf(g, r1, neu);
                            if (on) atomicAdd(syn1neg + (int64_t)tg[q] * stride, g * l1);
                        }
                }
            }
        }
        if (on) atomicAdd(syn0 + (int64_t)last * stride, neu);
        my_pairs++;
        c++;
        if (c == i) c++;
    }
    if (h_dirty && on) atomicAdd(syn1neg + (int64_t)word * stride, dh);
#undef SMALL_TOK
#undef SMALL_SOLO_WAIT
    if (lane == 0) {
        if (my_pairs) atomicAdd(&p.counters[0], my_pairs);
        if (my_words) atomicAdd(&p.counters[1], my_words);
    }
}

template <int DCH, bool BIG>
static inline void launch_train_b(const TrainParams& p, int pol, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st) {
    switch (pol) {
        case 0: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 1: hipLaunchKernelGGL((k_sgns_train<DCH, 1, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 10: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, true, false>), dim3(blocks), dim3(threads), 0, st, p); break;    // + hierarchical softmax
        case 12: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, true, false>), dim3(blocks), dim3(threads), shmem, st, p); break;
        case 13: if constexpr (DCH <= 4 && !BIG) hipLaunchKernelGGL((k_sgns_train_hsw<DCH, false, 3>), dim3(blocks), dim3(threads), shmem, st, p); break;   // hierarchical softmax, a wave per centre
        case 14:                                                                                                                                             // ... the negatives under commit locks
            if constexpr (DCH <= 4 && !BIG) {
                if (p.hot_rows > 0) hipLaunchKernelGGL((k_sgns_train_hsw<DCH, true, 3, true>), dim3(blocks), dim3(threads), shmem, st, p);                   // (a skewed vocabulary: its head by atomics)
                else hipLaunchKernelGGL((k_sgns_train_hsw<DCH, true, 3>), dim3(blocks), dim3(threads), shmem, st, p);
            }
            break;
        case 15:                                                                                                                                             // ... seven such waves a workgroup
            if constexpr (DCH <= 2 && !BIG) {
                // (the workgroup's 41 KB of static LDS and these up to 60 KB together pass 64 KB: asked for per kernel; an error comes back from the launch)
                if (p.hot_rows > 0) {
                    (void)hipFuncSetAttribute((const void*)k_sgns_train_hsw<DCH, true, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                    hipLaunchKernelGGL((k_sgns_train_hsw<DCH, true, 7, true>), dim3(blocks), dim3(threads), shmem, st, p);
                } else {
                    (void)hipFuncSetAttribute((const void*)k_sgns_train_hsw<DCH, true, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                    hipLaunchKernelGGL((k_sgns_train_hsw<DCH, true, 7>), dim3(blocks), dim3(threads), shmem, st, p);
                }
            }
            break;
        case 5:
            if (p.wd_ticks) hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, false, false, true>), dim3(blocks), dim3(threads), 0, st, p);      // (forced: with the watchdog)
            else hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p);
            break;
        case 6:
            if (p.wd_ticks) hipLaunchKernelGGL((k_sgns_train_locked<DCH, true, BIG, false, false, true>), dim3(blocks), dim3(threads), 0, st, p);
            else hipLaunchKernelGGL((k_sgns_train_locked<DCH, true, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p);
            break;
        case 7: hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, true, false>), dim3(blocks), dim3(threads), 0, st, p); break;
        // block schedule of the multi-GPU path (dge_model_set_partition): in-order, atomics, commit locks
        case 20: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, false, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 22: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, false, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        case 30: hipLaunchKernelGGL((k_sgns_train<DCH, 0, BIG, true, true>), dim3(blocks), dim3(threads), 0, st, p); break;      // ... with the hierarchical softmax
        case 32: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, true, true>), dim3(blocks), dim3(threads), shmem, st, p); break;
        case 25:
            if (p.wd_ticks) hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, false, true, true>), dim3(blocks), dim3(threads), 0, st, p);
            else hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, false, true>), dim3(blocks), dim3(threads), 0, st, p);
            break;
        case 27: hipLaunchKernelGGL((k_sgns_train_locked<DCH, false, BIG, true, true>), dim3(blocks), dim3(threads), 0, st, p); break;
        default: hipLaunchKernelGGL((k_sgns_train<DCH, 2, BIG, false, false>), dim3(blocks), dim3(threads), 0, st, p); break;
    }
}
template <int DCH>
static inline void launch_train(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st) {
    if (big) launch_train_b<DCH, true>(p, pol, blocks, threads, shmem, st);
    else launch_train_b<DCH, false>(p, pol, blocks, threads, shmem, st);
}

