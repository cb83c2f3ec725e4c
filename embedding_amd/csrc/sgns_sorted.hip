// sgns_sorted.hip — update_policy 8: the OWNER-COMPUTES schedule of the skip-gram negative-sampling trainer (gfx950).
//
// The Hogwild kernels of sgns_kernels.h move every row a pair touches as a read-modify-write that must exclude other workers (row
// locks) or combine at the memory side (float atomics).  That is the right shape while the tables are far larger than what is in
// flight (cfg3: 0.8 of the HBM roofline).  It is the wrong shape when the LIVE rows are few: a 100 k-row vocabulary (cfg2), one block
// of the 8-rank schedule (V/8 rows per table), the head of a skewed vocabulary.  By Little's law ~12 MB must be in flight to stream at
// HBM rate; on a 25 MB table that is half the rows — exclusive access cannot be had, and the memory-side atomic rate (1.3 TB/s) is a
// sixth of the plain rate.  Those tables fit the 256 MB Infinity Cache, so the remedy is to stop WRITING rows per update at all:
//
//   1. every (context row, target row, label) term of the batch becomes an 8-byte ITEM (k_sorted_emit) — the same pairs, window draws
//      and negative draws as the other kernels make.  (Round 4: one packed 64-bit word, key | other row | label, sorted on the key's bits only —
//      a radix pass over 8-byte keys runs 1.3 - 1.4x a pass over (4-byte key, 8-byte value) pairs, scripts/micro/sort_keys64.hip; the learning
//      rate left the item: a synchronous mini-batch trains at the rate of its first walk);
//   2. the items are sorted by TARGET row (stable radix sort); a worker that owns a run of items of one row keeps that row in
//      registers, reads the other side (syn0[context], read-only in this phase) through the caches, applies the row's updates one
//      after the other exactly as the sequential loop would, and stores the row once (phase A).  The step g of every item is kept;
//   3. the items are sorted by CONTEXT row; the owner of a context row sums g * syn1neg[target] over its items — the target rows as
//      they stood BEFORE the mini-batch (phase A writes the moved rows to a shadow table that is committed afterwards): reading the
//      moved rows instead would feed every item its own step back (a g^2 term along the context row, always of the same sign:
//      measured, it wrecks the embedding) — and adds the sum to syn0[context] once (phase B).
// No lock, no atomic, no lost update, and the result does not depend on how many workers ran: it is a deterministic function of the
// batch, which the oracle restates bit for bit (oracle/dge_oracle.c: sorted_*), at FULL concurrency.
// What it changes against the sequential loop: within one mini-batch (DGE_TUNE_SORTED_WALKS walks) the context rows are frozen while
// the target rows move (phase A), and the context rows then take the sum of their terms (phase B) — a synchronous mini-batch, the
// bounded staleness Hogwild has anyway (DL4J itself gathers 512 pairs per thread into one libnd4j aggregate batch).
// Work units are CHUNKS of a fixed number of sorted items, not rows, so a hot row costs no more than a cold one: a row whose items
// straddle chunk borders is processed as independent segments from the same starting row and their deltas are added in chunk order
// by the row's owner at the end of the mini-batch (k_sorted_finish).
#include <hipcub/hipcub.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <string.h>

#include <algorithm>
#include <vector>

#include "sgns_kernels.h"
#include "sgns_model.h"

struct dge_sorted_work {
    int64_t cap_units = 0;                       // (walk, centre) units the count buffers hold
    int32_t* cnt = nullptr; int64_t* off = nullptr;
    void* scan_tmp = nullptr; size_t scan_tmp_bytes = 0;
    // Two buffer sets: while mini-batch k runs its phases on the model's stream (bound by the caches), mini-batch k+1 is emitted and
    // sorted by target row on a second stream (bound by table look-up latency and by HBM): set k % 2 holds it0 = emitted items,
    // later phase A's output; it1 = the sorted items of whichever phase runs
    int64_t cap_items = 0;                       // item slots of every array
    uint64_t *it0[2] = {nullptr, nullptr}, *it1[2] = {nullptr, nullptr};
    void* sort_tmp[2] = {nullptr, nullptr}; size_t sort_tmp_bytes = 0;        // [0]: second stream (sort by target), [1]: model's stream (sort by context)
    hipStream_t aux = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool set_used[2] = {false, false};
    int64_t live = 0;                            // mini-batches run so far: set live % 2 is the next one's, ACROSS launches — an episode's first mini-batch then takes the set its
                                                 // predecessor's last one did not use, and its sort (second stream) starts while that last mini-batch's phases still run
    int64_t* seg = nullptr; int64_t cap_seg = 0; // first sorted position of every row (+ end): by target for set 0, set 1; by context
    float* shadow = nullptr; int64_t cap_shadow = 0;   // the target rows as phase A leaves them, committed after phase B
    float* scratch = nullptr; float* scratch_b = nullptr; int64_t cap_scratch_rows = 0;      // the segments' deltas of phase A / of phase B (k_sorted_finish reads both)
    int64_t* d_marks = nullptr; int64_t cap_marks = 0;
    int64_t* h_marks = nullptr; int64_t cap_h_marks = 0;      // pinned: the one read-back of a launch (first pair of every mini-batch) lands here
    // Block schedule: the items of ALL n target partitions of a rank's context partition, made once per global batch (k_block_count /
    // k_block_emit) instead of once per episode: bucket t holds, in walk order, the items of the pairs whose centre row is in partition t.
    int32_t* st_cnt = nullptr; int64_t* st_off = nullptr; int64_t st_cap_cells = 0;      // pairs per (bucket, walk) cell [n x walks], exclusive prefix
    void* st_scan_tmp = nullptr; size_t st_scan_bytes = 0;
    uint64_t* st_it = nullptr; int64_t st_cap_items = 0;
    unsigned long long* st_words = nullptr;      // in-vocabulary tokens of the batch
    std::vector<int64_t> st_bucket0;             // first pair of every bucket (+ end)
    // mini-batches of every bucket, laid out when the store is made (round 5: an episode reads nothing back — the host runs ahead of the device through all n episodes of a
    // batch, and the next episode's first sort overlaps this episode's last phases and the partition's hand-off): walks per mini-batch, first pair of each mini-batch (+ end)
    std::vector<int64_t> st_walks_per; std::vector<std::vector<int64_t>> st_marks;
    hipEvent_t ev_store = nullptr;               // recorded behind k_block_emit on the model's stream: the second stream waits for it before it sorts the store's items
    bool st_valid = false;
    struct Key { const int32_t* sen; int64_t n_rows; uint64_t gen; int32_t L, W, K, part_n, part_ctx; int64_t gidx_base, V, T; uint64_t seed; int64_t walks_knob;      // (items carry no learning rate)
                 bool operator==(const Key& o) const {
                     return sen == o.sen && n_rows == o.n_rows && gen == o.gen && L == o.L && W == o.W && K == o.K && part_n == o.part_n && part_ctx == o.part_ctx &&
                            gidx_base == o.gidx_base && V == o.V && T == o.T && seed == o.seed && walks_knob == o.walks_knob;
                 } } st_key_of{};
};

void dge_sorted_release(dge_model* m) {
    dge_sorted_work* s = m->sorted;
    if (!s) return;
    if (s->aux) { (void)hipStreamSynchronize(s->aux); (void)hipStreamDestroy(s->aux); }
    for (int x = 0; x < 2; x++) {
        dge_dev_free(s->it0[x]); dge_dev_free(s->it1[x]); dge_dev_free(s->sort_tmp[x]);
        if (s->ev_ready[x]) (void)hipEventDestroy(s->ev_ready[x]);
        if (s->ev_done[x]) (void)hipEventDestroy(s->ev_done[x]);
    }
    if (s->ev_store) (void)hipEventDestroy(s->ev_store);
    dge_dev_free(s->cnt); dge_dev_free(s->off); dge_dev_free(s->scan_tmp); dge_dev_free(s->seg); dge_dev_free(s->scratch); dge_dev_free(s->scratch_b); dge_dev_free(s->d_marks); dge_dev_free(s->shadow);
    dge_dev_free(s->st_cnt); dge_dev_free(s->st_off); dge_dev_free(s->st_scan_tmp); dge_dev_free(s->st_it); dge_dev_free(s->st_words);
    if (s->h_marks) (void)hipHostFree(s->h_marks);
    delete s;
    m->sorted = nullptr;
}

struct SortedParams {
    TrainParams t;
    const int32_t* cnt; const int64_t* off;      // pairs per (walk, centre) unit, exclusive prefix
    int64_t unit0, unit1;                        // units of this mini-batch
    int64_t pair0;                               // off[unit0]
    // An item is ONE 64-bit word.  Sorted by target (emit -> phase A): key << ks1 | other row << 1 | label (1 = a negative); sorted by context (phase A ->
    // phase B): key << ks2 | target row << 11 | step code (label << 10 | the sigmoid table's index, 1000 / 1001 = saturated): the step is (label - sigma) x the
    // mini-batch's learning rate on both sides, so the code carries it exactly.  The sorts look at the key's bits only (stable: the rest rides along).
    uint64_t* it_out; const uint64_t* it_in;
    int32_t ks1, ks2; uint32_t omask;            // ks1 = 1 + bits of a row number, ks2 = 11 + that; omask = (1 << bits) - 1
    int64_t mb_walk0;                            // first walk of the mini-batch: the learning rate of all its items
    int32_t run_nb;                              // k_sorted_emit<.., RUNS>: boundaries of the table's run form that are searched (a power of two)
    const int64_t* seg;                          // seg[k] = first sorted position with key >= k; seg[Vk] = valid items
    int64_t n_slots;                             // sorted array length (valid items first, then the skipped draws with key V)
    int32_t chunk;                               // items per work unit
    float* scratch;                              // [2 * chunks][stride]: deltas of the rows a chunk shares with its neighbours (the running phase's: A's or B's)
    float* scratch_a; float* scratch_b;          // k_sorted_finish: both phases' deltas (phase A's must outlive phase B)
    const int64_t* seg_tgt;                      // k_sorted_finish: the target-sorted segments (seg is then the context-sorted ones)
    float* shadow;                               // phase A writes the moved target rows here; phase B still reads the rows as they were
    // sort keys: row / kdiv.  Under the block schedule only rows = part (mod part_n) occur on either side, so the keys lose log2(part_n) bits
    // (the 125 k live rows of an 8-rank block: 17 bits instead of 20, two sort passes instead of three); row = key * kdiv + the side's part.
    int32_t kdiv, kpart_tgt, kpart_ctx;
    int32_t Vk;                                  // keys are 0 .. Vk - 1; Vk = a skipped draw (sorts behind every row)
};

// the (walk, centre) unit's window: contexts c in [lo, hi] without i
__device__ __forceinline__ void unit_window(const TrainParams& p, int64_t w, int i, int len, uint64_t& s, int& lo, int& hi) {
    const int64_t gbase = (p.gidx_base + w) * (int64_t)p.L;
    s = dge_mix64(p.seed + (uint64_t)(gbase + i));
    s = s * DGE_W2V_MULT + 11;
    const int radius = p.W - (int)dge_fast_mod(s, (uint64_t)p.W, p.W_magic);
    lo = max(0, i - radius);
    hi = min(len - 1, i + radius);
}

// the learning rate of a synchronous mini-batch: that of its first walk (the exact count of in-vocabulary tokens that precede it)
__device__ __forceinline__ float mb_alpha(const TrainParams& p, int64_t w0) {
    const int64_t wbw = p.wb[w0];
    const int64_t done = p.words_done_base + (p.words_scale == 1.0 ? wbw : (int64_t)((double)wbw * p.words_scale));
    float alpha = (float)((double)p.alpha0 * (1.0 - (double)done / (double)(p.all_words + 1)));
    return alpha < p.min_alpha ? p.min_alpha : alpha;
}
// the step of one term as a code (the sigmoid table's index, 1000: f beyond +MAX_EXP, 1001: beyond -MAX_EXP) and back: (label - sigma) * alpha, word2vec.c's expression
__device__ __forceinline__ int step_code(float f) {
    if (f > (float)MAX_EXP) return 1000;
    if (f < -(float)MAX_EXP) return 1001;
    const int idx = (int)((f + (float)MAX_EXP) * (float)(EXP_TABLE_SIZE / MAX_EXP / 2));
    return min(max(idx, 0), EXP_TABLE_SIZE - 1);
}
__device__ __forceinline__ float code_step(int code, float label, float alpha, const float* s_exp) {
    if (code == 1000) return (label - 1.0f) * alpha;
    if (code == 1001) return (label - 0.0f) * alpha;
    return (label - s_exp[code]) * alpha;
}

// totals of a launch's pairs and words: one atomic per WORKGROUP and counter, from launches of at most COUNT_BLOCKS workgroups — atomics on one address complete one
// per ~12 ns at the memory side, and a wave's worth each (12 500 x 2 on a cfg2 step) made the count kernel 300 us long (round 4)
#define COUNT_BLOCKS 256      /* the count kernels' grids: at most 2 x this */
__device__ __forceinline__ void count_tally(long long pn, long long words, unsigned long long* pairs_out, unsigned long long* words_out) {
    __shared__ long long s_t[2][4];
    for (int o = 32; o > 0; o >>= 1) { pn += __shfl_xor(pn, o); words += __shfl_xor(words, o); }
    if ((threadIdx.x & 63) == 0) { s_t[0][threadIdx.x >> 6] = pn; s_t[1][threadIdx.x >> 6] = words; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        long long a = 0, b = 0;
        for (int i = 0; i < nw; i++) { a += s_t[0][i]; b += s_t[1][i]; }
        if (a) atomicAdd(pairs_out, (unsigned long long)a);
        if (b) atomicAdd(words_out, (unsigned long long)b);
    }
}

// pairs of every (walk, centre) unit; totals of pairs and words for dge_model_stats
__global__ void __launch_bounds__(256) k_sorted_count(TrainParams p, int32_t* cnt) {
    const int64_t n_units = p.n_rows * (int64_t)p.L;
    long long pn = 0, words = 0;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += (int64_t)gridDim.x * blockDim.x) {
        int n = 0;
        const int64_t w = u / p.L; const int i = (int)(u % p.L);
        const int len = (int)p.len[w];
        const int32_t* sen = p.sen + w * p.L;
        if (i == 0 && (p.part_n <= 1 || p.part_ctx == p.part_tgt)) words += len;
        if (i < len && (p.part_n <= 1 || sen[i] % p.part_n == p.part_tgt)) {
            uint64_t s; int lo, hi;
            unit_window(p, w, i, len, s, lo, hi);
            if (p.part_n <= 1) n = (hi - lo + 1) - 1;               // every token of the compacted walk is a context, but the centre itself
            else for (int c = lo; c <= hi; c++) n += (c != i && sen[c] % p.part_n == p.part_ctx) ? 1 : 0;
        }
        cnt[u] = n;
        pn += n;
    }
    count_tally(pn, words, &p.counters[0], &p.counters[1]);
}

// the same counts with one thread per WALK (walks of up to 64 tokens: the partition tests become two bit masks over the walk's
// tokens, a unit's pair count a popcount) — the block schedule visits every walk of the global batch for a few pairs each
__global__ void __launch_bounds__(256) k_sorted_count_walks(TrainParams p, int32_t* cnt) {
    long long pn = 0, words = 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < p.n_rows; w += (int64_t)gridDim.x * blockDim.x) {
        const int len = (int)p.len[w];
        const int32_t* sen = p.sen + w * p.L;
        uint64_t ctx_mask = 0, tgt_mask = 0;
        if (p.part_n <= 1) ctx_mask = tgt_mask = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
        else for (int j = 0; j < len; j++) {
            const int r = sen[j] % p.part_n;
            ctx_mask |= (uint64_t)(r == p.part_ctx) << j; tgt_mask |= (uint64_t)(r == p.part_tgt) << j;
        }
        if (p.part_n <= 1 || p.part_ctx == p.part_tgt) words += len;
        for (int i = 0; i < p.L; i++) {
            int n = 0;
            if (i < len && ((tgt_mask >> i) & 1ull)) {
                uint64_t s; int lo, hi;
                unit_window(p, w, i, len, s, lo, hi);
                uint64_t wm = (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & (~0ull << lo) & ~(1ull << i);
                n = __popcll(ctx_mask & wm);
            }
            cnt[w * p.L + i] = n;
            pn += n;
        }
    }
    count_tally(pn, words, &p.counters[0], &p.counters[1]);
}

__global__ void k_sorted_marks(const int64_t* off, const int64_t* units, int n, int64_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = off[units[i]];
}
// first pair of every mini-batch of `walks_per` walks: off[base + min(k * walks_per, n_rows) * mult], k = 0 .. n - 1 (no marks travel from the host)
__global__ void k_sorted_marks_walks(const int64_t* off, int64_t base, int64_t mult, int64_t walks_per, int64_t n_rows, int n, int64_t* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = off[base + min((int64_t)k * walks_per, n_rows) * mult];
}

// items of one mini-batch: slot (pair - pair0) * (K+1) + d holds term d of the pair (d = 0: the centre, label 1; d >= 1: negative d).
// A negative that drew the centre itself is not trained (word2vec): its slot gets the key V and sorts behind every row.
// One 16-lane group per 16 (walk, centre) units; lane j draws negative j of a pair (the draws of k_sgns_train, stream for stream).
// the table's run form in LDS and a draw's table slot -> row through it (k_sorted_emit, k_block_emit; RUNS, p in scope)
#define EMIT_RUNS_LDS(run_nb_) \
    __shared__ double s_run_base[RUNS ? DGE_RUN_MAX : 1]; \
    __shared__ uint32_t s_run_row[RUNS ? DGE_RUN_MAX + 1 : 1]; \
    __shared__ uint32_t s_exc_slot[RUNS ? DGE_RUN_EXC : 1]; \
    __shared__ int32_t s_exc_row[RUNS ? DGE_RUN_EXC : 1]; \
    if (RUNS) { \
        for (int i = threadIdx.x; i < (run_nb_); i += blockDim.x) s_run_base[RUNS ? i : 0] = p.run_base[i]; \
        for (int i = threadIdx.x; i < (run_nb_) + 1; i += blockDim.x) s_run_row[RUNS ? i : 0] = p.run_row[i]; \
        for (int i = threadIdx.x; i < DGE_RUN_EXC; i += blockDim.x) { s_exc_slot[RUNS ? i : 0] = p.exc_slot[i]; s_exc_row[RUNS ? i : 0] = p.exc_row[i]; } \
        __syncthreads(); \
    }
#define EMIT_DRAW_ROW(run_nb_) \
    auto draw_row = [&](uint64_t slot) -> int32_t { \
        if (RUNS) { \
            const uint32_t a = (uint32_t)slot; \
            int32_t r = -2; \
            const double x = (double)(a - 1u) * p.T_inv; \
            if (a != 0u && s_run_base[0] < x) { \
                int lo = 0; \
                for (int st = (run_nb_) >> 1; st >= 1; st >>= 1) if (s_run_base[RUNS ? lo + st : 0] < x) lo += st; \
                const int64_t n = (int64_t)s_run_row[RUNS ? lo + 1 : 0] - (int64_t)s_run_row[RUNS ? lo : 0]; \
                int64_t k = 0; \
                if (n > 0) { \
                    const double qq = (x - s_run_base[RUNS ? lo : 0]) * (double)n / (s_run_base[RUNS ? lo + 1 : 0] - s_run_base[RUNS ? lo : 0]); \
                    k = (int64_t)ceil(qq) - 1; \
                    k = k < 0 ? 0 : (k > n ? n : k); \
                } \
                r = (int32_t)min((int64_t)s_run_row[RUNS ? lo : 0] + k, p.V - 1); \
                if (p.n_exc > 0) { \
                    int e = 0; \
                    for (int st = DGE_RUN_EXC / 2; st >= 1; st >>= 1) if (e + st < p.n_exc && s_exc_slot[RUNS ? e + st : 0] <= a) e += st; \
                    if (s_exc_slot[RUNS ? e : 0] == a) r = s_exc_row[RUNS ? e : 0]; \
                } \
            } \
            if (r != -2) return r; \
        } \
        return neg_table_row(p.ctab, slot); \
    };
// UPG = units a group takes: 16 under the block schedule (most units have no pair of the block), fewer otherwise.
// RUNS (round 4): the negatives' rows from the table's RUN form in LDS (neg_row_by_runs: a binary search over the run boundaries and a few f64 operations) where the
// model has one, instead of one 16-byte look-up per draw in the 16.7 MB rank-block table: those look-ups — 10.7 M random requests a cfg2 mini-batch — were what the
// kernel ran against (173 us whatever UPG; the fabric's request rate).  Only the first q.run_nb boundaries are searched (a power of two above the
// number of runs; the rest of base[] is +inf anyway).  Same rows, bit for bit: the run form was checked against the table slot by slot when the model was made.
template <int UPG, bool RUNS>
__global__ void __launch_bounds__(256) k_sorted_emit(SortedParams q) {
    const TrainParams& p = q.t;
    EMIT_RUNS_LDS(q.run_nb)
    const int lane = threadIdx.x & 15;
    const int sh = threadIdx.x & 48;
    // a group takes UPG consecutive units and works through those that have pairs
    const int64_t u0 = q.unit0 + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4) * UPG;
    if (u0 >= q.unit1) return;
    const int my_cnt = (lane < UPG && u0 + lane < q.unit1) ? q.cnt[u0 + lane] : 0;
    unsigned todo = (unsigned)(__ballot(my_cnt > 0) >> sh) & 0xFFFFu;
    if (!todo) return;
    // every lane prepares ITS unit (one round of loads for the whole group), the group then walks through the units' pairs
    int64_t my_slot = 0; uint64_t my_s = 0; int my_lo = 0, my_hi = -1; int32_t my_word = -1;
    if (my_cnt > 0) {
        const int64_t u = u0 + lane;
        const int64_t w = u / p.L; const int i = (int)(u % p.L);
        const int len = (int)p.len[w];
        my_word = p.sen[w * p.L + i];
        unit_window(p, w, i, len, my_s, my_lo, my_hi);
        my_slot = (q.off[u] - q.pair0) * (int64_t)(p.K + 1);
    }
    EMIT_DRAW_ROW(q.run_nb)
    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= lane; j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }
    const int K = p.K;
    // K <= 16: a TRIP takes up to P = 16 / K pairs of the centre at once — lane l = z * K + d draws negative d of the trip's pair z, lanes
    // 0 .. pairs - 1 write the positives: one round of table look-ups and four store instructions per trip instead of four per pair.
    // Streams: one stream per centre (lane l stands l + 1 steps behind the trip's start, the next trip starts where lane pairs * K - 1 stood);
    // under the block schedule one stream per pair (lane l stands d + 1 steps behind its pair's start).
    const int P = K == 0 ? 16 : max(1, 16 / max(K, 1));
    const int z_l = K ? lane / K : lane, d_l = K ? lane - z_l * K : 0;
    uint64_t mD = 1, cD = 0;
    for (int j = 0; j <= d_l; j++) { mD *= DGE_W2V_MULT; cD = cD * DGE_W2V_MULT + 11; }
    while (todo) {
        const int ul = __builtin_ctz(todo); todo &= todo - 1;
        const int64_t u = u0 + ul;
        const int64_t w = u / p.L; const int i = (int)(u % p.L);
        const int32_t* sen = p.sen + w * p.L;
        const int32_t word = __shfl(my_word, ul, 16);
        const int lo = __shfl(my_lo, ul, 16), hi = __shfl(my_hi, ul, 16);
        uint64_t s = shfl16_u64(my_s, ul);
        const uint64_t s_centre = s;
        int64_t slot = (int64_t)shfl16_u64((uint64_t)my_slot, ul);
        for (int c0 = lo; c0 <= hi; c0 += 16) {
            // 16 candidate contexts at a time: lane j tests position c0 + j
            const int cj = c0 + lane;
            int32_t tok = -1;
            if (cj <= hi && cj != i) { tok = sen[cj]; if (p.part_n > 1 && tok % p.part_n != p.part_ctx) tok = -1; }
            unsigned live = (unsigned)(__ballot(tok >= 0) >> sh) & 0xFFFFu;
            while (live) {
                if (K <= 16) {
                    const int npair = min(P, __popc(live));
                    int cl_mine = 0, cl_pos = 0;                 // candidate lane of the pair my negative belongs to / of pair `lane`
                    for (int z = 0; z < npair; z++) {
                        const int cl = __builtin_ctz(live); live &= live - 1;
                        if (z == z_l) cl_mine = cl;
                        if (z == lane) cl_pos = cl;
                    }
                    const int32_t last_mine = __shfl(tok, cl_mine, 16), last_pos = __shfl(tok, cl_pos, 16);
                    const bool neg_on = K > 0 && lane < npair * K;
                    const uint64_t sl = p.part_n > 1 ? dge_mix64(s_centre + (uint64_t)(c0 + cl_mine)) * mD + cD : s * mA + cA;
                    if (neg_on) {
                        int32_t t = draw_row(dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                        if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                        if (p.part_n > 1) t = part_row(t, p.part_n, p.part_tgt, p.V);
                        const int64_t at = slot + (int64_t)z_l * (K + 1) + 1 + d_l;
                        q.it_out[at] = ((uint64_t)(uint32_t)(t == word ? q.Vk : t / q.kdiv) << q.ks1) | ((uint64_t)(uint32_t)last_mine << 1) | 1ull;      // bit 0: a negative
                    }
                    if (lane < npair) {
                        const int64_t at = slot + (int64_t)lane * (K + 1);
                        q.it_out[at] = ((uint64_t)(uint32_t)(word / q.kdiv) << q.ks1) | ((uint64_t)(uint32_t)last_pos << 1);
                    }
                    if (K > 0 && p.part_n <= 1) s = shfl16_u64(sl, npair * K - 1);
                    slot += (int64_t)npair * (K + 1);
                } else {                                         // more negatives than lanes: one pair per trip, 16 draws at a time
                    const int cl = __builtin_ctz(live); live &= live - 1;
                    const int32_t lastv = __shfl(tok, cl, 16);
                    if (p.part_n > 1) s = dge_mix64(s_centre + (uint64_t)(c0 + cl));
                    const uint64_t oth = (uint64_t)(uint32_t)lastv << 1;
                    if (lane == 0) q.it_out[slot] = ((uint64_t)(uint32_t)(word / q.kdiv) << q.ks1) | oth;
                    for (int kd = 0; kd < K; kd += 16) {
                        const int kc = min(16, K - kd);
                        const uint64_t sl = s * mA + cA;
                        if (lane < kc) {
                            int32_t t = draw_row(dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                            if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                            if (p.part_n > 1) t = part_row(t, p.part_n, p.part_tgt, p.V);
                            q.it_out[slot + 1 + kd + lane] = ((uint64_t)(uint32_t)(t == word ? q.Vk : t / q.kdiv) << q.ks1) | oth | 1ull;
                        }
                        s = shfl16_u64(sl, kc - 1);
                    }
                    slot += K + 1;
                }
            }
        }
    }
}

// ---- Block schedule, items made once per global batch.  An episode of an n-rank schedule trains the pairs (context row in partition part_ctx,
// centre row in partition t); run per episode, the count and emit kernels above scan all n x B walks of the global batch n times for 1/n of
// a rank's pairs each time (profiles/r02_sim8_kernel_stats.csv: k_sorted_emit 24 % of an 8-rank step).  Instead, once per batch: every walk's
// pairs with context in part_ctx, counted per centre partition (k_block_count), one prefix sum over the [n x walks] cells — bucket after
// bucket, so every bucket's items are one contiguous run in walk order — and one emit pass that writes each pair's items into its bucket
// (k_block_emit: one 16-lane group per walk, tokens in registers).  Episode (part_ctx, t) then sorts and trains slices of bucket t: the same
// items in the same order as the per-episode kernels produce (bit-exact tests: tests/test_gpu_sorted.py).
__global__ void __launch_bounds__(256) k_block_count(TrainParams p, int32_t* cells, unsigned long long* words_out) {
    long long words = 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < p.n_rows; w += (int64_t)gridDim.x * blockDim.x) {
        const int len = (int)p.len[w];
        const int32_t* sen = p.sen + w * p.L;
        uint64_t ctx_mask = 0;
        for (int j = 0; j < len; j++) { const int32_t v = sen[j]; ctx_mask |= (uint64_t)(v - dge_fast_div32(v, p.part_n, p.N_magic) * p.part_n == p.part_ctx) << j; }
        words += len;
        int cnt[16];
#pragma unroll
        for (int t = 0; t < 16; t++) cnt[t] = 0;
        for (int i = 0; i < len; i++) {
            uint64_t s; int lo, hi;
            unit_window(p, w, i, len, s, lo, hi);
            const uint64_t wm = (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & (~0ull << lo) & ~(1ull << i);
            const int n = __popcll(ctx_mask & wm);
            const int32_t wv = sen[i]; const int t = wv - dge_fast_div32(wv, p.part_n, p.N_magic) * p.part_n;
#pragma unroll
            for (int z = 0; z < 16; z++) cnt[z] += z == t ? n : 0;
        }
#pragma unroll
        for (int t = 0; t < 16; t++) if (t < p.part_n) cells[(int64_t)t * p.n_rows + w] = cnt[t];
    }
    count_tally(0, words, words_out, words_out);
}

template <bool RUNS>
__global__ void __launch_bounds__(256) k_block_emit(TrainParams p, const int64_t* __restrict__ cell_off, uint64_t* __restrict__ it_out, int32_t Vk, int32_t ks1, int32_t run_nb) {
    EMIT_RUNS_LDS(run_nb)
    EMIT_DRAW_ROW(run_nb)
    const int lane = threadIdx.x & 15;
    const int sh = threadIdx.x & 48;
    const int L = p.L, K = p.K, N = p.part_n;
    uint64_t mA = 1, cA = 0;
    for (int j = 0; j <= lane; j++) { mA *= DGE_W2V_MULT; cA = cA * DGE_W2V_MULT + 11; }
    // K <= 16: up to P = 16 / K pairs of a centre per trip (k_sorted_emit): lane l = z * K + d draws negative d of the trip's pair z from that
    // pair's own stream, lanes 0 .. pairs - 1 write the positives
    const int P = K == 0 ? 16 : max(1, 16 / max(K, 1));
    const int z_l = K ? lane / K : lane, d_l = K ? lane - z_l * K : 0;
    uint64_t mD = 1, cD = 0;
    for (int j = 0; j <= d_l; j++) { mD *= DGE_W2V_MULT; cD = cD * DGE_W2V_MULT + 11; }
    // token at walk position c, c different in every lane: lane c & 15 holds it in register c >> 4
#define BLOCK_TOK(c_) ({ const int c__ = (c_); const int32_t a0 = __shfl(tk0, c__ & 15, 16), a1 = __shfl(tk1, c__ & 15, 16), a2 = __shfl(tk2, c__ & 15, 16), a3 = __shfl(tk3, c__ & 15, 16); \
                         (c__ >> 4) == 0 ? a0 : ((c__ >> 4) == 1 ? a1 : ((c__ >> 4) == 2 ? a2 : a3)); })
    // a group takes walk after walk (with the run form the grid is sized to fill the device, not to the batch: its LDS arrays are filled once per workgroup)
    for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; w < p.n_rows; w += ((int64_t)gridDim.x * blockDim.x) >> 4) {
    const int len = (int)p.len[w];
    if (len <= 1) continue;
    const int32_t* sen = p.sen + w * L;
    const int32_t tk0 = lane < len ? sen[lane] : -1, tk1 = lane + 16 < len ? sen[lane + 16] : -1;
    const int32_t tk2 = lane + 32 < len ? sen[lane + 32] : -1, tk3 = lane + 48 < len ? sen[lane + 48] : -1;
    const uint64_t ctx_mask = part_token_mask(tk0, tk1, tk2, tk3, N, p.part_ctx);
    if (!ctx_mask) continue;
    // lane t keeps the next free pair position of bucket t
    int64_t my_pos = lane < N ? cell_off[(int64_t)lane * p.n_rows + w] : 0;
    for (int i = 0; i < len; i++) {
        uint64_t s; int lo, hi;
        unit_window(p, w, i, len, s, lo, hi);
        uint64_t pm = ctx_mask & (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & (~0ull << lo) & ~(1ull << i);
        if (!pm) continue;
        const int32_t word = walk_tok(true, sen, i, tk0, tk1, tk2, tk3);
        const int32_t wkey = dge_fast_div32(word, N, p.N_magic);
        const int bucket = word - wkey * N;
        const uint64_t s_centre = s;
        int64_t slot = (int64_t)shfl16_u64((uint64_t)my_pos, bucket) * (int64_t)(K + 1);
        const int np = __popcll(pm);
        if (lane == bucket) my_pos += np;
        while (pm) {
            if (K <= 16) {
                const int npair = min(P, __popcll(pm));
                int c_mine = 0, c_pos = 0;                   // context position of the pair my negative belongs to / of pair `lane`
                for (int z = 0; z < npair; z++) {
                    const int c = __builtin_ctzll(pm); pm &= pm - 1ull;
                    if (z == z_l) c_mine = c;
                    if (z == lane) c_pos = c;
                }
                const int32_t last_pos = BLOCK_TOK(c_pos);
                const int32_t last_mine = __shfl(last_pos, z_l, 16);
                if (K > 0 && lane < npair * K) {
                    const uint64_t sl = dge_mix64(s_centre + (uint64_t)c_mine) * mD + cD;       // every pair draws from its own stream
                    int32_t t = draw_row(dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                    if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                    int32_t tk = dge_fast_div32(t, N, p.N_magic);              // part_row: the row of the bucket's partition nearest below the draw
                    if ((int64_t)tk * N + bucket >= p.V) tk--;
                    const int64_t at = slot + (int64_t)z_l * (K + 1) + 1 + d_l;
                    it_out[at] = ((uint64_t)(uint32_t)(tk == wkey ? Vk : tk) << ks1) | ((uint64_t)(uint32_t)last_mine << 1) | 1ull;      // bit 0: a negative
                }
                if (lane < npair) {
                    const int64_t at = slot + (int64_t)lane * (K + 1);
                    it_out[at] = ((uint64_t)(uint32_t)wkey << ks1) | ((uint64_t)(uint32_t)last_pos << 1);
                }
                slot += (int64_t)npair * (K + 1);
            } else {                                         // more negatives than lanes: one pair per trip, 16 draws at a time
                const int c = __builtin_ctzll(pm); pm &= pm - 1ull;
                const int32_t lastv = walk_tok(true, sen, c, tk0, tk1, tk2, tk3);
                const uint64_t oth = (uint64_t)(uint32_t)lastv << 1;
                if (lane == 0) it_out[slot] = ((uint64_t)(uint32_t)wkey << ks1) | oth;
                uint64_t sp = dge_mix64(s_centre + (uint64_t)c);
                for (int kd = 0; kd < K; kd += 16) {
                    const int kc = min(16, K - kd);
                    const uint64_t sl = sp * mA + cA;
                    if (lane < kc) {
                        int32_t t = draw_row(dge_fast_mod(sl >> 16, (uint64_t)p.T, p.T_magic));
                        if (t == 0 && p.V > 1) t = (int32_t)(sl % (uint64_t)(p.V - 1)) + 1;
                        int32_t tk = dge_fast_div32(t, N, p.N_magic);
                        if ((int64_t)tk * N + bucket >= p.V) tk--;
                        it_out[slot + 1 + kd + lane] = ((uint64_t)(uint32_t)(tk == wkey ? Vk : tk) << ks1) | oth | 1ull;
                    }
                    sp = shfl16_u64(sl, kc - 1);
                }
                slot += K + 1;
            }
        }
    }
    }
#undef BLOCK_TOK
    (void)sh;
}
__global__ void k_block_tally(unsigned long long* counters, unsigned long long pairs, const unsigned long long* words, int add_words) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (pairs) atomicAdd(&counters[0], pairs);
        if (add_words && *words) atomicAdd(&counters[1], *words);
    }
}

// seg[k] = first position of the sorted keys that is >= k, k = 0 .. Vk+1  (seg[Vk] = number of valid items, seg[Vk+1] = n)
__global__ void k_sorted_segments(const uint64_t* __restrict__ items, int ks, int64_t n, int64_t V, int64_t* seg) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > V + 1) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)(items[mid] >> ks) < r) lo = mid + 1; else hi = mid; }
    seg[r] = lo;
}


// One worker (16 lanes) per chunk of sorted items.  PB == false (phase A): key = target row (owned, syn1neg), the item carries the context row and
// the label; the row takes its updates in item order, the step of every item goes out as a code with (context, target) for phase B.
// PB == true (phase B): key = context row (owned, syn0), the item carries the target row and the step's code; the row takes the sum of g * syn1neg[target].
// A row that lies wholly inside the chunk is stored directly; a segment of a row shared with a neighbouring chunk leaves its DELTA in
// scratch slot 2*chunk (the segment starts the chunk) or 2*chunk+1 (it ends the chunk), summed up by k_sorted_finish.
#define SORTED_PIPE 4      /* rows of the other side in flight per worker (8 measured the same: cfg2 3.5 ms per launch either way, an 8-rank block 62.5 against 62.7 ms) */
template <int DCH, bool PB>
__global__ void __launch_bounds__(256) k_sorted_phase(SortedParams q) {
    const TrainParams& p = q.t;
    __shared__ float s_exp[EXP_TABLE_SIZE];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) s_exp[i] = p.exp_table[i];
    __syncthreads();
    const int lane = threadIdx.x & 15;
    const int64_t chunk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t start = chunk * q.chunk;
    if (start >= q.n_slots) return;
    const int64_t end = min(start + (int64_t)q.chunk, q.n_slots);
    const int64_t n_valid = q.seg[q.Vk];
    if (!PB)            // the slots behind the valid items (skipped draws) stay invalid for the second sort
        for (int64_t x = max(start, n_valid) + lane; x < end; x += 16) q.it_out[x] = (uint64_t)(uint32_t)q.Vk << q.ks2;
    if (start >= n_valid) return;
    const int64_t stop = min(end, n_valid);
    const float alpha = mb_alpha(p, q.mb_walk0);   // a synchronous mini-batch trains at the learning rate of its first walk

    const TableView own = make_view(PB ? p.syn0 : p.syn1neg, p.V, p.stride);
    const TableView oth = make_view(PB ? p.syn1neg : p.syn0, p.V, p.stride);
    const TableView scr = make_view(q.scratch, 2 * ((q.n_slots + q.chunk - 1) / q.chunk), p.stride);
    const TableView shd = make_view(q.shadow, q.Vk, p.stride);            // (indexed by key: only the rows of the target partition exist in it)

    const int32_t kpart = PB ? q.kpart_ctx : q.kpart_tgt;
    const int ks = PB ? q.ks2 : q.ks1, osh = PB ? 11 : 1;
#define OWN_ROW(k_) ((k_) * q.kdiv + kpart)
    int32_t cur = -1;                              // key of the open segment
    int64_t seg_start = start;
    Row<DCH> h, d;                                 // the owned row as it moves (phase A), the delta of the open segment
    row_zero(h); row_zero(d);

    // close the open segment [seg_start, seg_end) of row `cur`
#define SORTED_CLOSE(seg_end)                                                                                          \
    do {                                                                                                               \
        if (cur >= 0) {                                                                                                \
            const bool whole = seg_start == q.seg[cur] && (seg_end) == q.seg[cur + 1];                                 \
            if (whole) {                                                                                               \
                if (PB) {                                                                                              \
                    Row<DCH> r0;                                                                                       \
                    rowA_load<DCH, 0, false>(r0, own, OWN_ROW(cur), lane);                                             \
                    _Pragma("unroll") for (int c_ = 0; c_ < DCH; c_++) {                                               \
                        r0.v[c_].x += d.v[c_].x; r0.v[c_].y += d.v[c_].y; r0.v[c_].z += d.v[c_].z; r0.v[c_].w += d.v[c_].w; \
                    }                                                                                                  \
                    rowA_store<DCH, 0, false>(r0, own, OWN_ROW(cur), lane);                                            \
                } else rowA_store<DCH, 0, false>(h, shd, cur, lane);                                                   \
            } else rowA_store<DCH, 0, false>(d, scr, (int32_t)(2 * chunk + (seg_start == start ? 0 : 1)), lane);      \
        }                                                                                                              \
    } while (0)

    for (int64_t base = start; base < stop; base += 16) {
        // 16 items at a time: lane l holds item base + l
        const int64_t mine = base + lane;
        uint64_t it_l = 0;
        // (items are read once and written once: non-temporal, not to displace table rows in the caches — an 8-rank block 61.9 against 62.4 ms per
        //  episode, cfg2 3.4 against 3.5 ms; the same hint on the item GENERATORS' stores cost 3 %: the sort that follows wants them cached)
        if (mine < stop) it_l = __builtin_nontemporal_load(q.it_in + mine);
        const int32_t k_l = mine < stop ? (int32_t)(it_l >> ks) : -1;
        const uint32_t o_l = (uint32_t)(it_l >> osh) & q.omask;             // the other side's row
        const uint32_t c_l = PB ? (uint32_t)it_l & 0x7FFu : (uint32_t)it_l & 1u;      // phase B: the step's code; phase A: the label bit
        const int nb = (int)min((int64_t)16, stop - base);
        int out_c = 0;                              // phase A: the code of the step of the item this lane holds
        for (int g0 = 0; g0 < nb; g0 += SORTED_PIPE) {
            Row<DCH> o[SORTED_PIPE];
            int32_t key[SORTED_PIPE]; uint32_t oc[SORTED_PIPE];
#pragma unroll
            for (int z = 0; z < SORTED_PIPE; z++) {
                const int src = min(g0 + z, nb - 1);
                key[z] = __shfl(k_l, src, 16);
                oc[z] = (uint32_t)__shfl((int)c_l, src, 16);
                rowA_load<DCH, 0, false>(o[z], oth, (int32_t)(uint32_t)__shfl((int)o_l, src, 16), lane);
            }
#pragma unroll
            for (int z = 0; z < SORTED_PIPE; z++) {
                if (g0 + z >= nb) break;
                const int64_t idx = base + g0 + z;
                if (key[z] != cur) {
                    SORTED_CLOSE(idx);
                    cur = key[z]; seg_start = idx;
                    row_zero(d);
                    if (!PB) rowA_load<DCH, 0, false>(h, own, OWN_ROW(cur), lane);
                }
                if (PB) {
                    row_axpy(d, code_step((int)(oc[z] & 0x3FFu), (oc[z] >> 10) ? 0.0f : 1.0f, alpha, s_exp), o[z]);
                } else {
                    const float label = oc[z] ? 0.0f : 1.0f;
                    const int code = step_code(row_dot(h, o[z]));
                    const float g = code_step(code, label, alpha, s_exp);
                    row_axpy(h, g, o[z]);
                    row_axpy(d, g, o[z]);
                    if (lane == g0 + z) out_c = code | (int)(oc[z] << 10);         // (the item's record for the second sort leaves with the other 15 of its batch, below)
                }
            }
        }
        // phase A: context key | target row | step code of the 16 items, one coalesced store
        if (!PB && mine < stop)
            __builtin_nontemporal_store(((uint64_t)(o_l / (uint32_t)q.kdiv) << q.ks2) | ((uint64_t)(uint32_t)OWN_ROW(k_l) << 11) | (uint64_t)(uint32_t)out_c, q.it_out + mine);
    }
    SORTED_CLOSE(stop);
#undef SORTED_CLOSE
#undef OWN_ROW
}

// End of a mini-batch, one 16-lane group per key (round 5: one kernel where there were three — a fix-up pass behind each phase and the commit; each was a launch of
// tens of microseconds that mostly found nothing to do, 5 % of an 8-rank episode at the 1 M-walk global batch):
//   target side (syn1neg, segments sorted by target): a row that lay wholly inside one chunk was left by phase A in the shadow table and is committed from there; a row
//     whose items straddle chunks takes the deltas of its segments, in chunk order, on top of the row as it stood before the mini-batch (what phase B read);
//   context side (syn0, segments sorted by context): whole rows were updated by phase B itself; a straddling row takes its segments' deltas in chunk order.
// A segment's delta is in scratch slot 2 * chunk (the segment starts the chunk) or 2 * chunk + 1 (it begins inside it): k_sorted_phase.
template <int DCH>
__global__ void __launch_bounds__(256) k_sorted_finish(SortedParams q) {
    const TrainParams& p = q.t;
    const int lane = threadIdx.x & 15;
    const int64_t kr = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (kr >= q.Vk) return;
    const int64_t n_chunks = (q.n_slots + q.chunk - 1) / q.chunk;
    {   // target side
        const int64_t r0 = q.seg_tgt[kr], r1 = q.seg_tgt[kr + 1];
        if (r1 > r0) {
            const int64_t c0 = r0 / q.chunk, c1 = (r1 - 1) / q.chunk;
            const TableView own = make_view(p.syn1neg, p.V, p.stride);
            const int32_t r = (int32_t)(kr * q.kdiv + q.kpart_tgt);
            Row<DCH> row;
            if (c0 == c1) rowA_load<DCH, 0, false>(row, make_view(q.shadow, q.Vk, p.stride), (int32_t)kr, lane);
            else {
                const TableView scr = make_view(q.scratch_a, 2 * n_chunks, p.stride);
                rowA_load<DCH, 0, false>(row, own, r, lane);
                for (int64_t c = c0; c <= c1; c++) {
                    Row<DCH> dd;
                    rowA_load<DCH, 0, false>(dd, scr, (int32_t)(2 * c + (r0 <= c * q.chunk ? 0 : 1)), lane);
#pragma unroll
                    for (int x = 0; x < DCH; x++) { row.v[x].x += dd.v[x].x; row.v[x].y += dd.v[x].y; row.v[x].z += dd.v[x].z; row.v[x].w += dd.v[x].w; }
                }
            }
            rowA_store<DCH, 0, false>(row, own, r, lane);
        }
    }
    {   // context side
        const int64_t r0 = q.seg[kr], r1 = q.seg[kr + 1];
        if (r1 > r0) {
            const int64_t c0 = r0 / q.chunk, c1 = (r1 - 1) / q.chunk;
            if (c0 != c1) {
                const TableView own = make_view(p.syn0, p.V, p.stride);
                const TableView scr = make_view(q.scratch_b, 2 * n_chunks, p.stride);
                const int32_t r = (int32_t)(kr * q.kdiv + q.kpart_ctx);
                Row<DCH> row;
                rowA_load<DCH, 0, false>(row, own, r, lane);
                for (int64_t c = c0; c <= c1; c++) {
                    Row<DCH> dd;
                    rowA_load<DCH, 0, false>(dd, scr, (int32_t)(2 * c + (r0 <= c * q.chunk ? 0 : 1)), lane);
#pragma unroll
                    for (int x = 0; x < DCH; x++) { row.v[x].x += dd.v[x].x; row.v[x].y += dd.v[x].y; row.v[x].z += dd.v[x].z; row.v[x].w += dd.v[x].w; }
                }
                rowA_store<DCH, 0, false>(row, own, r, lane);
            }
        }
    }
}
template <int DCH>
static void launch_finish(const SortedParams& q, hipStream_t st) {
    hipLaunchKernelGGL((k_sorted_finish<DCH>), dim3((unsigned)(((int64_t)q.Vk * 16 + 255) / 256)), dim3(256), 0, st, q);
}
static void launch_finish_any(int dch, const SortedParams& q, hipStream_t st) {
    switch (dch) {
        case 1: launch_finish<1>(q, st); break;
        case 2: launch_finish<2>(q, st); break;
        case 3: launch_finish<3>(q, st); break;
        case 4: launch_finish<4>(q, st); break;
        case 6: launch_finish<6>(q, st); break;
        default: launch_finish<8>(q, st); break;
    }
}

// Items of one synchronous mini-batch.  Within a mini-batch the context rows are frozen and every row takes its terms without feedback
// from the other side, so the size is set by TERMS PER LIVE ROW: measured (scripts/quality_sorted.py, profiles/r02_quality_sorted.txt)
// the link-prediction AUC equals the atomics schedule's up to ~120 items per row and mini-batch, slips by 0.001 per ~70 items beyond and
// collapses between 320 and 390 (8 ranks: 73 ms per episode at 128 items per row, 69 at 256 — not worth the margin) —
// and the HOTTEST row counts, not the average one: on a Zipf-popular graph a head row took > 1e5 terms of a 96-per-row mini-batch and
// the tables went to NaN within an epoch.  Hence: 128 items per live row, at most 4096 for the hottest row, and no mini-batch below 5e5
// items (the two sorts and ~16 launches per mini-batch need that much to pay): 0 = this vocabulary is too skewed or too small.
int64_t dge_sorted_batch_items(const dge_model* m, int part_n) {
    const int n = std::max(part_n, 1);
    const int64_t live_rows = std::max<int64_t>(1, m->V / n);
    const double hottest = std::min(1.0, m->row_share_max * (double)n);      // its share of one block's terms
    int64_t items = std::min<int64_t>(96ll << 20, 128 * live_rows);
    // (round 5: 4 096 for the busiest row, was 2 048 — on cfg3 that bound was the one that bound (its busiest vertex holds 30x the mean count: 8.8 M items where the
    //  128-a-row rule allows 16 M) and an epoch of the cfg3-sized community graph on 8 ranks ends at the same AUC 0.9596 / loss 0.474 with 18 M-item mini-batches — the
    //  busiest row at ~4 200 terms — as with 9 M, 10 % faster; 36 M (a whole episode, 288 a row) loses it: 0.9565 / 0.497.  scripts/blocks_minibatch_quality.py,
    //  profiles/r05_blocks_minibatch_quality.txt)
    items = std::min<int64_t>(items, (int64_t)(4096.0 / std::max(hottest, 1e-12)));
    // (wide rows: from half a million items on — round 4: a 50 000-row vocabulary with rank^-0.5 popularity lands at 0.9 M and ran 3.4e8 edges/s at D = 256 under this
    //  schedule against 2.1e8 under the atomics the rule used to leave it with: scripts/policy_sweep.py)
    //  — on rows of more than 128 floats: with D = 64 the same vocabulary runs 7.9e8 under atomics against 4.2e8 here; the sorts do not shrink with the row)
    // One block of the multi-GPU schedule on a vocabulary large enough for the lock kernels (>= 32 768 rows a partition): those — the mixed kernel on a skewed
    // vocabulary — are the alternative there, not atomics, and a mini-batch that the busiest row keeps small loses to them: cfg5 at 2 ranks landed at 0.7 M items and
    // ran 4.6e7 edges/s per rank here against 8.6e7 under the mixed kernel at 4 ranks (round 5).  From 4 M items on (cfg3's blocks: 16 .. 18 M).
    if (n >= 2 && m->V / n >= 32768) return items >= (4 << 20) ? items : 0;
    return items >= (m->stride > 128 ? (1 << 19) : (1 << 20)) ? items : 0;
}

struct CastI64 { __host__ __device__ int64_t operator()(int32_t x) const { return (int64_t)x; } };
// The item sorts.  rocprim's onesweep sorts 8 key bits per pass; row numbers of 17 .. 18 bits (cfg2; a block of the 8- or 4-rank schedule) take
// 3 passes there and 2 with 9-bit digits (scripts/micro/sort_bits.hip: 48 M items of 17 bits 1.01 ms against 1.15, 12.8 M 0.31 against 0.37;
// 10-bit digits lose more per pass than they save).
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<512, 16>, 9,
                                                                       rocprim::block_radix_rank_algorithm::match>> SortWide;
// items are single 64-bit words sorted on the key's bits [shift, shift + key_bits) only: stable, the low bits (other row, label / step code) ride along
// (scripts/micro/sort_keys64.hip: 12.8 M items of 17 key bits 0.230 ms against 0.304 ms as (4-byte key, 8-byte value) pairs, 48 M 0.705 against 0.988)
// (19 .. 20 key bits — a block of the 2-rank schedule on cfg3 —: two passes of 10-bit digits measured 0.970 ms against 0.998 alone on 48 M items and nothing
//  in the trainer (8.09 against 8.20e8 edges/s): rocprim's default stays there)
static hipError_t sort_items(void* tmp, size_t& bytes, const uint64_t* in, uint64_t* out, int64_t n, int shift, int key_bits, hipStream_t st) {
    if (key_bits > 16 && key_bits <= 18) return rocprim::radix_sort_keys<SortWide>(tmp, bytes, in, out, (size_t)n, (unsigned)shift, (unsigned)(shift + key_bits), st);
    return rocprim::radix_sort_keys(tmp, bytes, in, out, (size_t)n, (unsigned)shift, (unsigned)(shift + key_bits), st);
}
static size_t sort_items_tmp_bytes(int64_t cap) {
    size_t a = 0, b = 0;
    (void)sort_items(nullptr, a, nullptr, nullptr, cap, 30, 18, 0);
    (void)sort_items(nullptr, b, nullptr, nullptr, cap, 30, 31, 0);
    return std::max(a, b);
}

typedef hipcub::TransformInputIterator<int64_t, CastI64, const int32_t*> CountIter;      // pair counts summed in 64 bits (an epoch-long launch has > 2^31 pairs)

static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

template <int DCH>
static void launch_phase(const SortedParams& q, bool phase_b, hipStream_t st) {
    const int64_t chunks = (q.n_slots + q.chunk - 1) / q.chunk;
    const unsigned blocks = grid_for(chunks * 16, 256);
    if (phase_b) hipLaunchKernelGGL((k_sorted_phase<DCH, true>), dim3(blocks), dim3(256), 0, st, q);
    else hipLaunchKernelGGL((k_sorted_phase<DCH, false>), dim3(blocks), dim3(256), 0, st, q);
}
static void launch_phase_any(int dch, const SortedParams& q, bool phase_b, hipStream_t st) {
    switch (dch) {
        case 1: launch_phase<1>(q, phase_b, st); break;
        case 2: launch_phase<2>(q, phase_b, st); break;
        case 3: launch_phase<3>(q, phase_b, st); break;
        case 4: launch_phase<4>(q, phase_b, st); break;
        case 6: launch_phase<6>(q, phase_b, st); break;
        default: launch_phase<8>(q, phase_b, st); break;
    }
}

int dge_sorted_train(dge_model* m, const TrainParams& p) {
    if ((uint64_t)m->V * (uint64_t)m->stride * 4ull >= 0xFFFFFFFFull)
        DGE_FAIL(DGE_ERR_ARG, "update_policy 8 addresses a table through one descriptor: tables of 4 GiB and more are not supported");
    if (m->cfg.use_hs) DGE_FAIL(DGE_ERR_ARG, "update_policy 8 does not carry the hierarchical-softmax term");
    if (!m->sorted) m->sorted = new dge_sorted_work();
    dge_sorted_work* s = m->sorted;
    hipStream_t st = m->stream;
    const int64_t n_units = p.n_rows * (int64_t)p.L;
    const int K1 = p.K + 1;
    int rc;
    // Block schedule: the items of every target partition are made once per global batch (k_block_count / k_block_emit) and kept for the
    // batch's n episodes — when they fit: at most half of the free memory (cfg3 at 8 ranks: 27.6 GB); otherwise episode by episode as before
    bool use_store = p.part_n > 1 && p.part_n <= 16 && p.L <= 64 && (int64_t)p.part_n * p.n_rows + 1 < 0x7fffffffll;
    int64_t total_pairs = 0;
    // the packed item's fields: `obits` bits of a row number (tables below 4 GiB: V < 2^24 at the narrowest stride), the key above them
    int obits = 1;
    while ((1ll << obits) < m->V) obits++;
    const int ks1 = 1 + obits, ks2 = 11 + obits;
    // mini-batches of whole walks (dge_sorted_batch_items: ~128 items per live row, the hottest row bounded)
    int64_t want_items = dge_sorted_batch_items(m, p.part_n);
    if (want_items == 0) want_items = 1 << 20;         // asked for explicitly on a vocabulary the rule would not pick it for (train_rows has checked that it is safe)
    // (walks per mini-batch: as many as give want_items — then EVENED OUT over the launch: with 36 M items and 11.4 M a mini-batch the fourth mini-batch was a 1.6 M-item
    //  remainder that cost a tenth of a full one's kernels' fixed parts and went through rocPRIM's small-input merge sort, ~25 launches: 6 % of an 8-rank episode at the
    //  1 M-walk global batch, profiles/r05_sim8_1M_timeline_before.txt)
    auto walks_per_for = [&](double items_per_walk) -> int64_t {
        int64_t wp = std::max<int64_t>(1, (int64_t)((double)want_items / std::max(items_per_walk, 1e-9)));
        if (g_dge_tuning[DGE_TUNE_SORTED_WALKS] > 0) return std::min<int64_t>(g_dge_tuning[DGE_TUNE_SORTED_WALKS], p.n_rows);
        wp = std::min(wp, p.n_rows);
        // (an eighth over the target is allowed where it saves a mini-batch — the target's constants, 128 items a row and 4 096 for the busiest, are not that sharp:
        //  one rank of 8 on cfg3 trains an episode's 36 M items in (as measured with the round's first bound of 2 048 for the busiest row) four mini-batches of 9.0 M instead of five of 7.2 M against a target of 8.8 M)
        const int64_t n_mb = std::max<int64_t>(1, (p.n_rows * 8 + wp * 9 - 1) / (wp * 9));
        return (p.n_rows + n_mb - 1) / n_mb;
    };
    if (!s->aux) {
        DGE_HIP(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
        for (int x = 0; x < 2; x++) {
            DGE_HIP(hipEventCreateWithFlags(&s->ev_ready[x], hipEventDisableTiming));
            DGE_HIP(hipEventCreateWithFlags(&s->ev_done[x], hipEventDisableTiming));
        }
        DGE_HIP(hipEventCreateWithFlags(&s->ev_store, hipEventDisableTiming));
    }
    if (use_store) {
        const dge_sorted_work::Key key{p.sen, p.n_rows, m->seen_gen, p.L, p.W, p.K, p.part_n, p.part_ctx, p.gidx_base, p.V, p.T, p.seed, (int64_t)g_dge_tuning[DGE_TUNE_SORTED_WALKS]};
        const int64_t n_cells = (int64_t)p.part_n * p.n_rows;
        if (!(s->st_valid && m->seen_gen != 0 && key == s->st_key_of)) {
            s->st_valid = false;
            if (n_cells + 1 > s->st_cap_cells) {
                DGE_HIP(hipStreamSynchronize(st));
                dge_dev_free(s->st_cnt); dge_dev_free(s->st_off); dge_dev_free(s->st_scan_tmp); s->st_cnt = nullptr; s->st_off = nullptr; s->st_scan_tmp = nullptr; s->st_cap_cells = 0;
                if ((rc = dge_dev_alloc(&s->st_cnt, (size_t)n_cells + 1))) return rc;
                if ((rc = dge_dev_alloc(&s->st_off, (size_t)n_cells + 1))) return rc;
                size_t b = 0;
                DGE_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b, CountIter(s->st_cnt, CastI64()), s->st_off, n_cells + 1, st));
                DGE_HIP(hipMalloc(&s->st_scan_tmp, b ? b : 1)); s->st_scan_bytes = b;
                s->st_cap_cells = n_cells + 1;
            }
            if (!s->st_words && (rc = dge_dev_alloc(&s->st_words, 1))) return rc;
            DGE_HIP(hipMemsetAsync(s->st_words, 0, sizeof(unsigned long long), st));
            DGE_HIP(hipMemsetAsync(s->st_cnt + n_cells, 0, sizeof(int32_t), st));
            hipLaunchKernelGGL(k_block_count, dim3(std::min<unsigned>(grid_for(p.n_rows, 256), COUNT_BLOCKS * 8)), dim3(256), 0, st, p, s->st_cnt, s->st_words);
            { size_t b = s->st_scan_bytes; DGE_HIP(hipcub::DeviceScan::ExclusiveSum(s->st_scan_tmp, b, CountIter(s->st_cnt, CastI64()), s->st_off, n_cells + 1, st)); }
            // first pair of every bucket
            std::vector<int64_t> cells((size_t)p.part_n + 1);
            for (int t = 0; t <= p.part_n; t++) cells[(size_t)t] = (int64_t)t * p.n_rows;
            if (p.part_n + 1 > s->cap_marks) {
                dge_dev_free(s->d_marks); s->d_marks = nullptr; s->cap_marks = 0;
                if ((rc = dge_dev_alloc(&s->d_marks, 2 * ((size_t)p.part_n + 1)))) return rc;
                s->cap_marks = p.part_n + 1;
            }
            s->st_bucket0.assign((size_t)p.part_n + 1, 0);
            DGE_HIP(hipMemcpyAsync(s->d_marks, cells.data(), cells.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_sorted_marks, dim3(1), dim3(256), 0, st, s->st_off, s->d_marks, p.part_n + 1, s->d_marks + s->cap_marks);
            DGE_HIP(hipMemcpyAsync(s->st_bucket0.data(), s->d_marks + s->cap_marks, cells.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            DGE_HIP(hipStreamSynchronize(st));
            const int64_t all_items = s->st_bucket0[(size_t)p.part_n] * K1;
            size_t free_b = 0, total_b = 0;
            DGE_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t have = (size_t)s->st_cap_items * 8;
            if ((size_t)all_items * 8 > have + (free_b + have) / 2) use_store = false;       // does not fit: this batch runs episode by episode
            else {
                if (all_items > s->st_cap_items) {
                    if (s->aux) DGE_HIP(hipStreamSynchronize(s->aux));
                    dge_dev_free(s->st_it); s->st_it = nullptr; s->st_cap_items = 0;
                    const int64_t cap = all_items + all_items / 16 + 1024;
                    if ((rc = dge_dev_alloc(&s->st_it, (size_t)cap))) return rc;
                    s->st_cap_items = cap;
                }
                if (all_items > 0) {
                    // (the run form only for vocabularies of up to 510 runs: with cfg3's 1 716 — an 11-step search in LDS and f64 arithmetic per draw — this kernel took
                    //  38 ms a batch with one walk per group and 40.8 ms with the groups looping over the batch from a device-filling grid, against 31 ms on the table:
                    //  unlike the lock kernel (§5.1) it is not bound by requests)
                    int32_t run_nb = 2; while (run_nb < p.n_runs + 2 && run_nb < DGE_RUN_MAX) run_nb <<= 1;
                    const bool runs = p.n_runs > 0 && run_nb <= 512;
                    const unsigned eg = runs ? std::min<unsigned>(grid_for(p.n_rows * 16, 256), (unsigned)m->n_cus * 8u) : grid_for(p.n_rows * 16, 256);
                    if (runs) hipLaunchKernelGGL(k_block_emit<true>, dim3(eg), dim3(256), 0, st, p, s->st_off, s->st_it,
                                                         (int32_t)((m->V + p.part_n - 1) / p.part_n), (int32_t)ks1, run_nb);
                    else hipLaunchKernelGGL(k_block_emit<false>, dim3(eg), dim3(256), 0, st, p, s->st_off, s->st_it,
                                            (int32_t)((m->V + p.part_n - 1) / p.part_n), (int32_t)ks1, run_nb);
                }
                DGE_HIP(hipGetLastError());
                // (tried in round 5 and dropped: the store filled on a third stream in eight chunks of walks with an event behind each, so that the batch's first episode starts
                //  behind the first chunk and the rest is generated beside its phases — same box, A / B: 8.03e8 edges/s per rank of 8 against 8.17 - 8.22e8 with the one
                //  generator launch in front: as with the sorts, every kernel fills the device and what runs beside the phases slows them by what it takes)
                DGE_HIP(hipEventRecord(s->ev_store, st));
                // the mini-batches of every bucket: the batch's second and last read-back
                s->st_walks_per.assign((size_t)p.part_n, 1); s->st_marks.assign((size_t)p.part_n, std::vector<int64_t>());
                int64_t n_marks = 0;
                for (int t = 0; t < p.part_n; t++) {
                    const int64_t tp = s->st_bucket0[(size_t)t + 1] - s->st_bucket0[(size_t)t];
                    const int64_t wp = walks_per_for((double)tp * K1 / (double)p.n_rows);
                    s->st_walks_per[(size_t)t] = wp;
                    n_marks += (p.n_rows + wp - 1) / wp + 1;
                }
                if (n_marks > s->cap_marks) {
                    dge_dev_free(s->d_marks); s->d_marks = nullptr; s->cap_marks = 0;
                    if ((rc = dge_dev_alloc(&s->d_marks, 2 * (size_t)n_marks))) return rc;
                    s->cap_marks = n_marks;
                }
                if (n_marks > s->cap_h_marks) {
                    if (s->h_marks) (void)hipHostFree(s->h_marks);
                    s->h_marks = nullptr; s->cap_h_marks = 0;
                    DGE_HIP(hipHostMalloc((void**)&s->h_marks, ((size_t)n_marks + 64) * sizeof(int64_t), hipHostMallocDefault));
                    s->cap_h_marks = n_marks + 64;
                }
                int64_t at = 0;
                for (int t = 0; t < p.part_n; t++) {
                    const int64_t wp = s->st_walks_per[(size_t)t], nm = (p.n_rows + wp - 1) / wp + 1;
                    hipLaunchKernelGGL(k_sorted_marks_walks, dim3(grid_for(nm, 256)), dim3(256), 0, st, s->st_off, (int64_t)t * p.n_rows, (int64_t)1, wp, p.n_rows, (int)nm, s->d_marks + at);
                    at += nm;
                }
                DGE_HIP(hipMemcpyAsync(s->h_marks, s->d_marks, (size_t)n_marks * sizeof(int64_t), hipMemcpyDeviceToHost, st));
                DGE_HIP(hipStreamSynchronize(st));
                at = 0;
                for (int t = 0; t < p.part_n; t++) {
                    const int64_t wp = s->st_walks_per[(size_t)t], nm = (p.n_rows + wp - 1) / wp + 1;
                    s->st_marks[(size_t)t].assign(s->h_marks + at, s->h_marks + at + nm);
                    at += nm;
                }
                s->st_key_of = key; s->st_valid = true;
            }
        }
    }
    if (use_store) {
        total_pairs = s->st_bucket0[(size_t)p.part_tgt + 1] - s->st_bucket0[(size_t)p.part_tgt];
        hipLaunchKernelGGL(k_block_tally, dim3(1), dim3(64), 0, st, p.counters, (unsigned long long)total_pairs, s->st_words, p.part_ctx == p.part_tgt ? 1 : 0);
        if (total_pairs == 0) return DGE_OK;
    } else {
    if (n_units + 1 >= 0x7fffffffll) DGE_FAIL(DGE_ERR_ARG, "update_policy 8: %lld (walk, centre) units in one launch exceed the prefix sum's 32-bit item count: train in several calls", (long long)n_units);
    if (n_units + 1 > s->cap_units) {
        DGE_HIP(hipStreamSynchronize(st));
        dge_dev_free(s->cnt); dge_dev_free(s->off); dge_dev_free(s->scan_tmp); s->cnt = nullptr; s->off = nullptr; s->scan_tmp = nullptr; s->cap_units = 0;
        if ((rc = dge_dev_alloc(&s->cnt, (size_t)n_units + 1))) return rc;
        if ((rc = dge_dev_alloc(&s->off, (size_t)n_units + 1))) return rc;
        size_t b = 0;
        DGE_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b, CountIter(s->cnt, CastI64()), s->off, n_units + 1, st));
        DGE_HIP(hipMalloc(&s->scan_tmp, b ? b : 1)); s->scan_tmp_bytes = b;
        s->cap_units = n_units + 1;
    }
    // pairs of every (walk, centre) unit and their exclusive prefix (one extra zero unit carries the total)
    DGE_HIP(hipMemsetAsync(s->cnt + n_units, 0, sizeof(int32_t), st));
    // (one thread per walk where the partition masks pay; without partitions a unit's count is its window's width: one thread per unit — 42 -> ~10 us on cfg2)
    if (p.L <= 64 && p.part_n > 1) hipLaunchKernelGGL(k_sorted_count_walks, dim3(std::min<unsigned>(grid_for(p.n_rows, 256), COUNT_BLOCKS * 2)), dim3(256), 0, st, p, s->cnt);
    else hipLaunchKernelGGL(k_sorted_count, dim3(std::min<unsigned>(grid_for(n_units, 256), COUNT_BLOCKS * 2)), dim3(256), 0, st, p, s->cnt);
    { size_t b = s->scan_tmp_bytes; DGE_HIP(hipcub::DeviceScan::ExclusiveSum(s->scan_tmp, b, CountIter(s->cnt, CastI64()), s->off, n_units + 1, st)); }

    }
    // How many walks make a mini-batch: under the item store the bucket's mini-batches were laid out with the store (nothing is read back per episode); otherwise from
    // the EXPECTED pairs of a full-length walk (DL4J's window: radius uniform in 1 .. W) — round 4: the launch's pair total used to be read back for this, one of two
    // host round trips that cost a cfg2 launch ~8 % (shorter walks only make mini-batches smaller)
    int64_t walks_per, n_sub;
    const int64_t* h_off;
    std::vector<int64_t> marks;
    if (use_store) {
        walks_per = s->st_walks_per[(size_t)p.part_tgt];
        n_sub = (p.n_rows + walks_per - 1) / walks_per;
        h_off = s->st_marks[(size_t)p.part_tgt].data();
        DGE_HIP(hipStreamWaitEvent(s->aux, s->ev_store, 0));          // (the store's items were written on the model's stream)
    } else {
        double items_per_walk = dge_expected_pairs_per_walk(p.L, p.W) * K1;
        // one block of an n-rank schedule holds the pairs (context in one partition, centre in another): 1 / n^2 of the walk's (without this a block
        // that runs without the item store — more than 16 ranks, or a store that does not fit — cut its mini-batches n^2 times too small)
        if (p.part_n > 1) items_per_walk /= (double)p.part_n * (double)p.part_n;
        walks_per = walks_per_for(items_per_walk);
        n_sub = (p.n_rows + walks_per - 1) / walks_per;
        marks.resize((size_t)n_sub + 1);
        // marks: the unit at which every mini-batch begins -> its first pair
        for (int64_t k = 0; k <= n_sub; k++) marks[(size_t)k] = std::min(k * walks_per, p.n_rows) * p.L;
        if (n_sub + 1 > s->cap_marks) {
            DGE_HIP(hipStreamSynchronize(st));
            dge_dev_free(s->d_marks); s->d_marks = nullptr;
            if ((rc = dge_dev_alloc(&s->d_marks, 2 * ((size_t)n_sub + 1)))) return rc;
            s->cap_marks = n_sub + 1;
        }
        if (n_sub + 1 > s->cap_h_marks) {
            if (s->h_marks) (void)hipHostFree(s->h_marks);
            s->h_marks = nullptr; s->cap_h_marks = 0;
            DGE_HIP(hipHostMalloc((void**)&s->h_marks, ((size_t)n_sub + 1 + 64) * sizeof(int64_t), hipHostMallocDefault));
            s->cap_h_marks = n_sub + 1 + 64;
        }
        // the launch's ONE read-back: the first pair of every mini-batch (the sorts take their item counts from the host), computed on the device, into pinned memory
        hipLaunchKernelGGL(k_sorted_marks_walks, dim3(grid_for(n_sub + 1, 256)), dim3(256), 0, st, s->off, (int64_t)0, (int64_t)p.L, walks_per, p.n_rows, (int)(n_sub + 1), s->d_marks);
        DGE_HIP(hipMemcpyAsync(s->h_marks, s->d_marks, ((size_t)n_sub + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        DGE_HIP(hipStreamSynchronize(st));
        h_off = s->h_marks;
        total_pairs = h_off[n_sub] - h_off[0]; if (total_pairs == 0) return DGE_OK;
    }

    int64_t max_slots = 0;
    for (int64_t k = 0; k < n_sub; k++) max_slots = std::max(max_slots, (h_off[k + 1] - h_off[k]) * K1);
    if (max_slots >= 0x7fffffffll) DGE_FAIL(DGE_ERR_ARG, "update_policy 8: a mini-batch of %lld items; set fewer walks per mini-batch", (long long)max_slots);
    int chunk = 128;                     // (64 .. 256 measure alike; 512 and up lose: fewer work units than the device holds)
    if (g_dge_tuning[DGE_TUNE_SORTED_CHUNK] > 0) chunk = (int)std::min<int64_t>(g_dge_tuning[DGE_TUNE_SORTED_CHUNK], 1 << 20);
    const int kdiv = std::max(p.part_n, 1);
    const int64_t Vk = (m->V + kdiv - 1) / kdiv;
    int key_bits = 1;
    while (key_bits < 31 && (1ll << key_bits) <= Vk) key_bits++;              // keys are 0 .. Vk (Vk = a skipped draw)
    if (ks2 + key_bits > 64) DGE_FAIL(DGE_ERR_ARG, "update_policy 8: %lld vocabulary rows do not fit the packed item", (long long)m->V);
    // (growth of a work buffer: earlier launches may still be using the old one on either stream — since round 5 nothing else makes the host wait for them)
    if (max_slots > s->cap_items) {
        DGE_HIP(hipStreamSynchronize(st)); DGE_HIP(hipStreamSynchronize(s->aux));
        const int64_t cap = max_slots + max_slots / 8 + 1024;
        size_t b = 0;
        for (int x = 0; x < 2; x++) {
            dge_dev_free(s->it0[x]); dge_dev_free(s->it1[x]); dge_dev_free(s->sort_tmp[x]);
            s->it0[x] = s->it1[x] = nullptr; s->sort_tmp[x] = nullptr;
        }
        s->cap_items = 0; s->set_used[0] = s->set_used[1] = false; s->live = 0;
        for (int x = 0; x < 2; x++) {
            if ((rc = dge_dev_alloc(&s->it0[x], (size_t)cap))) return rc;
            if ((rc = dge_dev_alloc(&s->it1[x], (size_t)cap))) return rc;
            b = sort_items_tmp_bytes(cap);
            DGE_HIP(hipMalloc(&s->sort_tmp[x], b ? b : 1));
        }
        s->sort_tmp_bytes = b;
        s->cap_items = cap;
    }
    if (Vk + 2 > s->cap_seg) {
        DGE_HIP(hipStreamSynchronize(st)); DGE_HIP(hipStreamSynchronize(s->aux));
        dge_dev_free(s->seg); s->seg = nullptr;
        if ((rc = dge_dev_alloc(&s->seg, 3 * ((size_t)Vk + 2)))) return rc;
        s->cap_seg = Vk + 2;
    }
    if (Vk > s->cap_shadow) {
        DGE_HIP(hipStreamSynchronize(st));
        dge_dev_free(s->shadow); s->shadow = nullptr;
        if ((rc = dge_dev_alloc(&s->shadow, (size_t)Vk * (size_t)m->stride + 64))) return rc;
        s->cap_shadow = Vk;
    }
    const int64_t need_rows = 2 * ((s->cap_items + chunk - 1) / chunk) + 2;
    if (need_rows > s->cap_scratch_rows) {
        DGE_HIP(hipStreamSynchronize(st));
        dge_dev_free(s->scratch); dge_dev_free(s->scratch_b); s->scratch = nullptr; s->scratch_b = nullptr;
        if ((uint64_t)need_rows * (uint64_t)m->stride * 4ull >= 0xFFFFFFFFull) DGE_FAIL(DGE_ERR_ARG, "update_policy 8: chunk size %d too small for %lld items", chunk, (long long)s->cap_items);
        if ((rc = dge_dev_alloc(&s->scratch, (size_t)need_rows * (size_t)m->stride))) return rc;
        if ((rc = dge_dev_alloc(&s->scratch_b, (size_t)need_rows * (size_t)m->stride))) return rc;
        s->cap_scratch_rows = need_rows;
    }

    SortedParams q;
    q.t = p; q.cnt = s->cnt; q.off = s->off; q.seg = s->seg; q.chunk = chunk; q.scratch = s->scratch; q.shadow = s->shadow;
    q.scratch_a = s->scratch; q.scratch_b = s->scratch_b; q.seg_tgt = s->seg;
    q.kdiv = kdiv; q.kpart_tgt = p.part_n > 1 ? p.part_tgt : 0; q.kpart_ctx = p.part_n > 1 ? p.part_ctx : 0; q.Vk = (int32_t)Vk;
    q.ks1 = ks1; q.ks2 = ks2; q.omask = (uint32_t)((1ull << obits) - 1ull); q.mb_walk0 = 0; q.it_in = nullptr; q.it_out = nullptr;
    int64_t* const seg_a[2] = {s->seg, s->seg + Vk + 2}; int64_t* const seg_b = s->seg + 2 * (Vk + 2);
    const int dch = m->stride / 64;
    // (the offsets were read back above: everything the second stream reads — counts, offsets, walks, the unigram table — is in place)
    for (int64_t k = 0; k < n_sub; k++) {
        const int64_t n = (h_off[k + 1] - h_off[k]) * K1;
        if (n == 0) continue;
        const int x = (int)(s->live++ & 1);
        if (!use_store) { q.unit0 = marks[(size_t)k]; q.unit1 = marks[(size_t)k + 1]; }
        q.pair0 = h_off[k]; q.n_slots = n;      // (unit0, unit1, pair0: the per-episode emit only)
        // second stream: items -> it0; sorted by target row -> it1; row segments
        if (s->set_used[x]) DGE_HIP(hipStreamWaitEvent(s->aux, s->ev_done[x], 0));         // the mini-batch that held this set has finished
        q.it_out = s->it0[x];
        q.mb_walk0 = std::min(k * walks_per, p.n_rows - 1);                                 // the mini-batch's learning rate: that of its first walk
        const uint64_t* src = s->it0[x];
        if (use_store) src = s->st_it + h_off[k] * K1;                              // the batch's items are already there
        else {
            q.run_nb = 2; while (q.run_nb < p.n_runs + 2 && q.run_nb < DGE_RUN_MAX) q.run_nb <<= 1;
            const bool runs = p.n_runs > 0 && q.run_nb <= 512;          // (more runs: the LDS fill and the longer search cost what the look-ups did — k_block_emit below)
            const int64_t n_units = q.unit1 - q.unit0;
            if (p.part_n > 1) {         // one 16-lane group per 16 units
                if (runs) hipLaunchKernelGGL((k_sorted_emit<16, true>), dim3(grid_for(n_units + 15, 256)), dim3(256), 0, s->aux, q);
                else hipLaunchKernelGGL((k_sorted_emit<16, false>), dim3(grid_for(n_units + 15, 256)), dim3(256), 0, s->aux, q);
            } else {                    // ... per 4 units
                if (runs) hipLaunchKernelGGL((k_sorted_emit<4, true>), dim3(grid_for((n_units + 3) / 4 * 16, 256)), dim3(256), 0, s->aux, q);
                else hipLaunchKernelGGL((k_sorted_emit<4, false>), dim3(grid_for((n_units + 3) / 4 * 16, 256)), dim3(256), 0, s->aux, q);
            }
        }
        size_t b = s->sort_tmp_bytes;
        DGE_HIP(sort_items(s->sort_tmp[0], b, src, s->it1[x], n, ks1, key_bits, s->aux));
        hipLaunchKernelGGL(k_sorted_segments, dim3(grid_for(Vk + 2, 256)), dim3(256), 0, s->aux, s->it1[x], ks1, n, Vk, seg_a[x]);
        DGE_HIP(hipEventRecord(s->ev_ready[x], s->aux));
        // model's stream — phase A: target rows move (into the shadow table); context key | target row | step code -> it0
        DGE_HIP(hipStreamWaitEvent(st, s->ev_ready[x], 0));
        q.it_in = s->it1[x]; q.it_out = s->it0[x]; q.seg = seg_a[x]; q.seg_tgt = seg_a[x]; q.scratch = s->scratch;
        launch_phase_any(dch, q, false, st);
        // sorted by context row -> it1; phase B: context rows take their sums
        b = s->sort_tmp_bytes;
        DGE_HIP(sort_items(s->sort_tmp[1], b, s->it0[x], s->it1[x], n, ks2, key_bits, st));
        hipLaunchKernelGGL(k_sorted_segments, dim3(grid_for(Vk + 2, 256)), dim3(256), 0, st, s->it1[x], ks2, n, Vk, seg_b);
        q.seg = seg_b; q.scratch = s->scratch_b;
        launch_phase_any(dch, q, true, st);            // reads the target rows as they stood BEFORE the mini-batch
        launch_finish_any(dch, q, st);                 // straddling rows of both sides take their segments' deltas, the target rows are committed
        DGE_HIP(hipEventRecord(s->ev_done[x], st));
        s->set_used[x] = true;
    }
    DGE_HIP(hipGetLastError());
    return DGE_OK;
}
