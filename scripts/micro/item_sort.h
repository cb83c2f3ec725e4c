// item_sort.h — EXPERIMENT, NOT PART OF libdge.so (round 5; verdict item "replace the generic radix sort").  Built, bit-identical to rocPRIM's output on 780 edge cases and
// on every size / key width the schedule sorts, and SLOWER in isolation: 0.75 - 0.86x at 17 key bits, 0.4 - 0.7x at 20 (profiles/r05_item_sort.txt: 7.2 M items of 17 key
// bits 0.214 ms against 0.161; 48 M 0.930 against 0.729).  Ablations (same file's history): with the look-back's LOADS removed (stores kept) the same kernels run 1.0 - 1.6x
// rocPRIM's speed — the cost is the walk: every workgroup slot of the device holds a tile of the same "wave" of tiles, all of them publish their AGGREGATE at about the same
// time and none has a PREFIX yet, so tile t of the wave walks back through ~t aggregates (8 per round trip) for each of its 512 digits: two orders of magnitude more status
// loads than items.  Overlapping the first look-back load with the LDS reorder, tiles of 8 192 items and an adaptive window bought 10 % together.  What would fix it is a
// bounded walk (aggregates over power-of-two spans of tiles published level by level: ~5 + t / 32 loads instead of t) — not built: even without any look-back the rank-and-
// scatter kernel alone (~60 us a pass at 7.2 M items) only matches rocPRIM's pass, so the ceiling of this design is parity, and what it would save in the pipeline (five fills
// and a scan kernel: ~35 us a sort) does not pay for the risk.  The schedule keeps rocPRIM's sort.  Kept with its harness (item_sort_bench.hip) for whoever takes it further.
//
// The owner-computes schedule's item sort (gfx950): a stable LSD radix sort of packed 64-bit items on a bit range [shift, shift + key_bits) of at most
// 31 bits, the low bits riding along.  Round 5: written for this one use in place of rocPRIM's generic onesweep, which was a third of an 8-rank episode's kernel time
// (profiles/r04_sim8_kernel_stats.csv) and, per sort, five buffer fills, a histogram kernel, a scan kernel and a pass per 8 (9) key bits.  Here a sort is ONE histogram
// kernel (the digit histograms of all passes in one read) and one kernel per pass — two passes for the 17 .. 20 key bits of this schedule (digits of up to 10 bits):
//   * nothing is filled between sorts: the decoupled look-back's status words carry the number of the pass they were written in (a word of another pass reads as
//     "not yet"), tiles take their numbers from a ticket counter that is never reset (the host knows where each pass's tickets begin), and the two histogram buffers
//     take turns — the first pass kernel of one sort clears the buffer of the next;
//   * the exclusive scan of a pass's digit histogram is redone by every tile from the 2^bits counts (L2-resident) instead of by a kernel of its own.
// One pass (k_rs_pass), per tile of 4 096 items in a workgroup of 256: every wave ranks its 1 024 items by digit in item order (lanes that hold the same digit find each
// other with one ballot per digit bit; a per-wave counter in LDS carries a digit's count from round to round), the waves' counts give the tile's digit counts, a thread
// per digit publishes them and looks back over the tiles in front (decoupled look-back: AGGREGATE / PREFIX words), the items are put in digit order in LDS and leave as
// runs of consecutive addresses.  Equal digits keep their order: the sort is stable, pass by pass, like the one it replaces (the bit-exact tests of
// tests/test_gpu_sorted.py did not move; scripts/micro/item_sort_bench.hip compares the two on every size and key width the schedule uses, and times them).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../embedding_amd/csrc/dge_internal.h"

#define RS_THREADS 256
#define RS_IPT 16
#define RS_TILE (RS_THREADS * RS_IPT)
#define RS_WAVES (RS_THREADS / 64)
#define RS_WAVE_ITEMS (RS_TILE / RS_WAVES)
#define RS_MAX_BITS 10
#define RS_MAX_DIGITS (1 << RS_MAX_BITS)
#define RS_MAX_PASSES 4
#define RS_LOOK 8
#define RS_FLAG_AGG 1ull
#define RS_FLAG_PREFIX 2ull

struct RsHistParams {
    const uint64_t* in; int64_t n;
    int n_pass; int shift[RS_MAX_PASSES]; int dbits[RS_MAX_PASSES];
    uint32_t* hist;                              // [RS_MAX_PASSES][RS_MAX_DIGITS], zero on entry
};
struct RsPassParams {
    const uint64_t* in; uint64_t* out; int64_t n;
    int shift, dbits;
    const uint32_t* hist;                        // this pass's digit counts [1 << dbits]
    unsigned long long* status;                  // [tiles][1 << dbits]: epoch << 34 | flag << 32 | count
    unsigned long long* ticket; unsigned long long ticket_base;
    unsigned long long epoch;
    uint32_t* clear; int n_clear;                // the histogram buffer of the NEXT sort: cleared by tile 0 (nullptr: not this pass's job)
};

__global__ void __launch_bounds__(256) k_rs_hist(RsHistParams q) {
    __shared__ uint32_t s_h[RS_MAX_PASSES * RS_MAX_DIGITS];
    for (int i = threadIdx.x; i < q.n_pass * RS_MAX_DIGITS; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t it = __builtin_nontemporal_load(q.in + i);
#pragma unroll
        for (int ps = 0; ps < RS_MAX_PASSES; ps++)
            if (ps < q.n_pass) atomicAdd(&s_h[ps * RS_MAX_DIGITS + (int)((it >> q.shift[ps]) & ((1ull << q.dbits[ps]) - 1ull))], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < q.n_pass * RS_MAX_DIGITS; i += blockDim.x) {
        const uint32_t v = s_h[i];
        if (v) atomicAdd(&q.hist[i], v);
    }
}

// a workgroup barrier that waits for this wave's LDS operations only — __syncthreads() also drains the wave's outstanding GLOBAL loads (its fence), which is
// exactly what the pass must not do while the look-back's answer is in flight
__device__ __forceinline__ void rs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive scan over the workgroup of per-thread values held DPT to a thread (thread t: entries t * DPT .. + DPT - 1, in order); returns the exclusive prefix of the thread's first entry
template <int DPT>
__device__ __forceinline__ uint32_t rs_block_excl(const uint32_t (&v)[DPT], uint32_t* s_wave /* [RS_WAVES] */) {
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < DPT; j++) mine += v[j];
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, o); if ((int)(threadIdx.x & 63) >= o) inc += u; }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += s_wave[w];
    __syncthreads();
    return base + inc - mine;
}

template <int DPT>      // digits per thread: (1 << dbits) <= RS_THREADS * DPT
__global__ void __launch_bounds__(RS_THREADS) k_rs_pass(RsPassParams q) {
    __shared__ uint64_t s_items[RS_TILE];
    __shared__ uint16_t s_cnt[RS_WAVES][RS_THREADS * DPT]; // a wave's running digit counts, then its offset inside the tile's run of the digit
    __shared__ uint32_t s_start[RS_THREADS * DPT];         // first position of the digit in the tile's sorted order
    __shared__ uint32_t s_gbase[RS_THREADS * DPT];         // output position of the digit's first item of this tile, minus s_start (mod 2^32)
    __shared__ uint32_t s_wave[RS_WAVES];
    __shared__ uint32_t s_tile;
    const int ND = 1 << q.dbits;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = (uint32_t)(atomicAdd(q.ticket, 1ull) - q.ticket_base);
    for (int i = threadIdx.x; i < RS_WAVES * RS_THREADS * DPT; i += RS_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile == 0 && q.clear) for (int i = threadIdx.x; i < q.n_clear; i += RS_THREADS) q.clear[i] = 0;
    const int64_t t0 = tile * RS_TILE;
    const int n_here = (int)min((int64_t)RS_TILE, q.n - t0);
    const uint64_t dmask = (uint64_t)(ND - 1);

    // ---- a wave ranks its 1 024 items by digit, in item order
    uint64_t item[RS_IPT]; uint32_t meta[RS_IPT];          // meta: digit << 16 | rank among the wave's items of that digit
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {                     // (all loads first: the ranking below is a chain through LDS)
        const int j = wv * RS_WAVE_ITEMS + i * 64 + lane;
        item[i] = j < n_here ? __builtin_nontemporal_load(q.in + t0 + j) : 0ull;
    }
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        const int j = wv * RS_WAVE_ITEMS + i * 64 + lane;
        const bool valid = j < n_here;
        const uint32_t d = (uint32_t)((item[i] >> q.shift) & dmask);
        // the lanes that hold the same digit: one ballot per digit bit, kept as two 32-bit halves (peers &= ~(ballot ^ -bit))
        const unsigned long long bv = __ballot(valid);
        uint32_t p_lo = (uint32_t)bv, p_hi = (uint32_t)(bv >> 32);
#ifdef RS_DBG_NORANK
        if (false)
#endif
#pragma unroll
        for (int b = 0; b < RS_MAX_BITS; b++) {
            if (b < q.dbits) {
                const uint32_t m = 0u - ((d >> b) & 1u);
                const unsigned long long bal = __ballot(m != 0u);
                p_lo &= ~((uint32_t)bal ^ m); p_hi &= ~((uint32_t)(bal >> 32) ^ m);
            }
        }
        const unsigned long long peers = ((unsigned long long)p_hi << 32) | p_lo;
        const uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        uint32_t pre = 0;
        if (valid) pre = s_cnt[wv][d];
        if (valid && before == 0) s_cnt[wv][d] = (uint16_t)(pre + (uint32_t)__popcll(peers));     // (the wave's LDS operations execute in order: every peer has read)
        meta[i] = (d << 16) | (pre + before);
    }
    __syncthreads();

    // ---- the tile's digit counts; a wave's offset inside a digit's run; thread t owns digits t * DPT ..
    uint32_t cnt[DPT], gh[DPT];
#pragma unroll
    for (int j = 0; j < DPT; j++) {
        const int d = threadIdx.x * DPT + j;
        uint32_t run = 0;
        if (d < ND) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = (uint16_t)run; run += c; }
        }
        cnt[j] = run;
        gh[j] = d < ND ? q.hist[d] : 0u;
    }
    const uint32_t tile_excl = rs_block_excl<DPT>(cnt, s_wave);        // position of the thread's first digit in the tile's order
    const uint32_t glob_excl = rs_block_excl<DPT>(gh, s_wave);         // ... and in the whole output
    // ---- publish the tile's counts and ASK for the word of the tile in front — then put the items in digit order in LDS while that answer travels (a barrier that
    //      only waits for LDS: rs_lds_barrier), and only then look at it: the look-back's round trip to memory was half of a pass when the workgroup sat on it
    unsigned long long w1[DPT];
    {
        uint32_t ts = tile_excl;
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = threadIdx.x * DPT + j;
            w1[j] = 0;
            if (d < ND) {
#ifndef RS_DBG_NOSTORE
                __hip_atomic_store(q.status + (uint64_t)tile * ND + d, (q.epoch << 34) | ((tile == 0 ? RS_FLAG_PREFIX : RS_FLAG_AGG) << 32) | (unsigned long long)cnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#ifdef RS_DBG_NOLOAD
                if (false)
#endif
                if (tile > 0) w1[j] = __hip_atomic_load(q.status + (uint64_t)(tile - 1) * ND + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_start[d] = ts;
            }
            ts += cnt[j];
        }
    }
    rs_lds_barrier();
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        const int j = wv * RS_WAVE_ITEMS + i * 64 + lane;
        if (j < n_here) {
            const uint32_t d = meta[i] >> 16;
            s_items[s_start[d] + s_cnt[wv][d] + (meta[i] & 0xFFFFu)] = item[i];
        }
    }
    // ---- decoupled look-back, a digit at a time: how many items of the digit lie in the tiles in front
    {
        uint32_t ts = tile_excl, gs = glob_excl;
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const int d = threadIdx.x * DPT + j;
            if (d < ND) {
                unsigned long long* my = q.status + (uint64_t)tile * ND + d;
                uint32_t excl = 0;
#ifdef RS_DBG_NOLOAD
                if (false)
#endif
                if (tile > 0) {
                    int64_t tp = tile - 1;
                    if ((w1[j] >> 34) == q.epoch) { excl += (uint32_t)w1[j]; tp = ((w1[j] >> 32) & 3ull) == RS_FLAG_PREFIX ? -1 : tp - 1; }
                    // (then a window of RS_LOOK tiles per step: their words are asked for together — the loads do not depend on each other — and taken in order; a word
                    //  of another pass means "not written yet": the window is asked for again from there)
                    for (; tp >= 0;) {
                        unsigned long long w[RS_LOOK];
#pragma unroll
                        for (int z = 0; z < RS_LOOK; z++) w[z] = __hip_atomic_load(q.status + (uint64_t)max(tp - z, (int64_t)0) * ND + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bool done = false;
#pragma unroll
                        for (int z = 0; z < RS_LOOK; z++) {
                            if (done || tp < 0) break;
                            if ((w[z] >> 34) != q.epoch) break;
                            excl += (uint32_t)w[z];
                            tp--;
                            if (((w[z] >> 32) & 3ull) == RS_FLAG_PREFIX) { done = true; tp = -1; }
                        }
                        if (done) break;
                    }
#ifndef RS_DBG_NOSTORE
                    __hip_atomic_store(my, (q.epoch << 34) | (RS_FLAG_PREFIX << 32) | (unsigned long long)(excl + cnt[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                }
                s_gbase[d] = gs + excl - ts;
            }
            ts += cnt[j]; gs += gh[j];
        }
    }
    rs_lds_barrier();
#pragma unroll
    for (int i = 0; i < RS_IPT; i++) {
        const int j = i * RS_THREADS + threadIdx.x;
        if (j < n_here) {
            const uint64_t it = s_items[j];
            const uint32_t d = (uint32_t)((it >> q.shift) & dmask);
#ifdef RS_DBG_LINEAR
            q.out[t0 + j] = it;
#else
            q.out[(uint32_t)(s_gbase[d] + (uint32_t)j)] = it;
#endif
        }
    }
}

// ---- host side: one sorter per stream (its buffers are in use from the call until the stream has passed the sort)
struct ItemSorter {
    uint32_t* hist[2] = {nullptr, nullptr};
    unsigned long long* status = nullptr; int64_t cap_tiles = 0;
    unsigned long long* ticket = nullptr; unsigned long long tickets_used = 0;
    unsigned long long epoch = 0;
    int parity = 0;
    uint64_t* tmp = nullptr; int64_t cap_items = 0;
    int n_cus = 256;

    void release() {
        dge_dev_free(hist[0]); dge_dev_free(hist[1]); dge_dev_free(status); dge_dev_free(ticket); dge_dev_free(tmp);
        hist[0] = hist[1] = nullptr; status = nullptr; ticket = nullptr; tmp = nullptr; cap_tiles = 0; cap_items = 0;
    }
    // room for sorts of up to n_max items (the caller has made sure the stream is idle when this grows a buffer)
    int ensure(int64_t n_max, hipStream_t st) {
        int rc;
        if (!hist[0]) {
            for (int x = 0; x < 2; x++) {
                if ((rc = dge_dev_alloc(&hist[x], (size_t)RS_MAX_PASSES * RS_MAX_DIGITS))) return rc;
                DGE_HIP(hipMemsetAsync(hist[x], 0, (size_t)RS_MAX_PASSES * RS_MAX_DIGITS * sizeof(uint32_t), st));
            }
            if ((rc = dge_dev_alloc(&ticket, 1))) return rc;
            DGE_HIP(hipMemsetAsync(ticket, 0, sizeof(unsigned long long), st));
            tickets_used = 0; epoch = 0; parity = 0;
        }
        const int64_t tiles = (n_max + RS_TILE - 1) / RS_TILE + 1;
        if (tiles > cap_tiles) {
            dge_dev_free(status); status = nullptr; cap_tiles = 0;
            if ((rc = dge_dev_alloc(&status, (size_t)tiles * RS_MAX_DIGITS))) return rc;
            DGE_HIP(hipMemsetAsync(status, 0, (size_t)tiles * RS_MAX_DIGITS * sizeof(unsigned long long), st));      // (epoch 0: no pass ever has it)
            cap_tiles = tiles;
        }
        if (n_max > cap_items) {
            dge_dev_free(tmp); tmp = nullptr; cap_items = 0;
            if ((rc = dge_dev_alloc(&tmp, (size_t)n_max + 64))) return rc;
            cap_items = n_max;
        }
        return DGE_OK;
    }
    // out = in sorted on bits [shift, shift + key_bits); in is not modified; n <= what ensure() was told
    int sort(const uint64_t* in, uint64_t* out, int64_t n, int shift, int key_bits, hipStream_t st) {
        if (n <= 0) return DGE_OK;
        if (key_bits < 1) key_bits = 1;
        const int n_pass = (key_bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
        if (n_pass > RS_MAX_PASSES || n > cap_items) DGE_FAIL(DGE_ERR_ARG, "item sort: %d key bits / %lld items beyond what the sorter was sized for", key_bits, (long long)n);
        RsHistParams h{};
        h.in = in; h.n = n; h.n_pass = n_pass; h.hist = hist[parity];
        int at = shift, left = key_bits;
        for (int ps = 0; ps < n_pass; ps++) { const int b = (left + (n_pass - ps) - 1) / (n_pass - ps); h.shift[ps] = at; h.dbits[ps] = b; at += b; left -= b; }
        const unsigned hg = (unsigned)std::min<int64_t>((n + 256 * 8 - 1) / (256 * 8), (int64_t)n_cus * 8);
        hipLaunchKernelGGL(k_rs_hist, dim3(hg), dim3(256), 0, st, h);
        const int64_t tiles = (n + RS_TILE - 1) / RS_TILE;
        const uint64_t* src = in;
        for (int ps = 0; ps < n_pass; ps++) {
            if (epoch >= (1ull << 30) - 2) {     // the status words' epoch field is about to wrap: start over on cleared words
                DGE_HIP(hipMemsetAsync(status, 0, (size_t)cap_tiles * RS_MAX_DIGITS * sizeof(unsigned long long), st));
                epoch = 0;
            }
            RsPassParams q{};
            q.in = src; q.out = ((n_pass - 1 - ps) & 1) ? tmp : out; q.n = n;
            q.shift = h.shift[ps]; q.dbits = h.dbits[ps];
            q.hist = hist[parity] + (size_t)ps * RS_MAX_DIGITS;
            q.status = status; q.ticket = ticket; q.ticket_base = tickets_used; q.epoch = ++epoch;
            q.clear = ps == 0 ? hist[parity ^ 1] : nullptr; q.n_clear = RS_MAX_PASSES * RS_MAX_DIGITS;
            if (q.dbits <= 8) hipLaunchKernelGGL((k_rs_pass<1>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, q);
            else if (q.dbits == 9) hipLaunchKernelGGL((k_rs_pass<2>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, q);
            else hipLaunchKernelGGL((k_rs_pass<4>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, q);
            tickets_used += (unsigned long long)tiles;
            src = q.out;
        }
        parity ^= 1;
        DGE_HIP(hipGetLastError());
        return DGE_OK;
    }
};
