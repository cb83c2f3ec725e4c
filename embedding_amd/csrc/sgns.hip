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
#include <charconv>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "dge_algos.h"
#include "dge_internal.h"
#include "sgns_kernels.h"
#include "sgns_model.h"
#include "fmt_g9.h"


std::atomic<int64_t> g_dge_tuning[DGE_TUNE_COUNT];
namespace { struct TuningInit { TuningInit() { for (auto& k : g_dge_tuning) k.store(-1); } } g_tuning_init; }
std::atomic<int64_t> g_dge_host_syncs{0};
extern "C" int dge_host_sync_count(int64_t* n) {
    if (!n) DGE_FAIL(DGE_ERR_ARG, "dge_host_sync_count: null output");
    *n = g_dge_host_syncs.load(std::memory_order_relaxed);
    return DGE_OK;
}
extern "C" int dge_set_tuning(int32_t knob, int64_t value) {
    if (knob < 0 || knob >= DGE_TUNE_COUNT) DGE_FAIL(DGE_ERR_ARG, "dge_set_tuning: unknown knob %d", knob);
    g_dge_tuning[knob] = value < 0 ? -1 : value;
    return DGE_OK;
}
extern "C" int dge_get_tuning(int32_t knob, int64_t* value) {
    if (knob < 0 || knob >= DGE_TUNE_COUNT || !value) DGE_FAIL(DGE_ERR_ARG, "dge_get_tuning: unknown knob %d", knob);
    *value = g_dge_tuning[knob];
    return DGE_OK;
}

#ifndef DGE_KERNELS_HASH
#define DGE_KERNELS_HASH "unknown"
#endif
#ifndef DGE_SORTED_HASH
#define DGE_SORTED_HASH "unknown"
#endif
extern "C" const char* dge_build_stamp(void) { return "kernels=" DGE_KERNELS_HASH " sorted=" DGE_SORTED_HASH; }

// ------------------------------------------------------------------------------------------ where the tables lie
// Which memory a table of random rows lies in decides how fast rows can be read AND WRITTEN BACK in it: allocations of half a gigabyte fall into
// two classes 15 % apart (a microbenchmark of random 512-byte rows read and stored back reaches 6.3 or 7.3 TB/s on them, nothing in between),
// a launch of the lock kernel takes 412-415 ms with both tables in fast memory and 473-479 ms with both in slow memory, and no property of
// the allocation visible from user space tells the classes apart (profiles/r03_placement.txt: not contiguity, alignment, page-table fragments,
// position or the neighbours) — but a 2-millisecond probe does.  So a table is the best of several virtual-memory allocations under that
// probe: candidates are created one after the other (all held, or the allocator would hand the same memory out again) until the fast class has
// shown (the best at least 14 % above the worst: probe rates come in three levels, ~4150 / ~4620 / ~4810 GB/s) or TABLE_CANDIDATES have been seen;
// the best stays.  Fast memory is 1 allocation in 2 ... 6 on most boxes; candidates are hipMalloc and virtual-memory allocations in turn.
#define TABLE_CANDIDATES 32
typedef unsigned int pv4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_probe_table(char* base, uint64_t rows, int iters, float* sink) {
    const int lane = threadIdx.x & 15;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = 0x9E3779B97F4A7C15ull * (group + 1);
    float acc = 0.f;
    for (int i = 0; i < iters; i += 8) {
        pv4u v[8][2]; char* pp[8];
#pragma unroll
        for (int z = 0; z < 8; z++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            pp[z] = base + ((s >> 20) % rows) * 512u + (uint32_t)lane * 16u;
            v[z][0] = __builtin_nontemporal_load((pv4u*)pp[z]); v[z][1] = __builtin_nontemporal_load((pv4u*)(pp[z] + 256));
        }
#pragma unroll
        for (int z = 0; z < 8; z++) {
            acc += __uint_as_float(v[z][0].x ^ v[z][1].y);
            __hip_atomic_store((unsigned*)pp[z], v[z][0].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              // the same bytes back, write-through
            __hip_atomic_store((unsigned*)(pp[z] + 256), v[z][1].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
struct VmAlloc { size_t bytes; hipMemGenericAllocationHandle_t h; };
static std::map<void*, VmAlloc> g_vm_allocs;
static std::mutex g_vm_mu;
static int vm_alloc(void** out, size_t bytes, int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    const size_t g = (size_t)2 << 20, sz = (bytes + g - 1) / g * g;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return DGE_ERR_DEVICE; }
    void* va = nullptr;
    if (hipMemAddressReserve(&va, sz, g, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return DGE_ERR_DEVICE; }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemMap(va, sz, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemAddressFree(va, sz); (void)hipMemRelease(h); return DGE_ERR_DEVICE; }
    if (hipMemSetAccess(va, sz, &acc, 1) != hipSuccess) { (void)hipGetLastError(); (void)hipMemUnmap(va, sz); (void)hipMemAddressFree(va, sz); (void)hipMemRelease(h); return DGE_ERR_DEVICE; }
    { std::lock_guard<std::mutex> lk(g_vm_mu); g_vm_allocs[va] = VmAlloc{sz, h}; }
    *out = va;
    return DGE_OK;
}
// frees what table_alloc (or hipMalloc) returned
static void table_free(void* p) {
    if (!p) return;
    VmAlloc a{0, {}};
    { std::lock_guard<std::mutex> lk(g_vm_mu); auto it = g_vm_allocs.find(p); if (it != g_vm_allocs.end()) { a = it->second; g_vm_allocs.erase(it); } }
    if (a.bytes) { (void)hipMemUnmap(p, a.bytes); (void)hipMemRelease(a.h); (void)hipMemAddressFree(p, a.bytes); }
    else (void)hipFree(p);
}
static int table_alloc(float** out, size_t floats, int device, hipStream_t st, int* seen, double* rate_best, double* rate_worst) {
    *out = nullptr;
    const size_t bytes = floats * sizeof(float);
    if (seen) *seen = 0;
    // small tables live in the caches: nothing to choose (and the probe needs rows to draw from)
    size_t free_b = 0, total_b = 0;
    // (and tables of 2 GiB and more are left to hipMalloc.  Round 3: the runtime aborted inside the probing of a 4.5 GB virtual-memory allocation.  Round 4: creating
    //  nine models of 2 x 3.4 GB in one process — ~100 virtual-memory allocations of 3.4 GB made, probed and released — ended twice in four runs inside this function,
    //  once as "Memory access fault by GPU node" under the probe kernel, once as an abort() of the runtime (tests/test_gpu_configs.py, full-size cfg5 on 8 ranks).
    //  The placement classes were measured on half-gigabyte tables; above 2 GiB the choice is not worth a process.)
    if (bytes < ((size_t)64 << 20) || bytes >= ((size_t)2 << 30) || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return dge_dev_alloc(out, floats);
    const int n_max = (int)std::max<size_t>(1, std::min<size_t>(TABLE_CANDIDATES, free_b / 4 / bytes));      // candidates may take a quarter of the free memory
    dge_tmp<float> sink;
    int rc = sink.alloc(4);
    if (rc) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); (void)hipGetLastError(); return dge_dev_alloc(out, floats); }
    std::vector<void*> cand; std::vector<double> rate;
    double best = 0, worst = 1e30;
    for (int k = 0; k < n_max; k++) {
        // candidates alternate between the two kinds of allocation: which kind the fast memory turns up in differs from box to box (on some every
        // hipMalloc allocation is slow and one virtual-memory allocation in two is fast, on others 24 virtual-memory allocations in a row are slow)
        void* q = nullptr;
        if (k & 1) { if (vm_alloc(&q, bytes, device) != DGE_OK) break; }       // (the virtual-memory API refused, or memory ran out: what we have, or hipMalloc below)
        else if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        double r = 0;
        bool ok = hipMemsetAsync(q, 0, bytes, st) == hipSuccess;
        for (int rep = 0; rep < 3 && ok; rep++) {
            ok = hipEventRecord(e0, st) == hipSuccess;
            hipLaunchKernelGGL(k_probe_table, dim3(4096), dim3(256), 0, st, (char*)q, (uint64_t)(bytes / 512), 64, sink.p);
            ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
            float ms = 0.f;
            if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) r = std::max(r, 65536.0 * 64 * 1024.0 / (ms * 1e-3) / 1e9);
        }
        if (!ok) { (void)hipGetLastError(); table_free(q); break; }
        cand.push_back(q); rate.push_back(r);
        best = std::max(best, r); worst = std::min(worst, r);
        if (cand.size() >= 2 && best >= 1.14 * worst) break;                   // the fast class has shown (an intermediate one, ~11 % above the slowest, exists too)
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (cand.empty()) return dge_dev_alloc(out, floats);
    size_t pick = 0;
    for (size_t k = 1; k < cand.size(); k++) if (rate[k] > rate[pick]) pick = k;
    for (size_t k = 0; k < cand.size(); k++) if (k != pick) table_free(cand[k]);
    *out = (float*)cand[pick];
    if (seen) *seen = (int)cand.size();
    if (rate_best) *rate_best = best;
    if (rate_worst) *rate_worst = worst;
    return DGE_OK;
}

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

// the run form of the table (neg_row_by_runs) against the table itself, slot by slot: the slots at which they differ
__global__ void k_runs_verify(const int32_t* __restrict__ table, int64_t T, const double* base, const uint32_t* row, const uint32_t* exc_slot,
                              const int32_t* exc_row, int n_exc, double T_inv, int64_t V, unsigned* n_bad, uint32_t* bad_slot, int32_t* bad_row, unsigned cap) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= T) return;
    const int32_t r = neg_row_by_runs((uint32_t)a, base, row, exc_slot, exc_row, n_exc, T_inv, V);
    if (r != -2 && r != table[a]) {
        const unsigned idx = atomicAdd(n_bad, 1u);
        if (idx < cap) { bad_slot[idx] = (uint32_t)a; bad_row[idx] = table[a]; }
    }
}

// the table in rank-block form (neg_table_row, sgns_kernels.h): thread (block b, word k) collects the step bits of slots 96b + 32k .. + 31
__global__ void k_table_pack(const int32_t* __restrict__ table, int64_t T, int64_t n_blocks, uint32_t* __restrict__ ctab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks * 3) return;
    const int64_t b = i / 3; const int k = (int)(i - b * 3);
    const int64_t a0 = b * DGE_CTAB_SLOTS + 32 * k;
    uint32_t bits = 0;
    int32_t prev = a0 > 0 && a0 - 1 < T ? table[a0 - 1] : 0;
    for (int j = 0; j < 32; j++) {
        const int64_t a = a0 + j;
        if (a >= T) break;
        const int32_t cur = table[a];
        if (a > b * DGE_CTAB_SLOTS && cur != prev) bits |= 1u << j;
        prev = cur;
    }
    ctab[b * 4 + 1 + k] = bits;
    if (k == 0) ctab[b * 4] = (uint32_t)(b * DGE_CTAB_SLOTS < T ? table[b * DGE_CTAB_SLOTS] : 0);
}
// and back (dge_model_table): one thread per slot
__global__ void k_table_unpack(const uint4* __restrict__ ctab, int64_t T, int32_t* __restrict__ table) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < T) table[a] = neg_table_row(ctab, (uint64_t)a);
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
// filter the sentence before windowing); len = tokens kept.  One 16-lane group per walk: lane j takes tokens j, j + 16, ... (coalesced
// 64-byte reads and writes; one thread per walk, every thread striding through its own row, ran 6.3 ms per 8 M walks of 24 tokens)
__global__ void __launch_bounds__(256) k_remap_compact(const int32_t* __restrict__ walks, int64_t n_rows, int32_t L, const int32_t* __restrict__ remap,
                                                        int32_t NV, int32_t* __restrict__ sen, int64_t* __restrict__ len_out) {
    const int lane = threadIdx.x & 15, sh = threadIdx.x & 48;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (r >= n_rows) return;
    const int32_t* in = walks + r * L;
    int32_t* out = sen + r * L;
    int len = 0;
    for (int j0 = 0; j0 < L; j0 += 16) {
        const int j = j0 + lane;
        int32_t v = -1;
        if (j < L) { const int32_t t = in[j]; if (t >= 0 && t < NV) v = remap[t]; }
        const unsigned keep = (unsigned)(__ballot(v >= 0) >> sh) & 0xFFFFu;
        if (v >= 0) out[len + __popc(keep & ((1u << lane) - 1u))] = v;
        len += __popc(keep);
    }
    for (int j = len + lane; j < L; j += 16) out[j] = -1;
    if (lane == 0) len_out[r] = len;
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

// The atomics wave in isolation (lk_post / lk_atomics_wave with its LDS accumulators of the hottest rows): the 12 workers of every workgroup post
// messages "add 1.0 to every element of these rows" — half of the rows among the n_acc hottest — and afterwards every element of row r must equal
// the number of times r was posted (counted with integer atomics; integers < 2^24 are exact in float): nothing parked in LDS may be lost or added twice.
// Messages alternate between kind 1 (syn1neg: bank 0) and kind 2 (syn0: bank 1); with div > 1 the rows are those of one block of a div-rank schedule:
// kind 1 rows = 1 (mod div), kind 2 rows = div - 1 (mod div), a slot = the row's rank inside its partition.
__global__ void __launch_bounds__(256) k_selftest_atomics_wave(float* table, unsigned long long* counts, int32_t n_rows, int stride, int iters, uint64_t seed, int n_acc, int drain, int div) {
    constexpr int DCH = 2;
    __shared__ __attribute__((aligned(16))) float s_mb[LK_MB_WORKERS * 2 * LkBox<DCH>::FLOATS];
    __shared__ int s_mb_flag[LK_MB_WORKERS * 2];
    __shared__ int s_mb_done;
    __shared__ float s_acc[2 * LK_ACC_ROWS(DCH) * DCH * 64];
    __shared__ int s_acc_cnt[2 * LK_ACC_ROWS(DCH)];
    if (threadIdx.x < LK_MB_WORKERS * 2) s_mb_flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_mb_done = 0;
    for (int i = threadIdx.x; i < 2 * LK_ACC_ROWS(DCH) * DCH * 64; i += blockDim.x) s_acc[i] = 0.f;
    if (threadIdx.x < 2 * LK_ACC_ROWS(DCH)) s_acc_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 15, wk = threadIdx.x >> 4;
    TableView tv = make_view(table, n_rows, stride);
    tv.valid = (uint32_t)stride;
    const int part_tgt = 1 % div, part_ctx = div - 1, n_part = n_rows / div;       // (rows of a partition: part, part + div, ... — n_rows >= div)
    if (wk >= LK_MB_WORKERS) {
        const int n = min(n_acc, LK_ACC_ROWS(DCH));
        lk_atomics_wave<DCH>(s_mb, s_mb_flag, &s_mb_done, LK_MB_WORKERS, tv, tv, tv, LkAcc{s_acc, s_acc_cnt, n, n, max(drain, 1), div, part_tgt, part_ctx});
        return;
    }
    unsigned n_posts = 0;
    Row<DCH> ones;
#pragma unroll
    for (int c = 0; c < DCH; c++) ones.v[c] = make_float4(1.f, 1.f, 1.f, 1.f);
    const int64_t worker = (int64_t)blockIdx.x * LK_MB_WORKERS + wk;
    for (int it = 0; it < iters; it++) {
        int32_t row = -1;
        if (lane < NEG_BATCH) {
            const uint64_t hsh = dge_mix64(seed + (uint64_t)((worker * iters + it) * 16 + lane));
            const int32_t rank = (int32_t)((hsh & 1ull) ? (hsh >> 1) % (uint64_t)min(8, n_part) : (hsh >> 1) % (uint64_t)n_part);
            row = rank * div + ((it & 1) ? part_ctx : part_tgt);
            atomicAdd(&counts[row], 1ULL);
        }
        lk_post<DCH>(s_mb, s_mb_flag, wk, n_posts, (it & 1) ? 2 : 1, row, 1.0f, ones, lane);
    }
    if (lane == 0) __hip_atomic_fetch_add(&s_mb_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

extern "C" int dge_selftest_atomics_wave(int device, int32_t n_rows, int32_t n_acc, int32_t drain, int32_t blocks, int32_t iters, uint64_t seed,
                                         int64_t* total_updates, double* max_abs_error) {
    return dge_selftest_atomics_wave_block(device, n_rows, n_acc, drain, 1, blocks, iters, seed, total_updates, max_abs_error);
}
extern "C" int dge_selftest_atomics_wave_block(int device, int32_t n_rows, int32_t n_acc, int32_t drain, int32_t div, int32_t blocks, int32_t iters, uint64_t seed,
                                               int64_t* total_updates, double* max_abs_error) {
    if (n_rows <= 0 || blocks <= 0 || iters <= 0 || n_acc < 0 || drain <= 0 || div <= 0 || n_rows < div || !total_updates || !max_abs_error)
        DGE_FAIL(DGE_ERR_ARG, "dge_selftest_atomics_wave: bad argument");
    int rc = dge_require_device(device);
    if (rc) return rc;
    const int stride = 128;
    float* d_tab = nullptr; unsigned long long* d_cnt = nullptr;
    if ((rc = dge_dev_alloc(&d_tab, (size_t)n_rows * stride))) return rc;
    if ((rc = dge_dev_alloc(&d_cnt, (size_t)n_rows))) return rc;
    DGE_HIP(hipMemset(d_tab, 0, (size_t)n_rows * stride * sizeof(float)));
    DGE_HIP(hipMemset(d_cnt, 0, (size_t)n_rows * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest_atomics_wave, dim3((unsigned)blocks), dim3(256), 0, 0, d_tab, d_cnt, n_rows, stride, iters, seed, n_acc, drain, div);
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipDeviceSynchronize());
    std::vector<float> tab((size_t)n_rows * stride); std::vector<unsigned long long> cnt((size_t)n_rows);
    DGE_HIP(hipMemcpy(tab.data(), d_tab, tab.size() * sizeof(float), hipMemcpyDeviceToHost));
    DGE_HIP(hipMemcpy(cnt.data(), d_cnt, cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    dge_dev_free(d_tab); dge_dev_free(d_cnt);
    double worst = 0.0; int64_t total = 0;
    for (int32_t r = 0; r < n_rows; r++) {
        total += (int64_t)cnt[(size_t)r];
        for (int c = 0; c < stride; c++) worst = std::max(worst, fabs((double)tab[(size_t)r * stride + c] - (double)cnt[(size_t)r]));
    }
    *total_updates = total; *max_abs_error = worst;
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
    table_free(m->d_syn0); table_free(m->d_syn1neg); dge_dev_free(m->d_locks); dge_dev_free(m->d_ctab);
    dge_dev_free(m->d_run_base); dge_dev_free(m->d_run_row); dge_dev_free(m->d_exc_slot); dge_dev_free(m->d_exc_row);
    dge_dev_free(m->d_snap); dge_dev_free(m->d_vocab_ids);
    table_free(m->d_syn1); dge_dev_free(m->d_hs_off); dge_dev_free(m->d_hs_points); dge_dev_free(m->d_hs_codes);
    dge_dev_free(m->d_counts); dge_dev_free(m->d_remap); dge_dev_free(m->d_exp);
    dge_dev_free(m->d_sen); dge_dev_free(m->d_len); dge_dev_free(m->d_wb); dge_dev_free(m->d_scan_tmp); dge_dev_free(m->d_counters);
    dge_sorted_release(m);
    if (m->ev_peer) (void)hipEventDestroy(m->ev_peer);
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
    if (cfg->update_policy < 0 || cfg->update_policy == 4 || cfg->update_policy > 8) DGE_FAIL(DGE_ERR_ARG, "dge_model_create: unknown update_policy %d", cfg->update_policy);
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
    // the two tables and their lock words first: before the unigram table and the 0.8 GB of temporaries its construction takes
    const size_t tab = (size_t)V * (size_t)m->stride;
    m->ctab_blocks = (m->T + DGE_CTAB_SLOTS - 1) / DGE_CTAB_SLOTS;
    // syn1neg first: fast memory is scarce on some boxes, and a pair touches K + 1 rows of syn1neg for one of syn0 (syn1neg in fast memory and syn0 in slow:
    // 415 ms per launch; the other way round 423-456).  The lock words and the negative-sampling table are too small for the probe to classify (cache resident
    // when probed alone) and do NOT share a table's allocation: a table with them appended (511 MiB instead of 487) never landed in fast memory in 32 tries,
    // on any of three boxes — allocations of 511 ... 520 MiB never do (scripts/micro/size_class.hip) — so their placement is left to dge_model_tune_placement.
    MC(table_alloc(&m->d_syn1neg, tab + 64, device, st, &m->placed_seen[1], &m->placed_best[1], &m->placed_worst[1]));
    MC(table_alloc(&m->d_syn0, tab + 64, device, st, &m->placed_seen[0], &m->placed_best[0], &m->placed_worst[0]));
    MC(dge_dev_alloc(&m->d_locks, 2 * ((size_t)V + 1)));      // [0,V]: syn1neg rows, [V+1,2V+1]: syn0 rows
    MC(dge_dev_alloc(&m->d_ctab, (size_t)m->ctab_blocks + 1));
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
    dge_tmp<int32_t> d_flat;                                 // word2vec's one-row-per-slot table: a temporary, the trainers read its rank-block form
    MC(d_flat.alloc((size_t)m->T));
    if (V > 0) {
        std::vector<double> cum((size_t)V);
        double twp = 0.0; const double power = 0.75;
        for (int64_t i = 0; i < V; i++) twp += pow((double)m->h_counts[(size_t)i], power);
        {
            double s2 = 0.0;
            for (int64_t i = 0; i < V; i++) { double q = pow((double)m->h_counts[(size_t)i], power) / twp; s2 += q * q; }
            m->neg_collision = s2; m->neg_norm = twp;
            m->row_share_max = std::max((double)m->h_counts[0] / (double)std::max<int64_t>(tw, 1), pow((double)m->h_counts[0], power) / twp);
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
        hipLaunchKernelGGL(k_table_fill, dim3(grid_for(m->T, 256)), dim3(256), 0, st, d_m.p, V, m->T, d_flat.p);
        MH(hipStreamSynchronize(st));
        // --- the run form of the table, if this vocabulary has one (neg_row_by_runs): runs of adjacent rows of equal count, checked against the table
        if (m->T < 0xFFFFFFFFll && g_dge_tuning[DGE_TUNE_TABLE_RUNS] != 0) {
            std::vector<double> rb(DGE_RUN_MAX, std::numeric_limits<double>::infinity());
            std::vector<uint32_t> rr(DGE_RUN_MAX + 1, (uint32_t)V);
            // the runs are collected from the vocabulary's END — the tail is where equal counts abound — until the arrays are full; head rows in front of the
            // first run keep the table look-up (cfg3: 1 716 runs cover its whole vocabulary of a million rows, counts 2 .. 6 905, and match the table in
            // every one of its 1e8 slots; a vocabulary of several thousand distinct counts keeps its most frequent rows on the table)
            int runs = 0; int64_t i = V;
            std::vector<int64_t> starts;
            const int max_runs = g_dge_tuning[DGE_TUNE_TABLE_RUNS] > 0 ? (int)std::min<int64_t>(g_dge_tuning[DGE_TUNE_TABLE_RUNS], DGE_RUN_MAX - 2) : DGE_RUN_MAX - 2;
            while (i > 0 && runs < max_runs) {
                int64_t j = i - 1;
                while (j > 0 && m->h_counts[(size_t)j - 1] == m->h_counts[(size_t)i - 1]) j--;
                starts.push_back(j); runs++; i = j;
            }
            std::reverse(starts.begin(), starts.end());
            for (int r = 0; r < runs; r++) {
                const int64_t r0 = starts[(size_t)r];
                rb[(size_t)r] = r0 == 0 ? 0.0 : cum[(size_t)r0 - 1];
                rr[(size_t)r] = (uint32_t)r0;
            }
            rb[(size_t)runs] = cum[(size_t)V - 1];       // the boundary behind the last run (behind it: +inf, row V)
            {   // whether the run form is kept is decided by comparing it with the table
                dge_tmp<unsigned> d_nbad; dge_tmp<uint32_t> d_bs; dge_tmp<int32_t> d_br;
                const unsigned cap = 4096;
                MC(d_nbad.alloc(1)); MC(d_bs.alloc(cap)); MC(d_br.alloc(cap));
                MC(dge_dev_alloc(&m->d_run_base, DGE_RUN_MAX)); MC(dge_dev_alloc(&m->d_run_row, DGE_RUN_MAX + 1));
                MC(dge_dev_alloc(&m->d_exc_slot, DGE_RUN_EXC)); MC(dge_dev_alloc(&m->d_exc_row, DGE_RUN_EXC));
                MH(hipMemcpyAsync(m->d_run_base, rb.data(), DGE_RUN_MAX * sizeof(double), hipMemcpyHostToDevice, st));
                MH(hipMemcpyAsync(m->d_run_row, rr.data(), (DGE_RUN_MAX + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                std::vector<uint32_t> es(DGE_RUN_EXC, 0xFFFFFFFFu); std::vector<int32_t> er(DGE_RUN_EXC, 0);
                unsigned n_bad = 0; int n_exc = 0; bool good = false;
                for (int round = 0; round < 2; round++) {      // first without exceptions (collects them), then with them (must leave nothing)
                    MH(hipMemsetAsync(d_nbad.p, 0, sizeof(unsigned), st));
                    MH(hipMemcpyAsync(m->d_exc_slot, es.data(), DGE_RUN_EXC * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    MH(hipMemcpyAsync(m->d_exc_row, er.data(), DGE_RUN_EXC * sizeof(int32_t), hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(k_runs_verify, dim3(grid_for(m->T, 256)), dim3(256), 0, st, d_flat.p, m->T, m->d_run_base, m->d_run_row, m->d_exc_slot, m->d_exc_row,
                                       n_exc, 1.0 / (double)m->T, V, d_nbad.p, d_bs.p, d_br.p, cap);
                    MH(hipMemcpyAsync(&n_bad, d_nbad.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    MH(hipStreamSynchronize(st));
                    if (round == 1) { good = n_bad == 0; break; }
                    if (n_bad > DGE_RUN_EXC) break;
                    if (n_bad > 0) {
                        std::vector<uint32_t> bs(n_bad); std::vector<int32_t> br(n_bad);
                        MH(hipMemcpy(bs.data(), d_bs.p, n_bad * sizeof(uint32_t), hipMemcpyDeviceToHost));
                        MH(hipMemcpy(br.data(), d_br.p, n_bad * sizeof(int32_t), hipMemcpyDeviceToHost));
                        std::vector<size_t> order(n_bad);
                        for (size_t q = 0; q < n_bad; q++) order[q] = q;
                        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return bs[x] < bs[y]; });
                        for (size_t q = 0; q < n_bad; q++) { es[q] = bs[order[q]]; er[q] = br[order[q]]; }
                    }
                    n_exc = (int)n_bad;
                }
                if (good) { m->n_runs = runs; m->n_exc = n_exc; }
            }
        }
    } else {
        MH(hipMemsetAsync(d_flat.p, 0, (size_t)m->T * sizeof(int32_t), st));
    }
    hipLaunchKernelGGL(k_table_pack, dim3(grid_for(m->ctab_blocks * 3, 256)), dim3(256), 0, st, d_flat.p, m->T, m->ctab_blocks, (uint32_t*)m->d_ctab);
    MH(hipStreamSynchronize(st));

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
    MH(hipMemsetAsync(m->d_syn1neg, 0, (tab + 64) * sizeof(float), st));
    if (V) hipLaunchKernelGGL(k_init_syn0, dim3(grid_for(V, 256)), dim3(256), 0, st, m->d_syn0, V, m->D, m->stride, cfg->seed);
    if (cfg->use_hs) {
        // inner-node table (V rows allocated, V-1 used: the tables stay the same size for the delta exchange) and paths
        std::vector<int64_t> node_w;
        const int longest = dge_huffman_paths(m->h_counts.data(), V, m->h_hs_off, m->h_hs_points, m->h_hs_codes, &node_w);
        // cold inner nodes: on fewer than 2e-5 of all paths (their weights ascend with the node number: a prefix)
        { int64_t tot = 0; for (int64_t c : m->h_counts) tot += c;
          const int64_t limit = (int64_t)((double)tot * 2e-5);
          m->hs_cold_auto = (int32_t)(std::upper_bound(node_w.begin(), node_w.end(), limit) - node_w.begin());
          // busy inner nodes: on a tenth of all paths and more (a suffix; the balanced tree of a flat vocabulary has 15 of them, a skewed one a few more) — at most 64
          const int64_t busy = (int64_t)((double)tot * 0.1);
          m->hs_rep_auto = (int32_t)std::min<int64_t>(64, node_w.end() - std::lower_bound(node_w.begin(), node_w.end(), busy));
          for (int k = 0; k < 32; k++)     // first node on more than k/32 of all paths (weights ascend with the node number)
              m->hs_rep_thr32[k] = (int32_t)(std::upper_bound(node_w.begin(), node_w.end(), (int64_t)((double)tot * k / 32.0)) - node_w.begin()); }
        // (behind the table: HS_REP_ROWS spare rows for k_sgns_train_hsw's copies of the busiest inner nodes — zero between launches)
        MC(table_alloc(&m->d_syn1, tab + 64 + (size_t)HS_REP_ROWS * m->stride, device, st, &m->placed_seen[2], &m->placed_best[2], &m->placed_worst[2]));
        MH(hipMemsetAsync(m->d_syn1, 0, (tab + 64 + (size_t)HS_REP_ROWS * m->stride) * sizeof(float), st));
        if (longest > 40) { model_release(m); delete m; DGE_FAIL(DGE_ERR_ARG, "dge_model_create: a Huffman code of %d bits exceeds word2vec's MAX_CODE_LENGTH 40", longest); }
        MC(dge_dev_alloc(&m->d_hs_off, (size_t)V + 1)); MC(dge_dev_alloc(&m->d_hs_points, m->h_hs_points.size())); MC(dge_dev_alloc(&m->d_hs_codes, (size_t)V));
        MH(hipMemcpyAsync(m->d_hs_off, m->h_hs_off.data(), ((size_t)V + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        if (!m->h_hs_points.empty()) MH(hipMemcpyAsync(m->d_hs_points, m->h_hs_points.data(), m->h_hs_points.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (V) MH(hipMemcpyAsync(m->d_hs_codes, m->h_hs_codes.data(), (size_t)V * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    }
    MH(hipMemsetAsync(m->d_locks, 0, 2 * ((size_t)V + 1) * sizeof(int), st));
    MC(dge_dev_alloc(&m->d_counters, 8));       // pairs, words, the lock kernels' walk counter, their watchdog flag; the block kernels' lock statistics: pairs put back, rounds that left rows unwon, rounds
    MH(hipMemsetAsync(m->d_counters, 0, 8 * sizeof(unsigned long long), st));
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
    m->cap_rows = nr; m->cap_L = nl; m->seen_gen = 0;
    return DGE_OK;
}

// the trainer kernels are compiled per row width in sgns_train_dch.hip (one object per DCH)
void dge_launch_train_dch1(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st);
void dge_launch_train_dch2(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st);
void dge_launch_train_dch3(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st);
void dge_launch_train_dch4(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st);
void dge_launch_train_dch6(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st);
void dge_launch_train_dch8(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st);

// Launch timing: a pair of events per kernel launch, read by dge_model_stats.  A host that trains in a long loop without asking for
// stats must not pile up event handles: launches of one model complete in stream order, so once 64 pairs are pending the finished
// ones at the front are folded into the totals and destroyed.
static void reap_finished_events(dge_model* m) {
    size_t n = 0;
    while (n < m->pending.size() && hipEventQuery(m->pending[n].b) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->pending[n].a, m->pending[n].b) == hipSuccess) { if (m->pending[n].kind == 0) m->kernel_ms += ms; else m->walk_ms += ms; }
        (void)hipEventDestroy(m->pending[n].a); (void)hipEventDestroy(m->pending[n].b);
        n++;
    }
    m->pending.erase(m->pending.begin(), m->pending.begin() + (ptrdiff_t)n);
}
static int timing_begin(dge_model* m, EventPair& ev, int kind) {
    if (m->pending.size() >= 64) reap_finished_events(m);
    ev.kind = kind; ev.a = nullptr; ev.b = nullptr;
    DGE_HIP(hipEventCreate(&ev.a));
    if (hipEventCreate(&ev.b) != hipSuccess) { (void)hipEventDestroy(ev.a); DGE_FAIL(DGE_ERR_DEVICE, "hipEventCreate failed"); }
    if (hipEventRecord(ev.a, m->stream) != hipSuccess) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); DGE_FAIL(DGE_ERR_DEVICE, "hipEventRecord failed"); }
    return DGE_OK;
}
static int timing_end(dge_model* m, EventPair& ev, int rc_so_far) {
    if (rc_so_far != DGE_OK || hipEventRecord(ev.b, m->stream) != hipSuccess) {
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b);
        if (rc_so_far != DGE_OK) return rc_so_far;
        DGE_FAIL(DGE_ERR_DEVICE, "hipEventRecord failed");
    }
    m->pending.push_back(ev);
    return DGE_OK;
}

// Head of the vocabulary inside ONE BLOCK of an n-rank block schedule (dge_model_set_partition).  A block's live rows are the V / n rows of its
// partition, and a row of the partition takes n times its share of the block's accesses (negatives drawn from the whole table are moved to the
// partition's row nearest below: n rows' worth of draws; contexts: the pairs whose context lies in the partition), so over the locked rows the
// expected failures per try-lock are W * 5 * n^2 * sum_{partition}(q_i^2 + p_i^2) ~ W * 5 * n * sum_{all}(q_i^2 + p_i^2): the rule of
// dge_model_create with the bound divided by n.  And a context row whose own pairs — serialised by its lock — are more than half of what one worker
// trains in the launch (W * n * p_i > 0.5) goes to the atomics side as well.  Rows [0, head) take atomics, the rest stay under the commit locks.
// Round 5, with the hottest rows' chains gone (the accumulator banks), the head size was swept again per graph (profiles/r05_skewed_knobs_*.txt, r05_blocks_head_quality_*.txt):
// a block's speed has a flat optimum — cfg3_zipf 20 000 .. 160 000 rows within 2 %, cfg5 15 000 .. 60 000 (best 30 000: 8.47e7 against 8.31e7 at the 60 320 of the
// 0.1 bound), the cfg3-sized community graph with a Zipf fifth 20 000 .. 40 000 (10 % faster than the 183 340 of the 0.1 bound) — and the embedding does not depend on it
// (AUC / loss equal to the fourth digit from 10 000 to 183 340 rows).  A bound of 0.2 puts all three inside their optimum (44 222 / ~30 000 / 60 380 rows).
static int64_t block_head(dge_model* m, int n, int64_t W) {
    if (m->block_head_n == n && m->block_head_workers == W) return m->block_head_rows;
    const double power = 0.75, twp = m->neg_norm, tw = (double)std::max<int64_t>(m->total_words, 1);
    double tail = 0.0; int64_t H = m->V;
    const double bound = 0.2 / (5.0 * (double)W * (double)std::max(n, 1));     // (round 5: 0.1 until the accumulator banks; see below)
    while (H > 0) {
        const double c = (double)m->h_counts[(size_t)(H - 1)];
        const double q = pow(c, power) / twp, pp = c / tw;
        if (tail + q * q + pp * pp >= bound) break;
        tail += q * q + pp * pp; H--;
    }
    int64_t Hs = 0;
    while (Hs < m->V && (double)W * (double)n * (double)m->h_counts[(size_t)Hs] / tw > 0.5) Hs++;
    m->block_head_n = n; m->block_head_workers = W; m->block_head_rows = std::max(H, Hs);
    return m->block_head_rows;
}

// What update_policy 0 resolves to for a device-filling launch over the whole vocabulary on one GPU — 5 (commit locks), 7 (locks, the head by atomics) or 2
// (atomics); the owner-computes schedule (8) is taken instead of 2 where it applies (train_rows).  The rule's constants were fitted on the four bench graphs
// and then checked — and moved — against a sweep of vocabulary size x popularity exponent x row width (scripts/policy_sweep.py, profiles/r04_policy_sweep.txt):
//   * the commit locks are the fast schedule while a try-lock rarely fails: expected failures per attempt ~ workers * 5 * sum q_i^2 < 0.4 on a vocabulary of
//     >= 131 072 rows (round 3: 0.25 and 262 144 — a flat 200 000-row vocabulary runs 1.46e9 edges/s under locks against 1.19e9 owner-computes);
//   * a skewed vocabulary keeps the locks for its tail when the head that has to leave them is at most a quarter of the rows (round 3: an eighth — rank^-0.5
//     popularity over 300 000 rows: 9.5e8 against 7.3e8 owner-computes);
//   * when the busiest row's share caps the workers below a quarter of the device (48 / its share of the tokens < 4096 workers; train_rows then caps the workers at 96 in flight), the lock protocol has
//     nothing to win over atomics (rank^-1 over 300 000 words: 1.37e8 against 5.9e7).
// (the first two conditions alone: a try-lock on a syn1neg row rarely fails — what the hierarchical-softmax kernel's lock form needs; it never locks a context row,
//  so a vocabulary with a handful of rows whose OWN pairs would serialise under a syn0 lock, policy 7 with a tiny head, takes it as well)
static bool syn1neg_locks_work(const dge_model* m) {
    const double fail = (double)((int64_t)m->n_cus * 3 * 16) * 5.0 * m->neg_collision;
    return (int64_t)(48.0 / std::max(m->row_share_max, 1e-12)) >= 4096 && m->V >= 131072 && fail < 0.4;
}
static int auto_policy(const dge_model* m, bool hs) {
    if (hs) return 2;
    const double fail = (double)((int64_t)m->n_cus * 3 * 16) * 5.0 * m->neg_collision;
    if ((int64_t)(48.0 / std::max(m->row_share_max, 1e-12)) < 4096) return 2;
    if (m->V >= 131072 && fail < 0.4) return m->hot_rows_serial > 0 ? 7 : 5;
    if (m->V >= 131072 && m->hot_rows_auto <= m->V / 4) return 7;
    return 2;
}

// k_sgns_train_hsw's copies of the busiest inner nodes: row j of `rows` += its HS_REP - 1 copies, which are cleared (one workgroup a node; runs behind the trainer)
__global__ void k_hs_rep_fold(float* rows, float* rep, int32_t n_rep, int32_t stride) {
    const int j = blockIdx.x;
    for (int e = threadIdx.x; e < stride; e += blockDim.x) {
        float v = rows[(size_t)j * stride + e];
        for (int c = 0; c < HS_REP - 1; c++) { float* q = rep + ((size_t)c * n_rep + j) * stride + e; v += *q; *q = 0.f; }
        rows[(size_t)j * stride + e] = v;
    }
}

static int train_rows(dge_model* m, const int32_t* d_rows, int64_t n_rows, int32_t L, int64_t walk_index_base, int32_t epoch,
                      int64_t words_before, double words_scale, int64_t total_walks, uint64_t corpus_gen) {
    if (n_rows == 0 || m->V == 0) return DGE_OK;
    int rc = ensure_work(m, n_rows, L);
    if (rc) return rc;
    hipStream_t st = m->stream;
    // vocabulary rows of the walks, left-packed, and the words that precede each walk: kept while the same unchanged rows come again
    // (the N episodes of a block-schedule batch train the same walks N times)
    if (!(m->seen_rows == d_rows && m->seen_n == n_rows && m->seen_L == L && m->seen_gen == corpus_gen && corpus_gen != 0)) {
        hipLaunchKernelGGL(k_remap_compact, dim3(grid_for(n_rows * 16, 256)), dim3(256), 0, st, d_rows, n_rows, L, m->d_remap, m->NV, m->d_sen, m->d_len);
        size_t bytes = m->scan_tmp_bytes;
        DGE_HIP(hipcub::DeviceScan::ExclusiveSum(m->d_scan_tmp, bytes, m->d_len, m->d_wb, n_rows, st));
        m->seen_rows = d_rows; m->seen_n = n_rows; m->seen_L = L; m->seen_gen = corpus_gen;
    }

    TrainParams p;
    p.sen = m->d_sen; p.len = m->d_len; p.wb = m->d_wb;
    p.syn0 = m->d_syn0; p.syn1neg = m->d_syn1neg; p.exp_table = m->d_exp;
    p.ctab = m->d_ctab;
    p.run_base = m->d_run_base; p.run_row = m->d_run_row; p.exc_slot = m->d_exc_slot; p.exc_row = m->d_exc_row;
    p.n_runs = g_dge_tuning[DGE_TUNE_TABLE_RUNS] == 0 ? 0 : m->n_runs; p.n_exc = m->n_exc; p.T_inv = 1.0 / (double)std::max<int64_t>(m->T, 1);
    p.n_rows = n_rows; p.L = L; p.W = m->cfg.window; p.K = m->cfg.negative; p.stride = m->stride;
    p.V = m->V; p.T = m->T; p.seed = m->cfg.seed;
    p.T_magic = ~0ull / (uint64_t)std::max<int64_t>(m->T, 1); p.W_magic = ~0ull / (uint64_t)std::max(m->cfg.window, 1);
    p.gidx_base = (int64_t)epoch * total_walks + walk_index_base;
    p.words_done_base = (int64_t)epoch * m->total_words + words_before;
    p.all_words = (int64_t)std::max(m->cfg.epochs, 1) * m->total_words;
    p.words_scale = words_scale;
    p.alpha0 = m->cfg.alpha; p.min_alpha = m->cfg.min_alpha;
    p.D = m->D;
    p.counters = m->d_counters;
    p.next_walk = nullptr; p.wd_ticks = 0;
    p.locks = m->d_locks;
    p.syn1 = m->d_syn1; p.hs_off = m->d_hs_off; p.hs_points = m->d_hs_points; p.hs_codes = m->d_hs_codes;
    p.hs_hot0 = 0x7fffffff; p.hs_n_hot = 0; p.hs_drain = 1; p.hot_rows = 0; p.hs_cold = 0; p.hs_wave = 0;
    p.hs_rep0 = 0x7fffffff; p.hs_rep_n = 0; for (int k = 0; k < HS_REP; k++) p.hs_rep_thr[k] = 0x7fffffff;
    p.part_n = m->part_n; p.part_ctx = m->part_ctx; p.part_tgt = m->part_tgt; p.syn0_free = 0; p.acc_rows = 0; p.acc_drain = 16;
    p.N_magic = 0xFFFFFFFFu / (uint32_t)std::max(m->part_n, 1);
    p.big_seg_shift = 0;
    p.filler_row = (int32_t)(0xFFFFFFF0u / ((uint32_t)m->stride * 4u)) - 1;      // offset + the largest in-row displacement stays below 2^32
    if (g_dge_tuning[DGE_TUNE_SEGMENT_SHIFT] >= 0) p.big_seg_shift = (int32_t)g_dge_tuning[DGE_TUNE_SEGMENT_SHIFT];   // tests: several segments on a small table
    const bool part = m->part_n > 1;
    const bool hs = m->cfg.use_hs != 0;

    // auto: where the Hogwild kernels are bound by contention on a FLAT vocabulary — one too small for row locks (they fall back to
    // atomics: cfg2), a block of a schedule of 3 and more ranks (V/N live rows per table) — the owner-computes schedule is the faster
    // one at equal link-prediction AUC (one block of 8 ranks 5.7e8 vs 4.6e8 edges/s, a 100 k-row vocabulary at D = 128 6.8e8 vs 4.2e8:
    // profiles/r02_quality_sorted.txt).  A skewed vocabulary stays with the mixed policy 7: a synchronous mini-batch hands a hot row
    // thousands of terms at once with no feedback between them, and the embedding diverges (dge_sorted_batch_items).
    bool sorted_auto = false;
    if (m->cfg.update_policy == 0 && m->cfg.workers == 0 && !hs && (uint64_t)m->V * (uint64_t)m->stride * 4ull < 0xFFFFFFFFull) {
        const bool locks_work = auto_policy(m, hs) != 2;                                            // -> commit locks, all rows (5) or the tail (7)
        if (part) sorted_auto = m->part_n >= 2 && dge_sorted_batch_items(m, m->part_n) > 0;         // (per rank on cfg3, owner-computes vs locks, round 3 with the items made once per batch: N = 2 7.5e8 vs 7.2e8, N = 4 7.8e8 vs 7.4e8, N = 8 7.5e8 vs 4.8e8)
        else sorted_auto = !locks_work && dge_sorted_batch_items(m, 1) > 0;                         // (what used to fall back to atomics)
    }
    const bool allow_unsafe = g_dge_tuning[DGE_TUNE_ALLOW_UNSAFE] > 0;
    const double launch_pairs = (double)n_rows * dge_expected_pairs_per_walk(L, m->cfg.window) / ((double)m->part_n * (double)m->part_n);      // (full-length walks: an upper estimate)
    if ((m->cfg.update_policy == 8 && m->cfg.workers != 1) || sorted_auto) {
        // owner-computes schedule (sgns_sorted.hip): items sorted by row, no locks, no atomics, deterministic
        if (hs) DGE_FAIL(DGE_ERR_ARG, "update_policy 8 does not carry the hierarchical-softmax term");
        // FORCED on a vocabulary the rule would not pick it for (dge_sorted_batch_items == 0): a small vocabulary is merely slow, but on a skewed one the busiest row takes
        // thousands of terms of one synchronous mini-batch with no feedback between them and the tables go to NaN within an epoch (profiles/r02_quality_zipf_sorted_diverges.txt;
        // rank^-1 over 50 000 words: 9 % of a million-item mini-batch on one row).  Refused instead (round 5; a mini-batch size set by hand — DGE_TUNE_SORTED_WALKS — is the caller's business).
        if (!sorted_auto && !allow_unsafe && g_dge_tuning[DGE_TUNE_SORTED_WALKS] <= 0 && dge_sorted_batch_items(m, m->part_n) == 0) {
            const double hottest = std::min(1.0, m->row_share_max * (double)std::max(m->part_n, 1));
            const double mb_items = std::min((double)(1 << 20), launch_pairs * (double)(m->cfg.negative + 1));
            if (hottest * mb_items > 8192.0)
                DGE_FAIL(DGE_ERR_ARG, "update_policy 8 (owner-computes) on this vocabulary: its busiest row holds %.2g of the terms, ~%.0f of one synchronous mini-batch "
                         "with no feedback between them (the schedule keeps that below 4096; far beyond, the tables diverge): use update_policy 0 (auto), 2 or 7", hottest, hottest * mb_items);
        }
        EventPair ev;
        if ((rc = timing_begin(m, ev, 0))) return rc;
        if ((rc = timing_end(m, ev, dge_sorted_train(m, p)))) return rc;
        m->launches++;
        m->last_policy = 8; m->last_workers = 0; m->last_hot_rows = 0;
        m->last_kernel = part ? "k_sorted_phase (owner-computes, one block of the multi-GPU schedule: k_block_emit + 2 item sorts + 2 phases)" : "k_sorted_phase (owner-computes: k_sorted_emit + 2 item sorts + 2 phases)";
        return DGE_OK;
    }
    int64_t workers;
    if (m->cfg.workers == 0) {
        // fill the device: 4 blocks of 16 workers per CU, but never more concurrent walks than half the vocabulary
        // (Hogwild's premise is sparse collisions: measured, a 2.3k-row table keeps 0.99 cosine to the in-order
        // result up to ~1k workers and loses it beyond; the reference ran 8 workers on <= 6.4k rows)
        const int auto_pol = m->cfg.update_policy == 0 ? auto_policy(m, hs) : 0;
        const bool auto_locked = auto_pol == 5, auto_mixed = auto_pol == 7;
        const int blocks_per_cu = (m->cfg.update_policy == 5 || m->cfg.update_policy == 6 || m->cfg.update_policy == 7 || auto_locked || auto_mixed) ? ((m->cfg.update_policy == 7 || auto_mixed) ? (m->stride <= 128 ? DGE_HOTMIX_WAVES : 2) : (m->stride == 64 ? 4 : DGE_LOCKED_WAVES)) : 4;   // (rows of one chunk leave room for a 4th wave per SIMD in the lock kernel; a 5th under atomics gains nothing: cfg2 7.6e8 either way)   // what the kernel's VGPR budget keeps resident
        workers = (int64_t)m->n_cus * blocks_per_cu * 16;
        // Round 5 (scripts/small_vocab_workers.py, profiles/r05_small_vocab_workers*.txt): measured again on the reference's own tract size — 6 408 rows, D = 20, a graph with
        // community structure — with what matters downstream instead of the cosine to the in-order result: held-out link AUC and loss.  Negative sampling: AUC 0.9295 at
        // 3 204 ... 16 384 workers alike (sequential oracle 0.9294, its 8 Hogwild threads 0.9274), loss 0.6928 -> 0.6942 at 9 612 (sequential 0.6924, 8 threads 0.7111);
        // with the hierarchical softmax 9 612 workers keep AUC 0.9213 / loss 0.737 (8 CPU threads: 0.9207 / 0.741) and 12 816 lose it (0.915 / 0.81).  So from 4 096
        // rows on the cap is 1.5 workers a row: cfg1 7.2e8 -> 1.28e9 edges/s, with the tree term 1.64e8 -> 5.1e8.  Below 4 096 rows the round-1 cap stays.
        // The small-row kernel (rows of 17 .. 32 floats without the tree term: k_sgns_train_small) reaches its request-rate ceiling with ONE worker a row — 1.41e9 edges/s
        // at 6 408, 9 612 and 16 384 workers alike, loss 0.6935 / 0.6943 / 0.716 (profiles/r05_small_row_kernel.txt) — so it runs one a row.
        const bool small_kernel = !hs && m->part_n <= 1 && m->cfg.dim > 16 && m->cfg.dim <= 32 && m->stride == 64 && g_dge_tuning[DGE_TUNE_SMALL_ROWS] != 0;
        workers = std::min(workers, std::max<int64_t>(64, m->V >= 4096 ? (small_kernel ? m->V : m->V * 3 / 2) : m->V / 2));
        // ... nor so many that ONE row has dozens of its updates in flight at once: every one of them is computed from the same stale row, and their
        // sum — along the direction the contexts share — is a gradient step M times too long.  A vocabulary whose busiest row takes 9 % of the tokens
        // (Zipf(1) over 50 000 words: text without sub-sampling, not a flow graph) went to NaN within one launch of 16 384 workers
        // (scripts/policy_sweep.py, round 4); cfg3 with Zipf destinations and cfg5 keep 34 and 18 in flight and train to the atomics-free AUC.
        // Round 5 swept that bound on a graph WITH structure whose busiest vertex holds 2.8 % of the tokens (scripts/hot_row_inflight.py, profiles/r05_hot_row_inflight.txt;
        // sequential oracle AUC 0.7472 / loss 1.988, its 8 Hogwild threads 0.7346 / 2.044): 24 / 48 / 96 in flight 0.7469 / 0.7464 / 0.7454 at loss 1.97, 192: 0.7402 / 2.00,
        // 384: 0.7333 / 2.08 — and the SPEED peaks at 96 (3.2e8 edges/s; 2.7e8 at 48, 3.0e8 at 192: beyond, the busiest rows' atomics queue at the memory side).  So: 96.
        // Copies of the hottest rows (what k_sgns_train_hsw does for the Huffman root) would lift the atomic wall, not this one: staleness caps the in-flight count first.
        workers = std::min(workers, std::max<int64_t>(64, (int64_t)(96.0 / std::max(m->row_share_max, 1e-12))));
        workers = std::min(workers, (n_rows + 15) / 16 * 16);
    } else workers = m->cfg.workers;
    if (m->cfg.workers == 0 && g_dge_tuning[DGE_TUNE_WORKERS] > 0) workers = std::min<int64_t>(g_dge_tuning[DGE_TUNE_WORKERS], (n_rows + 15) / 16 * 16);     // ablation knob
    p.n_workers = workers;
    // update policy (see Policy<>, k_sgns_train_locked and dge_train_config.update_policy)
    int pol = m->cfg.update_policy;
    if (pol == 0) {
        // auto.  The commit-lock kernel is the fast one while lock attempts rarely fail: a try fails when the row is among
        // the ~5 rows another worker holds, i.e. with probability ~ workers * 5 * sum_i q_i^2 (q = unigram^0.75 sampling
        // probabilities).  cfg3 (uniform-ish, 1M rows, 12k workers): 0.07 -> locked, 8.9e8 edges/s.  A Zipf-popular
        // vocabulary (cfg5) gives >> 1: the same kernel spins on its hot rows (measured 5e5 edges/s) while memory-side
        // atomics are indifferent to the skew (5.9e7 = their byte rate) -> atomics.
        // In between (a skewed head over a long tail — cfg5, and what real trip data looks like) the head rows alone are
        // taken out of the lock protocol: policy 7.  (auto_policy above)
        pol = workers == 1 ? 100 : auto_policy(m, hs);
    }
    // FORCED commit locks (5 / 6) on a vocabulary with a busy row: a pair holds its context row's lock for its whole duration, so the pairs of row i run one behind
    // the other — p_i x pairs of them while the launch as a whole should last pairs / W pair-times — and the waiting workers keep hammering that lock word: measured
    // (profiles/r04_policy_sweep.txt) 60x slower at W x p = 6 (rank^-0.5 over 1e6 rows), "minutes" on rank^-1.  Auto moves such rows to the atomics side (7); a forced 5 / 6 is
    // refused beyond W x p = 2 (the community graph's W x p = 1.2 runs 1.8x slower: still a choice).  Same rule inside a block, whose rows take n times their share.
    if ((m->cfg.update_policy == 5 || m->cfg.update_policy == 6) && workers > 1 && !allow_unsafe) {
        const double chain = (double)workers * m->row_share_max * (double)std::max(m->part_n, 1);
        // (and only where that chain is long: a contended hand-over of a row lock takes ~100 us — rank^-0.5 over 1e6 rows: 38 000 pairs of the busiest row in 5 s —, so a
        //  launch whose busiest row has fewer than 50 000 pairs is merely slow for seconds: the edge-case tests on 1- and 3-row vocabularies)
        if (chain > 2.0 && m->row_share_max * (double)std::max(m->part_n, 1) * launch_pairs > 5e4)
            DGE_FAIL(DGE_ERR_ARG, "update_policy %d (commit locks on every row) on this vocabulary: its busiest row holds %.2g of the tokens, %lld workers x that share = %.1f pairs "
                     "queue behind ONE row lock at any time and the launch would be that row's chain (bound 2): use update_policy 0 (auto) or 7 (the head by atomics)",
                     m->cfg.update_policy, m->row_share_max, (long long)workers, chain);
    }
    if (pol == 7) {
        const double fail_all = (double)((int64_t)m->n_cus * 3 * 16) * 5.0 * m->neg_collision;
        // a flat vocabulary with a few busy rows: only those; a skewed one: the whole head
        p.hot_rows = (int32_t)std::min<int64_t>((m->cfg.update_policy == 0 && fail_all < 0.25) ? m->hot_rows_serial : std::max(m->hot_rows_auto, m->hot_rows_serial), m->V);
        if (g_dge_tuning[DGE_TUNE_HOT_ROWS] >= 0) p.hot_rows = (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_HOT_ROWS], m->V);     // ablation knob
        // off unless asked for: measured on cfg3_zipf it buys 2-4 % and shifts the trained scores (profiles/r03_zipf_ablation.txt)
        p.acc_rows = g_dge_tuning[DGE_TUNE_ACC_ROWS] > 0 ? (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_ACC_ROWS], 64) : 0;       // (the kernel caps it at what its LDS holds)
        if (g_dge_tuning[DGE_TUNE_ACC_DRAIN] > 0) p.acc_drain = (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_ACC_DRAIN], 1 << 20);
    }
    if (pol == 100 || (workers == 1 && pol != 5 && pol != 6 && pol != 7 && pol != 2 && pol != 1)) pol = 0;   // in-order: plain accesses
    if (pol == 3 || pol == 8) pol = 0;          // (policy 8 with one worker: the in-order schedule)
    if (part && L > 64) DGE_FAIL(DGE_ERR_ARG, "the block schedule keeps a walk's tokens in registers: walks of up to 64 tokens, not %d", L);
    if (part) {
        // One block of the multi-GPU schedule: the live rows are V/part_n per table, so lock attempts collide part_n times
        // as often as on the whole table (measured on cfg3 with bench.py --sim-ranks, profiles/r01_block_schedule_sim.txt).
        p.hot_rows = 0;
        // A skewed vocabulary keeps the head / tail split inside a block (round 4; until then such a block ran with float atomics on every row): the
        // block's own head — block_head, the one-GPU rule with the block's collision rate — by atomics through the workgroup's atomics wave, the tail
        // under the commit locks.  Two resident workgroups of 12 workers a compute unit (k_sgns_train_locked<HOTMIX, PART>).
        const int64_t w_mixed = (int64_t)m->n_cus * 2 * 12;
        const int64_t head_b = (m->cfg.update_policy == 0 || m->cfg.update_policy == 7) && workers > 1 && m->V / m->part_n >= 32768 ? block_head(m, m->part_n, w_mixed) : 0;
        const int64_t head_knob = g_dge_tuning[DGE_TUNE_HOT_ROWS];
        // with the hierarchical softmax (round 4): the in-order schedule or memory-side atomics, as on one GPU — inner-node rows are split by node % n like the
        // vocabulary rows, every block visits every centre for the path nodes of its target partition (k_sgns_train<.., HS, PART>)
        if (hs) pol = pol == 0 ? 20 : 22;
        else if (pol == 0) pol = 20;
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
            } else if (m->V / m->part_n >= 32768 && head_b <= m->V / 4) {
                // a skewed vocabulary (cfg5, cfg3 with Zipf destinations): the block's head by atomics, its tail under the locks
                pol = 27; p.hot_rows = (int32_t)head_b; p.syn0_free = 0;
                if (m->cfg.workers == 0) { workers = std::min<int64_t>(workers, (int64_t)m->n_cus * 2 * 16); p.n_workers = workers; }
            } else pol = 22;
        }
        else if (pol == 2) pol = 22;
        else if (pol == 5) pol = 25;
        else if (pol == 7) { pol = 27; p.hot_rows = (int32_t)head_b; p.syn0_free = 1; }      // locks on syn1neg's tail only (the pair's syn0 row by atomics)
        else DGE_FAIL(DGE_ERR_ARG, "the block schedule runs under update_policy 0 (auto), 2, 3, 5 or 7, not %d", m->cfg.update_policy);
        if (pol == 27) {
            if (head_knob >= 0) p.hot_rows = (int32_t)std::min<int64_t>(head_knob, m->V);                                       // ablation knobs
            if (g_dge_tuning[DGE_TUNE_BLOCK_SYN0_FREE] >= 0) p.syn0_free = g_dge_tuning[DGE_TUNE_BLOCK_SYN0_FREE] > 0 ? 1 : 0;
            // the partition's hottest rows of BOTH tables add up in the atomics wave's LDS accumulators (lk_atomics_wave, LkAcc): in a block one row's atomics are the
            // longest chain of the launch
            // longest chain of the launch (cfg3_zipf at 8 ranks: 22.6 -> 15.2 ms a block).  Updates a flush: what keeps a row's parked updates — at most one flush short
            // in every workgroup — under 2 048, half of what the owner-computes schedule lets a row take from one stale value (dge_sorted_batch_items); measured on the
            // cfg3-sized Zipf graph at 8 ranks with 512 workgroups: 4 a flush AUC 0.826 / loss 1.280 (banks off 0.817 / 1.285), 8 a flush 0.825 / 1.296, 16 a flush
            // diverges (profiles/r05_blocks_acc_quality_zipf.txt).  (the kernel caps the rows at what its LDS holds: 16 a bank, 8 from 129 floats a row on)
            // (the flush period is set below, once the launch's workgroups are known)
            // ... where that chain is long against the block: the partition's busiest row takes part_n x max(its share of the contexts, K x its share of the negative
            // draws) of the block's pairs, ~78 ns each, against ~3 TB/s of row traffic for a pair.  Where it is short the banks buy nothing and cost a little: the
            // community graph with a Zipf fifth (chain a quarter of the block) loses 0.0034 AUC / 2.3 % of the loss with them and gains 1 % (tests/test_gpu_blocks_scale.py).
            const double top = (double)m->h_counts[0];
            const double chain_share = (double)m->part_n * std::max(top / (double)std::max<int64_t>(m->total_words, 1), (double)m->cfg.negative * pow(top, 0.75) / m->neg_norm);
            const double pair_s = 8.0 * (double)m->stride * (double)(m->cfg.negative + 2) / 3e12;
            // (scripts/block_head_rule.py: that ratio is 1.08 on cfg3_zipf — banks: +50 % and a better AUC —, 0.55 on cfg5 — +10 % —, 0.32 on the community graph: on from 0.4)
            const int32_t auto_rows = chain_share * 78e-9 > 0.4 * pair_s ? 16 : 0;
            p.acc_rows = g_dge_tuning[DGE_TUNE_ACC_ROWS] >= 0 ? (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_ACC_ROWS], 64) : auto_rows;
        }
    }
    size_t shmem = 0;
    if (hs) {
        pol = pol == 0 ? 10 : (pol == 20 ? 30 : (pol == 22 ? 32 : 12));      // dge_model_create admitted policies 0/2/3 only; 30 / 32: one block of the multi-GPU schedule
        if ((pol == 12 || pol == 32) && workers > 1) {         // (one worker: the sequential schedule — no LDS accumulators, no cold class, every node by atomics it waits for)
            // LDS accumulators for the inner nodes nearest the root: 30 KB a block (3 blocks a CU stay resident beside the atomics wave's boxes)
            const int64_t row_b = (int64_t)m->stride * 4 + 4;
            p.hs_n_hot = (int32_t)std::min<int64_t>(std::max<int64_t>(m->V - 1, 0), 30720 / row_b);
            p.hs_hot0 = (int32_t)(std::max<int64_t>(m->V - 1, 0) - p.hs_n_hot);
            p.hs_drain = 64;
            if (g_dge_tuning[DGE_TUNE_HS_DRAIN] >= 1) p.hs_drain = (int32_t)g_dge_tuning[DGE_TUNE_HS_DRAIN];      // ablation knob
            // the cold end of the tree: plain read-modify-write instead of atomics (see k_sgns_train)
            p.hs_cold = (int32_t)std::min<int64_t>(m->hs_cold_auto, p.hs_hot0);
            if (g_dge_tuning[DGE_TUNE_HS_COLD] >= 0) p.hs_cold = (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_HS_COLD], p.hs_hot0);
            shmem = (size_t)p.hs_n_hot * (size_t)row_b;
        }
    }
    unsigned threads = workers == 1 ? 64u : 256u;
    unsigned blocks = (unsigned)((workers * 16 + threads - 1) / threads);
    const bool big_tables = (uint64_t)m->V * (uint64_t)m->stride * 4ull >= 0xFFFFFFFFull || g_dge_tuning[DGE_TUNE_FORCE_SEGMENTS] > 0;
    // (from 65 536 vocabulary rows on, like the atomics wave below: on the reference's own 6 408-row tract vocabulary the walks in flight are capped by the
    //  vocabulary — 801 waves — and the pair-by-pair kernel's 3 204 groups are faster: 1.64e8 against 1.22e8 edges/s; DGE_TUNE_HS_CENTRE = 1 forces it)
    if (pol == 12 && workers > 1 && m->stride <= 256 && L <= 64 && !big_tables &&
        (g_dge_tuning[DGE_TUNE_HS_CENTRE] > 0 || (g_dge_tuning[DGE_TUNE_HS_CENTRE] < 0 && m->V >= 65536))) {
        // Hierarchical softmax, a wave per centre (k_sgns_train_hsw, round 4): the centre's path nodes stay in the registers of a wave's four groups for all
        // its contexts and their gathered updates leave once per centre.  `workers` = walks in flight = waves that train: two resident workgroups of three
        // such waves (and one atomics wave) a compute unit; never more than an eighth of the vocabulary (a wave works on four context rows at a time).
        pol = 13;
        int nw = 3;
        // ... and where the negative-sampling kernels would run under commit locks (a flat vocabulary: auto_policy 5), the pair's negatives and the centre's
        // gathered syn1neg update go under the rows' locks instead of out as atomics (k_sgns_train_hsw<.., NLOCK>) — in ONE workgroup of seven training waves a
        // compute unit, which share their LDS accumulators (DGE_TUNE_HS_CENTRE: 1 keeps atomics, 2 = locks in workgroups of three waves, 3 = of seven)
        const int64_t centre_knob = g_dge_tuning[DGE_TUNE_HS_CENTRE];
        // ... and on a SKEWED vocabulary whose head the mixed policy 7 would take out of the lock protocol (round 5): the same kernel with that head by atomics, the tail's
        // negatives under locks (p.hot_rows; DGE_TUNE_HOT_ROWS sets it by hand)
        const bool mixed_ok = m->V >= 131072 && (int64_t)(48.0 / std::max(m->row_share_max, 1e-12)) >= 4096 && std::max(m->hot_rows_auto, m->hot_rows_serial) <= m->V / 4;
        if (centre_knob == 2 || centre_knob == 3 || (centre_knob < 0 && m->cfg.update_policy == 0 && (syn1neg_locks_work(m) || mixed_ok))) {
            const bool mixed = !syn1neg_locks_work(m) && mixed_ok && centre_knob < 0;
            pol = (centre_knob == 2 || m->stride > 128 || mixed) ? 14 : 15;      // (rows of more than 128 floats: three-wave workgroups only — seven waves' message boxes do not fit the LDS; a head by atomics: three waves to an atomics wave, not seven)
            if (pol == 15) nw = 7;
            if (!syn1neg_locks_work(m) && mixed_ok) p.hot_rows = (int32_t)std::min<int64_t>(std::max(m->hot_rows_auto, m->hot_rows_serial), m->V);
            if (g_dge_tuning[DGE_TUNE_HOT_ROWS] >= 0) p.hot_rows = (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_HOT_ROWS], m->V);
        }
        if (m->cfg.workers == 0 && !(g_dge_tuning[DGE_TUNE_WORKERS] > 0))
            workers = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)m->n_cus * (nw == 3 ? (m->stride <= 128 ? 6 : 3) : 7), std::max<int64_t>(1, m->V / 8)), std::max<int64_t>(16, (int64_t)(12.0 / std::max(m->row_share_max, 1e-12)))));      // (wide rows: one three-wave workgroup a compute unit)
        workers = std::max<int64_t>(1, std::min<int64_t>(workers, n_rows));
        p.n_workers = workers;
        blocks = (unsigned)((workers + nw - 1) / nw);
        threads = (unsigned)(nw + 1) * 64u;
        // an LDS accumulator now takes one addition per CENTRE (16 pairs' worth on cfg3): drained every 4 additions instead of every 64 — every 8 where seven
        // waves share it (the same number of additions parked device-wide: workgroups x drain)
        p.hs_drain = g_dge_tuning[DGE_TUNE_HS_DRAIN] >= 1 ? (int32_t)g_dge_tuning[DGE_TUNE_HS_DRAIN] : (nw == 3 ? 4 : 8);
        // Measured on cfg3 and the cfg3-sized community graph (profiles/r04_hs_waves7.txt; three waves, drain 4: 3.3e8 edges/s, AUC 0.9534): seven waves with 15 KB of
        // accumulators (the 29 nodes next to the root) and a drain every 8 additions 3.6e8 at AUC 0.9533; with 30 KB 3.64e8 / 0.9526, with 60 KB 3.7e8 / 0.9519 — every
        // accumulator is a row whose readers lag workgroups x drain / 2 updates behind, so fewer of them and shared by more waves is the better trade (three waves at
        // drain 8: 3.63e8 / 0.9518).
        // The busiest inner nodes in copies instead of LDS accumulators (k_sgns_train_hsw, HS_REP): the default; DGE_TUNE_HS_HOT_KB > 0 brings the accumulators back
        // (that many KB of them; the drain period then matters again) for comparison.
        const int64_t hot_kb_knob = g_dge_tuning[DGE_TUNE_HS_HOT_KB];
        if (hot_kb_knob <= 0 && m->hs_rep_auto > 0 && m->V > 1) {
            const int64_t n_rep = std::min<int64_t>(std::min<int64_t>(m->hs_rep_auto, HS_REP_NODES), m->V - 1);
            p.hs_rep_n = (int32_t)n_rep; p.hs_rep0 = (int32_t)(m->V - 1 - n_rep);      // (the copies: rows V .. of syn1, zero between launches: k_hs_rep_fold)
            // copies per node: ceil(share of the paths x F), F = HS_REP (the root: HS_REP = 16 copies, a node on half the paths 8, ...; DGE_TUNE_HS_COPIES = F for comparison: 4 = the root four copies, ...)
            { const int64_t f_knob = g_dge_tuning[DGE_TUNE_HS_COPIES];
              const int F = (int)std::min<int64_t>(HS_REP, f_knob >= 1 ? f_knob : (int64_t)HS_REP);
              for (int k = 1; k < HS_REP; k++) p.hs_rep_thr[k] = k < F ? std::max(m->hs_rep_thr32[std::min(31, k * 32 / F)], p.hs_rep0) : 0x7fffffff;      // more than k copies: share x F > k
              p.hs_rep_thr[0] = 0; }
            p.hs_n_hot = 0; p.hs_hot0 = (int32_t)std::max<int64_t>(m->V - 1, 0); shmem = 0;
            { const int64_t cold_knob = g_dge_tuning[DGE_TUNE_HS_COLD]; p.hs_cold = (int32_t)std::min<int64_t>(cold_knob >= 0 ? cold_knob : (int64_t)m->hs_cold_auto, p.hs_rep0); }
        } else {                   // LDS accumulators (DGE_TUNE_HS_HOT_KB > 0; or a tree without a busy node): seven waves, one workgroup a compute unit — up to 100 KB; three waves, two workgroups — up to 30 KB each
            const int64_t row_b = (int64_t)m->stride * 4 + 4;
            const int64_t hot_kb = g_dge_tuning[DGE_TUNE_HS_HOT_KB];
            p.hs_n_hot = (int32_t)std::min<int64_t>(std::max<int64_t>(m->V - 1, 0), (hot_kb > 0 ? std::min<int64_t>(hot_kb, nw == 7 ? 100 : 30) * 1024 : (nw == 7 ? 15360 : 30720)) / row_b);
            p.hs_hot0 = (int32_t)(std::max<int64_t>(m->V - 1, 0) - p.hs_n_hot);
            { const int64_t cold_knob = g_dge_tuning[DGE_TUNE_HS_COLD]; p.hs_cold = (int32_t)std::min<int64_t>(cold_knob >= 0 ? cold_knob : (int64_t)m->hs_cold_auto, p.hs_hot0); }
            shmem = (size_t)p.hs_n_hot * (size_t)row_b;
        }
    }
    // (not on small vocabularies, where the worker count is capped at half the rows and every pair is a latency chain: the reference's own
    //  801 x 8 tract graph with hierarchical softmax runs 407 ms per 6.5e7 pairs on its 3 204 workers, 552 ms on 2 400 workers and a wave)
    if ((pol == 12 || pol == 32) && workers > 1 && (g_dge_tuning[DGE_TUNE_HS_WAVE] > 0 || (g_dge_tuning[DGE_TUNE_HS_WAVE] < 0 && m->V >= 65536))) {
        // hierarchical softmax under atomics: every workgroup's fourth wave issues the atomics of its 12 workers (k_sgns_train, lk_atomics_wave)
        p.hs_wave = 1;
        // (three workgroups a compute unit stay resident next to their LDS accumulators and message boxes: DGE_HS_WAVES)
        if (m->cfg.workers == 0 && !(g_dge_tuning[DGE_TUNE_WORKERS] > 0)) { workers = std::max<int64_t>(std::min<int64_t>(workers / 16 * 12, (int64_t)m->n_cus * DGE_HS_WAVES * 12), 2); p.n_workers = workers; }
        blocks = (unsigned)((workers + 11) / 12);
    }
    if ((pol == 7 || pol == 27) && workers > 1) {
        // the mixed kernels keep every workgroup's fourth wave for the head rows' atomics (k_sgns_train_locked): 12 workers a workgroup
        if (m->cfg.workers == 0 && !(g_dge_tuning[DGE_TUNE_WORKERS] > 0)) { workers = std::max<int64_t>(workers / 16 * 12, 2); p.n_workers = workers; }
        blocks = (unsigned)((workers + 11) / 12);
        if (pol == 27) p.acc_drain = g_dge_tuning[DGE_TUNE_ACC_DRAIN] > 0 ? (int32_t)std::min<int64_t>(g_dge_tuning[DGE_TUNE_ACC_DRAIN], 1 << 20)
                                                                        : (int32_t)std::max<int64_t>(1, std::min<int64_t>(8, 2048 / std::max(blocks, 1u)));
    }

    // per-segment descriptors (TableView) for tables of 4 GiB and more; DGE_TUNE_FORCE_SEGMENTS selects that code path on small
    // tables too so that the parity tests can cover it
    const bool big = (uint64_t)m->V * (uint64_t)m->stride * 4ull >= 0xFFFFFFFFull || g_dge_tuning[DGE_TUNE_FORCE_SEGMENTS] > 0;
    // Walks handed out by a launch-wide counter wherever several workers run, the order of the walks is free (not the in-order schedule) and
    // a worker trains enough walks for the hand-out to even anything out.  Measured on one model in one process (scripts/ab_inproc.py): cfg3
    // 415 against 426 ms per launch, cfg3_zipf 666 against 678, hierarchical softmax 3.24 against 3.28 s, atomics 914 against 925; cfg2 under
    // atomics (6 walks per worker) 5.6 against 5.4 — there the workers keep their fixed walks.
    if (workers > 1 && n_rows >= 32 * workers && pol != 0 && !(g_dge_tuning[DGE_TUNE_STATIC_WALKS] > 0)) {
        p.next_walk = m->d_counters + 2;
        DGE_HIP(hipMemsetAsync(p.next_walk, 0, sizeof(unsigned long long), st));
    }
    // A launch whose length is one busy row's chain of pairs — forced policy 5 or 6 on a vocabulary with such a row — is bound by a pair's latency, not by
    // requests, and the 11 dependent LDS reads of the table's run form are slower than a table look-up that hits the caches (15.4-15.9 s against 12.8 s
    // on the community graph of scripts/quality_scale.py): there the table stays.
    if (m->hot_rows_serial > 0 && workers > 1) p.n_runs = 0;
    // The lock kernels' watchdog: a worker that is still waiting for a row lock when the launch has run 100x longer than its bytes take at the roofline (+ 5 s) gives up,
    // counts itself in counters[3] and leaves; dge_model_stats then reports DGE_ERR_STATE.  Checked on the waiting paths only (100 MHz s_memrealtime ticks).
    {
        const double bytes = launch_pairs * 8.0 * (double)m->stride * (double)(m->cfg.negative + 2);
        double budget_s = 5.0 + 100.0 * bytes / 8e12;
        if (g_dge_tuning[DGE_TUNE_WATCHDOG_MS] > 0) budget_s = (double)g_dge_tuning[DGE_TUNE_WATCHDOG_MS] * 1e-3;
        // only where the commit locks on every row were FORCED (update_policy 5 / 6): what auto picks them for cannot make them wait, and the watchdog is its own kernel
        // instantiation so that the headline launch does not pay for it (sgns_kernels.h: WDOG)
        const bool forced_locks = m->cfg.update_policy == 5 || m->cfg.update_policy == 6;
        p.wd_ticks = (!forced_locks || g_dge_tuning[DGE_TUNE_WATCHDOG_MS] == 0) ? 0ull : (uint64_t)(budget_s * 1e8);
    }
    // k_sgns_train_hsw's copies of the busiest inner nodes must be zero when the launch starts.  k_hs_rep_fold leaves them so behind every launch; an abandoned launch
    // (an error between the two) would not — so they are cleared here as well (hs_rep_n x 15 rows, < 1 MB)
    if (p.hs_rep_n > 0) DGE_HIP(hipMemsetAsync(m->d_syn1 + (size_t)m->V * m->stride, 0, (size_t)(HS_REP - 1) * (size_t)p.hs_rep_n * (size_t)m->stride * sizeof(float), st));
    // rows of 17 .. 32 floats under the atomics policy (the reference's own layerSize 20): half a wave a worker, a row = one request each way (k_sgns_train_small)
    const bool small_rows = pol == 2 && (workers > 1 || g_dge_tuning[DGE_TUNE_SMALL_ROWS] > 0) && m->cfg.dim > 16 && m->cfg.dim <= 32 && m->stride == 64 && L <= 64 && !big && g_dge_tuning[DGE_TUNE_SMALL_ROWS] != 0;
    if (small_rows) { threads = 256u; blocks = (unsigned)((workers * 32 + 255) / 256); }
    EventPair ev;
    if ((rc = timing_begin(m, ev, 0))) return rc;
    if (small_rows) hipLaunchKernelGGL((k_sgns_train_small<32>), dim3(blocks), dim3(threads), 0, st, p);
    else
    switch (m->stride / 64) {
        case 1: dge_launch_train_dch1(p, pol, big, blocks, threads, shmem, st); break;
        case 2: dge_launch_train_dch2(p, pol, big, blocks, threads, shmem, st); break;
        case 3: dge_launch_train_dch3(p, pol, big, blocks, threads, shmem, st); break;
        case 4: dge_launch_train_dch4(p, pol, big, blocks, threads, shmem, st); break;
        case 6: dge_launch_train_dch6(p, pol, big, blocks, threads, shmem, st); break;
        default: dge_launch_train_dch8(p, pol, big, blocks, threads, shmem, st); break;
    }
    if (p.hs_rep_n > 0)      // the copies of the busiest inner nodes go back into their rows (and are zero again for the next launch's memset to find nothing)
        hipLaunchKernelGGL(k_hs_rep_fold, dim3((unsigned)p.hs_rep_n), dim3(128), 0, st, m->d_syn1 + (size_t)p.hs_rep0 * m->stride, m->d_syn1 + (size_t)m->V * m->stride, p.hs_rep_n, m->stride);
    if ((rc = timing_end(m, ev, DGE_OK))) return rc;
    m->launches++;
    {   // which trainer kernel ran (dge_model_kernel): the bench line names it from here, not from the policy number
        const int base = pol >= 30 ? pol - 30 : (pol >= 20 ? pol - 20 : (pol >= 10 ? pol - 10 : pol));
        const bool blk = pol >= 20;
        std::string k;
        if (pol == 13) k = "k_sgns_train_hsw<atomics, 3 waves> (hierarchical softmax, a wave per centre)";
        else if (pol == 14) k = std::string("k_sgns_train_hsw<negatives under commit locks") + (p.hot_rows > 0 ? ", head rows by atomics" : "") + ", 3 waves> (hierarchical softmax, a wave per centre)";
        else if (pol == 15) k = std::string("k_sgns_train_hsw<negatives under commit locks") + (p.hot_rows > 0 ? ", head rows by atomics" : "") + ", 7 waves> (hierarchical softmax, a wave per centre)";
        else if (base == 5 || base == 6 || base == 7) k = std::string("k_sgns_train_locked<") + (base == 6 ? "strict" : "relaxed") + (base == 7 ? ", head rows by atomics" : "") + (blk ? ", one block" : "") + ">";
        else if (small_rows) k = "k_sgns_train_small<atomics, 32 lanes a worker>";
        else k = std::string("k_sgns_train<") + (base == 2 ? "atomics" : (base == 1 ? "row rmw" : "in-order")) + (hs ? ", hierarchical softmax pair by pair" : "") + (blk ? ", one block" : "") + ">";
        m->last_kernel = k;
    }
    m->last_policy = pol >= 30 ? pol - 30 : (pol >= 20 ? pol - 20 : (pol == 13 ? 2 : (pol == 14 || pol == 15 ? (p.hot_rows > 0 ? 7 : 5) : (pol >= 10 ? pol - 10 : pol)))); m->last_workers = workers; m->last_hot_rows = p.hot_rows;
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
    return train_rows(m, w->d + row0 * w->L, n_rows, w->L, walk_index_base, epoch, words_before, words_scale, total_walks, w->gen);
}

extern "C" int dge_model_walk_and_train(dge_model* m, const dge_graph* g, dge_walks* w, int64_t row0, int64_t n_rows, int64_t walk_seed,
                                        int64_t walk_index_base, int32_t epoch, int64_t words_before, double words_scale, int64_t total_walks) {
    if (!m || !g || !w || row0 < 0 || n_rows < 0 || row0 + n_rows > w->n) DGE_FAIL(DGE_ERR_ARG, "dge_model_walk_and_train: bad argument");
    if (!g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_model_walk_and_train: alias tables not built");
    if (w->device != m->device || g->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_walk_and_train: handles live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    w->gen = dge_next_generation();
    EventPair ev;
    int rc = timing_begin(m, ev, 1);
    if (rc) return rc;
    rc = timing_end(m, ev, dge_launch_walks_strided(g, m->stream, w->d + row0 * w->L, n_rows, w->L, walk_seed, walk_index_base, nullptr));
    if (rc) return rc;
    if (total_walks <= 0) total_walks = w->n;
    return train_rows(m, w->d + row0 * w->L, n_rows, w->L, walk_index_base, epoch, words_before, words_scale, total_walks, w->gen);
}

extern "C" int dge_model_tune_placement(dge_model* m, const dge_walks* w, int64_t row0, int64_t n_rows, int32_t candidates, double* ms_before, double* ms_after,
                                        int32_t* arrays_moved);
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
    // The placement search first — where it can pay (results are unaffected: dge_model_tune_placement).  A pass of the search is 2 + 4 (candidates - 1)
    // probe launches of n_probe walks (a model that started well needs one pass), and what it can win is ~10 % of the training launches' time
    // (profiles/r03_placement.txt): it runs when a tenth of the projected training — epochs x walks — is more than one pass.  The reference's own
    // iterations(1) fit of cfg3 (10 M walks, 3.9 s) is below that: 1.3 - 8 s of probes would be a net loss there (round-3 verdict); 4 epochs are above.
    if (!rc && cfg->epochs > 0 && cfg->workers != 1 && m->V >= 262144 && w->n >= 262144) {
        const int candidates = 4;
        const int64_t n_probe = std::min<int64_t>(w->n / 8, 262144);
        if (0.10 * (double)cfg->epochs * (double)w->n > (double)(2 + 4 * (candidates - 1)) * (double)n_probe) {
            double before = 0, after = 0; int32_t moved = 0;
            rc = dge_model_tune_placement(m, w, 0, n_probe, candidates, &before, &after, &moved);
        }
    }
    for (int ep = 0; ep < cfg->epochs && !rc; ep++) rc = dge_model_train(m, w, 0, w->n, 0, ep, 0, 1.0, w->n);
    if (!rc) { hipError_t e = hipStreamSynchronize(m->stream); if (e != hipSuccess) { dge_set_error("training failed: %s", hipGetErrorName(e)); rc = DGE_ERR_DEVICE; } }
    if (!rc) { dge_train_stats st; rc = dge_model_stats(m, &st); }      // (a launch ended by the lock kernels' watchdog is an error of the fit, not a model)
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
    dge_tmp<int32_t> flat;                                   // word2vec's one-row-per-slot form, expanded from the rank blocks
    int rc = flat.alloc((size_t)m->T);
    if (rc) return rc;
    DGE_HIP(hipStreamSynchronize(m->stream));
    hipLaunchKernelGGL(k_table_unpack, dim3(grid_for(m->T, 256)), dim3(256), 0, m->stream, m->d_ctab, m->T, flat.p);
    DGE_HIP(hipStreamSynchronize(m->stream));
    DGE_HIP(hipMemcpy(m->h_table.data(), flat.p, (size_t)m->T * sizeof(int32_t), hipMemcpyDeviceToHost));
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
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { if (e.kind == 0) m->kernel_ms += ms; else m->walk_ms += ms; }
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
    unsigned long long c[4] = {0, 0, 0, 0};
    DGE_HIP(hipMemcpy(c, m->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    out->pairs = (int64_t)c[0]; out->words = (int64_t)c[1];
    out->kernel_ms = m->kernel_ms; out->walk_kernel_ms = m->walk_ms; out->launches = m->launches;
    if (c[3]) {     // a lock kernel's watchdog ended a launch (train_rows: wd_ticks): reported once, here — the first call that looks at the device after the launch
        DGE_HIP(hipMemsetAsync(m->d_counters + 3, 0, sizeof(unsigned long long), m->stream));
        DGE_HIP(hipStreamSynchronize(m->stream));
        DGE_FAIL(DGE_ERR_STATE, "a training launch was ended by its watchdog: %llu workers waited for row locks beyond the launch's time budget and left their walks untrained "
                 "(update_policy %d on this vocabulary: use 0 (auto) or 7)", c[3], m->cfg.update_policy);
    }
    return DGE_OK;
}

// ---- dge_model_row_rates: how fast THIS model's memory answers the three things the lock kernel does to it — rows read at random, rows read and
// written back (write-through stores, as a commit does), exchanges on random lock words.  Two models of one process can differ by 15 % in
// training speed while the device's copy rate does not move (profiles/r02_box_drift.txt): the difference follows the allocation, and this
// probe shows which access it is without training anything.  16 lanes per row, 8 rows in flight per group; tables below 4 GiB.
template <int MODE>
__global__ void __launch_bounds__(256) k_probe_rows(float* t0, float* t1, int* locks, const uint4* ctab, int64_t T, int64_t V, int32_t stride, int64_t reads_per_group,
                                                    float* sink) {
    const int lane = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = dge_mix64(0x9E3779B97F4A7C15ull * (uint64_t)(group + 1));
    const TableView v0 = make_view(t0, V, stride), v1 = make_view(t1, V, stride);
    float acc = 0.f;
    if (MODE == 3) {
        for (int64_t i = 0; i < reads_per_group; i += 4) {
            int32_t t[4];
#pragma unroll
            for (int z = 0; z < 4; z++) { s = s * DGE_W2V_MULT + 11; t[z] = neg_table_row(ctab, ((s >> 16) + (uint64_t)lane * 0x9E3779B1ull) % (uint64_t)T); }
            acc += (float)(t[0] ^ t[1] ^ t[2] ^ t[3]);
        }
    } else if (MODE == 2) {
        for (int64_t i = 0; i < reads_per_group; i++) {
            s = s * DGE_W2V_MULT + 11;
            const int64_t w = (int64_t)(((s >> 16) + (uint64_t)lane * 0x9E3779B1ull) % (uint64_t)(2 * V));
            acc += (float)__hip_atomic_exchange(&locks[w], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        for (int64_t i = 0; i < reads_per_group; i += 8) {
            v4u v[8]; uint32_t off[8]; uint32_t mix = 0;
#pragma unroll
            for (int z = 0; z < 8; z++) {
                s = s * DGE_W2V_MULT + 11;
                off[z] = (uint32_t)((s >> 16) % (uint64_t)V) * v0.row_bytes + (uint32_t)lane * 16u;
                v[z] = __builtin_amdgcn_raw_buffer_load_b128((z & 1) ? v1.rsrc : v0.rsrc, (int)off[z], 0, 0);
                for (uint32_t c = 256; c < v0.row_bytes; c += 256) {
                    const v4u u = __builtin_amdgcn_raw_buffer_load_b128((z & 1) ? v1.rsrc : v0.rsrc, (int)(off[z] + c), 0, 0);
                    mix ^= u.x;
                }
            }
#pragma unroll
            for (int z = 0; z < 8; z++) {
                acc += __uint_as_float(v[z].x ^ mix);
                if (MODE == 1) {           // the same bytes back, write-through (aux 16 = sc1), every 256-byte piece of the row
                    for (uint32_t c = 0; c < v0.row_bytes; c += 256) {
                        const v4u u = c ? __builtin_amdgcn_raw_buffer_load_b128((z & 1) ? v1.rsrc : v0.rsrc, (int)(off[z] + c), 0, 0) : v[z];
                        __builtin_amdgcn_raw_buffer_store_b128(u, (z & 1) ? v1.rsrc : v0.rsrc, (int)(off[z] + c), 0, 16);
                    }
                }
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;                 // keeps the loads alive
}

extern "C" int dge_model_row_rates(dge_model* m, double* read_gb_per_s, double* rewrite_gb_per_s, double* lock_exchanges_per_s, double* table_lookups_per_s) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_row_rates: null model");
    if ((uint64_t)m->V * (uint64_t)m->stride * 4ull >= 0xFFFFFFFFull) DGE_FAIL(DGE_ERR_ARG, "dge_model_row_rates: tables of 4 GiB and more are not probed");
    DGE_HIP(hipSetDevice(m->device));
    int rc = drain_events(m);
    if (rc) return rc;
    DGE_HIP(hipStreamSynchronize(m->stream));
    const int64_t groups = 256 * 16 * 16, reads = 256;     // 65 536 groups x 256 rows
    dge_tmp<float> sink;
    if ((rc = sink.alloc(4))) return rc;
    hipEvent_t e0, e1;
    DGE_HIP(hipEventCreate(&e0)); DGE_HIP(hipEventCreate(&e1));
    double best[4] = {0, 0, 0, 0};
    for (int mode = 0; mode < 4; mode++) {
        for (int r = 0; r < 3; r++) {
            DGE_HIP(hipEventRecord(e0, m->stream));
            const dim3 grid((unsigned)(groups * 16 / 256));
#define PROBE(M) hipLaunchKernelGGL(k_probe_rows<M>, grid, dim3(256), 0, m->stream, m->d_syn0, m->d_syn1neg, m->d_locks, m->d_ctab, m->T, m->V, m->stride, reads, sink.p)
            if (mode == 0) PROBE(0); else if (mode == 1) PROBE(1); else if (mode == 2) PROBE(2); else PROBE(3);
#undef PROBE
            DGE_HIP(hipEventRecord(e1, m->stream));
            DGE_HIP(hipEventSynchronize(e1));
            float ms = 0.f; DGE_HIP(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)groups * reads;
            const double rate = mode >= 2 ? n * 16.0 / (ms * 1e-3) : n * m->stride * 4.0 * (mode == 1 ? 2.0 : 1.0) / (ms * 1e-3) / 1e9;
            if (rate > best[mode]) best[mode] = rate;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DGE_HIP(hipGetLastError());
    if (read_gb_per_s) *read_gb_per_s = best[0];
    if (rewrite_gb_per_s) *rewrite_gb_per_s = best[1];
    if (lock_exchanges_per_s) *lock_exchanges_per_s = best[2];
    if (table_lookups_per_s) *table_lookups_per_s = best[3];
    return DGE_OK;
}

// ---- dge_model_tune_placement.  Which physical memory hipMalloc hands an array decides a training launch's duration by up to 15 %, array by
// array, and no allocation rule (contiguous blocks, aligned ranges, shuffled 2 MiB chunks) nor any cheap probe of the memory predicts it
// (profiles/r02_box_drift.txt, profiles/r03_placement.txt).  So the library searches with the only probe that works, the caller's own launch:
// rows [row0, row0 + n_rows) of `w` are trained once for a baseline; then, one array at a time (negative-sampling table, lock words, syn1neg,
// syn0), a copy in freshly allocated memory takes the array's place, the same rows are trained again, and the faster placement stays.
// Rejected placements are only freed at the end (the allocator would hand the same memory out again).  The tables' contents and the
// model's counters are saved first and restored last: training results are exactly those of an untuned model.
static int tune_time_launch(dge_model* m, const dge_walks* w, int64_t row0, int64_t n_rows, double* ms) {
    hipEvent_t a = nullptr, b = nullptr;
    DGE_HIP(hipEventCreate(&a));
    if (hipEventCreate(&b) != hipSuccess || hipEventRecord(a, m->stream) != hipSuccess) {
        (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b);
        DGE_FAIL(DGE_ERR_DEVICE, "dge_model_tune_placement: cannot time the probe launch");
    }
    int rc = train_rows(m, w->d + row0 * w->L, n_rows, w->L, 0, 0, 0, 1.0, std::max<int64_t>(w->n, 1), w->gen);
    if (rc == DGE_OK) {
        if (hipEventRecord(b, m->stream) != hipSuccess || hipEventSynchronize(b) != hipSuccess) { dge_set_error("dge_model_tune_placement: the probe launch failed"); rc = DGE_ERR_DEVICE; }
        float t = 0.f;
        if (rc == DGE_OK && hipEventElapsedTime(&t, a, b) == hipSuccess) *ms = t;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return rc;
}

extern "C" int dge_model_tune_placement(dge_model* m, const dge_walks* w, int64_t row0, int64_t n_rows, int32_t candidates, double* ms_before, double* ms_after,
                                        int32_t* arrays_moved) {
    if (!m || !w || row0 < 0 || n_rows <= 0 || row0 + n_rows > w->n || candidates < 1) DGE_FAIL(DGE_ERR_ARG, "dge_model_tune_placement: bad argument");
    if (w->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_tune_placement: corpus and model live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    int rc = drain_events(m);
    if (rc) return rc;
    if (ms_before) *ms_before = 0; if (ms_after) *ms_after = 0; if (arrays_moved) *arrays_moved = 0;
    if (m->V == 0) return DGE_OK;
    hipStream_t st = m->stream;
    const size_t tab_bytes = ((size_t)m->V * (size_t)m->stride + 64) * sizeof(float);
    const size_t lock_bytes = 2 * ((size_t)m->V + 1) * sizeof(int), ctab_bytes = ((size_t)m->ctab_blocks + 1) * sizeof(uint4);
    // what a probe launch changes: the two tables (syn1 too under hierarchical softmax), the counters, the launch statistics
    dge_tmp<char> keep0, keep1, keep2;
    if ((rc = keep0.alloc(tab_bytes)) || (rc = keep1.alloc(tab_bytes))) return rc;
    if (m->d_syn1 && (rc = keep2.alloc(tab_bytes))) return rc;
    unsigned long long counters[3] = {0, 0, 0};
    DGE_HIP(hipMemcpyAsync(keep0.p, m->d_syn0, tab_bytes, hipMemcpyDeviceToDevice, st));
    DGE_HIP(hipMemcpyAsync(keep1.p, m->d_syn1neg, tab_bytes, hipMemcpyDeviceToDevice, st));
    if (m->d_syn1) DGE_HIP(hipMemcpyAsync(keep2.p, m->d_syn1, tab_bytes, hipMemcpyDeviceToDevice, st));
    DGE_HIP(hipMemcpyAsync(counters, m->d_counters, sizeof(counters), hipMemcpyDeviceToHost, st));
    DGE_HIP(hipStreamSynchronize(st));
    const double k_ms = m->kernel_ms, w_ms = m->walk_ms; const int64_t launches = m->launches;
    const int lp = m->last_policy; const int64_t lw = m->last_workers; const int32_t lh = m->last_hot_rows;

    std::vector<void*> graveyard;
    double best = 0, first = 0;
    int moved = 0;
    // Every probe starts from the tables as they were: a launch's duration depends on them under hierarchical softmax (a path node whose dot product
    // has left the sigmoid's table is skipped), and probes that train the same walks again and again get faster by themselves — the search then
    // "found" 24 improvements and 723 -> 280 ms on a model whose launches did not change (profiles/r03_final_numbers.txt).
    auto reset_tables = [&]() -> int {
        DGE_HIP(hipMemcpyAsync(m->d_syn0, keep0.p, tab_bytes, hipMemcpyDeviceToDevice, st));
        DGE_HIP(hipMemcpyAsync(m->d_syn1neg, keep1.p, tab_bytes, hipMemcpyDeviceToDevice, st));
        if (m->d_syn1) DGE_HIP(hipMemcpyAsync(m->d_syn1, keep2.p, tab_bytes, hipMemcpyDeviceToDevice, st));
        return DGE_OK;
    };
    rc = tune_time_launch(m, w, row0, n_rows, &best);          // warm-up (work buffers, the owner-computes schedule's lazy allocations)
    if (rc == DGE_OK) rc = reset_tables();
    if (rc == DGE_OK) rc = tune_time_launch(m, w, row0, n_rows, &best);
    first = best;
    struct Slot { void** p; size_t bytes; };
    Slot slots[4] = {{(void**)&m->d_ctab, ctab_bytes}, {(void**)&m->d_locks, lock_bytes}, {(void**)&m->d_syn1neg, tab_bytes}, {(void**)&m->d_syn0, tab_bytes}};
    // A pass tries every array in up to candidates - 1 other allocations.  A pass that found nothing ends the search (a model that started
    // well costs one pass); one that did means the model started badly, and arrays it left alone may still be badly placed: passes go on while
    // they find something, at most six (a single pass left one process in three at 119.4 -> 115.7 ms: 447.8 ms per launch where its neighbours ran
    // 414; at most three passes left one in six at 112.4: 434.5; profiles/r03_bench_repeat_search.txt).
    for (int pass = 0; pass < 6 && rc == DGE_OK; pass++) {
        const int moved_before = moved;
        for (int a = 0; a < 4 && rc == DGE_OK; a++)
            for (int c = 1; c < candidates && rc == DGE_OK; c++) {
                void* fresh = nullptr;
                if (hipMalloc(&fresh, slots[a].bytes) != hipSuccess) { (void)hipGetLastError(); break; }      // out of memory: keep what we have
                void* old = *slots[a].p;
                if (hipMemcpyAsync(fresh, old, slots[a].bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { (void)hipFree(fresh); dge_set_error("dge_model_tune_placement: copy failed"); rc = DGE_ERR_DEVICE; break; }
                *slots[a].p = fresh;
                double t = 0;
                rc = reset_tables();
                if (rc == DGE_OK) rc = tune_time_launch(m, w, row0, n_rows, &t);
                if (rc == DGE_OK && t < best * 0.995) { best = t; graveyard.push_back(old); moved++; break; }      // this array is well placed now: next array
                else { *slots[a].p = old; graveyard.push_back(fresh); }
            }
        if (moved == moved_before) break;
    }
    hipError_t e = hipStreamSynchronize(st);
    for (void* g : graveyard) table_free(g);                  // (a table that came from table_alloc is a virtual-memory allocation)
    if (rc == DGE_OK && e != hipSuccess) { dge_set_error("dge_model_tune_placement: %s", hipGetErrorName(e)); rc = DGE_ERR_DEVICE; }
    // put everything back as it was before the probes
    DGE_HIP(hipMemcpyAsync(m->d_syn0, keep0.p, tab_bytes, hipMemcpyDeviceToDevice, st));
    DGE_HIP(hipMemcpyAsync(m->d_syn1neg, keep1.p, tab_bytes, hipMemcpyDeviceToDevice, st));
    if (m->d_syn1) DGE_HIP(hipMemcpyAsync(m->d_syn1, keep2.p, tab_bytes, hipMemcpyDeviceToDevice, st));
    DGE_HIP(hipMemsetAsync(m->d_locks, 0, lock_bytes, st));
    DGE_HIP(hipMemcpyAsync(m->d_counters, counters, sizeof(counters), hipMemcpyHostToDevice, st));
    DGE_HIP(hipStreamSynchronize(st));
    int rc2 = drain_events(m);
    m->kernel_ms = k_ms; m->walk_ms = w_ms; m->launches = launches;
    m->last_policy = lp; m->last_workers = lw; m->last_hot_rows = lh;
    m->seen_gen = 0;                                           // (the next launch derives its rows again)
    if (rc == DGE_OK) rc = rc2;
    if (ms_before) *ms_before = first; if (ms_after) *ms_after = best; if (arrays_moved) *arrays_moved = moved;
    m->search_runs++; m->search_ms_before = first; m->search_ms_after = best; m->search_moved = moved;
    return rc;
}

extern "C" int dge_model_placement_search(const dge_model* m, int32_t* runs, double* ms_before, double* ms_after, int32_t* arrays_moved) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_placement_search: null model");
    if (runs) *runs = m->search_runs;
    if (ms_before) *ms_before = m->search_ms_before;
    if (ms_after) *ms_after = m->search_ms_after;
    if (arrays_moved) *arrays_moved = m->search_moved;
    return DGE_OK;
}

extern "C" int dge_model_table_placement(const dge_model* m, int32_t table, int32_t* candidates, double* best_gb_per_s, double* worst_gb_per_s) {
    if (!m || table < 0 || table > 2) DGE_FAIL(DGE_ERR_ARG, "dge_model_table_placement: table is 0 (syn0), 1 (syn1neg) or 2 (syn1)");
    if (candidates) *candidates = m->placed_seen[table];
    if (best_gb_per_s) *best_gb_per_s = m->placed_best[table];
    if (worst_gb_per_s) *worst_gb_per_s = m->placed_worst[table];
    return DGE_OK;
}

extern "C" int dge_model_table_runs(const dge_model* m, int32_t* n_runs, int32_t* n_exceptions) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_table_runs: null model");
    if (n_runs) *n_runs = m->n_runs;
    if (n_exceptions) *n_exceptions = m->n_exc;
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

extern "C" int dge_model_lock_stats(const dge_model* mc, int64_t* pairs_put_back, int64_t* rounds_short, int64_t* rounds) {
    dge_model* m = const_cast<dge_model*>(mc);
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_lock_stats: null model");
    DGE_HIP(hipSetDevice(m->device));
    DGE_HIP(hipStreamSynchronize(m->stream));
    unsigned long long c[3] = {0, 0, 0};
    DGE_HIP(hipMemcpy(c, m->d_counters + 4, sizeof(c), hipMemcpyDeviceToHost));
    if (pairs_put_back) *pairs_put_back = (int64_t)c[0];
    if (rounds_short) *rounds_short = (int64_t)c[1];
    if (rounds) *rounds = (int64_t)c[2];
    return DGE_OK;
}

extern "C" int dge_model_kernel(const dge_model* m, char* buf, int32_t cap) {
    if (!m || !buf || cap <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_model_kernel: bad argument");
    if (m->last_policy < 0) DGE_FAIL(DGE_ERR_STATE, "dge_model_kernel: nothing has been trained yet");
    snprintf(buf, (size_t)cap, "%s", m->last_kernel.c_str());
    return DGE_OK;
}

extern "C" int dge_model_reset_stats(dge_model* m) {
    if (!m) DGE_FAIL(DGE_ERR_ARG, "dge_model_reset_stats: null model");
    int rc = drain_events(m);
    if (rc) return rc;
    // on the model's own stream (a non-blocking one: a null-stream memset is not ordered against it — a short launch right behind reset_stats lost a tenth of
    // its pair count to the memset landing late: tests/test_gpu_quality.py, round 4)
    DGE_HIP(hipMemsetAsync(m->d_counters, 0, 2 * sizeof(unsigned long long), m->stream));
    DGE_HIP(hipMemsetAsync(m->d_counters + 4, 0, 3 * sizeof(unsigned long long), m->stream));
    DGE_HIP(hipStreamSynchronize(m->stream));
    m->kernel_ms = 0; m->walk_ms = 0; m->launches = 0;
    return DGE_OK;
}

// dge_fmt_g9 (fmt_g9.h) against snprintf("%.9g") on `n` pseudo-random floats: half of them random bit patterns, half values of an embedding's range; host code only
extern "C" int dge_selftest_fmt_g9(int64_t n, uint64_t seed, int64_t* fast_path, int64_t* mismatches) {
    if (n < 0 || !fast_path || !mismatches) DGE_FAIL(DGE_ERR_ARG, "dge_selftest_fmt_g9: bad argument");
    int64_t fast = 0, bad = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    char a[64], b[64];
    for (int64_t i = 0; i < n; i++) {
        s = dge_mix64(s + (uint64_t)i);
        uint32_t u = (uint32_t)(s >> 32);
        float f;
        if (i & 1) memcpy(&f, &u, 4);
        else f = (float)(((double)(s & 0xFFFFFFFFull) / 4294967296.0 * 2.0 - 1.0) * ((i & 6) == 0 ? 1e-3 : ((i & 6) == 2 ? 1.0 : 40.0)));
        char* e = dge_fmt_g9(f, a);
        if (!e) continue;
        *e = 0; fast++;
        snprintf(b, sizeof(b), "%.9g", (double)f);
        if (strcmp(a, b) != 0) bad++;
    }
    *fast_path = fast; *mismatches = bad;
    return DGE_OK;
}

// WordVectorSerializer.writeWordVectors: V lines of D decimal numbers.  At the reference's sizes (6 408 x 20) that is nothing; at cfg3's (10^6 x 128 =
// 1.3e8 conversions, 1.5 GB of text) one thread formats for ~25 s — longer than the epoch trained.  Rows are formatted in slabs by up to 16 host threads
// (each row into its own string, the slab written in row order): same bytes as the serial loop.
extern "C" int dge_write_vec(dge_model* m, const char* const* names, const char* path, int header) {
    if (!m || !path) DGE_FAIL(DGE_ERR_ARG, "dge_write_vec: null argument");
    int rc = sync_tables_to_host(m, true, false);
    if (rc) return rc;
    FILE* f = fopen(path, "w");
    if (!f) DGE_FAIL(DGE_ERR_IO, "dge_write_vec: cannot open %s", path);
    if (header) fprintf(f, "%lld %d\n", (long long)m->V, m->D);
    const int64_t V = m->V; const int D = m->D;
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 16u), V * (int64_t)D / 65536));
    const int64_t slab = 4096 * (int64_t)n_thr;                                  // rows formatted before they are written
    // two sets of slab buffers: while slab k is being written (one thread, in row order), slab k + 1 is being formatted — the text file of cfg3 is 1.67 GB,
    // and writing it takes as long as formatting it
    std::vector<std::string> out[2] = {std::vector<std::string>((size_t)n_thr), std::vector<std::string>((size_t)n_thr)};
    bool ok = true;
    std::thread writer;
    int cur = 0;
    for (int64_t r0 = 0; r0 < V; r0 += slab, cur ^= 1) {
        const int64_t r1 = std::min(V, r0 + slab);
        std::vector<std::string>& ob = out[cur];
        auto work = [&, r0, r1](int t) {
            std::string& sbuf = ob[(size_t)t];
            const int64_t a = r0 + (r1 - r0) * t / n_thr, b = r0 + (r1 - r0) * (t + 1) / n_thr;
            // the rows' text goes straight into the buffer: at most 17 characters an element (sign, nine digits, point, e-XX, the blank in front)
            size_t cap = 0;
            for (int64_t r = a; r < b; r++) { const int32_t id = m->h_vocab_ids[(size_t)r]; cap += (names && names[id] ? strlen(names[id]) : 12) + (size_t)D * 26 + 2; }
            sbuf.resize(cap);
            char* o = sbuf.data();
            for (int64_t r = a; r < b; r++) {
                const int32_t id = m->h_vocab_ids[(size_t)r];
                if (names && names[id]) { const size_t n = strlen(names[id]); memcpy(o, names[id], n); o += n; } else o = std::to_chars(o, o + 12, id).ptr;
                const float* v = m->h_syn0.data() + r * D;
                // "%.9g" of every element: dge_fmt_g9 (integer arithmetic, the same bytes: fmt_g9.h) for the values an embedding holds, and for the rest
                // std::to_chars(double, general, 9), which is specified to give printf's "%.9g"
                for (int j = 0; j < D; j++) {
                    *o++ = ' ';
                    char* e = dge_fmt_g9(v[j], o);
                    o = e ? e : std::to_chars(o, o + 25, (double)v[j], std::chars_format::general, 9).ptr;
                }
                *o++ = '\n';
            }
            sbuf.resize((size_t)(o - sbuf.data()));
        };
        if (n_thr == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        }
        if (writer.joinable()) writer.join();                                   // the previous slab is on its way to the file: now this one
        if (!ok) break;
        writer = std::thread([&ok, &ob, f, n_thr]() {
            for (int t = 0; t < n_thr && ok; t++) ok = ob[(size_t)t].empty() || fwrite(ob[(size_t)t].data(), 1, ob[(size_t)t].size(), f) == ob[(size_t)t].size();
        });
    }
    if (writer.joinable()) writer.join();
    if (fclose(f) != 0 || !ok) DGE_FAIL(DGE_ERR_IO, "dge_write_vec: write to %s failed", path);
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

// `peer`: the caller's stream the buffer is produced / consumed on.  DGE_STREAM_BLOCKING (the synchronous entry points): a reader of a caller's buffer first waits for
// the whole device, a writer returns after the model's stream has drained.  Otherwise the copy is STREAM-ORDERED and the host never waits: an import makes the
// model's stream wait for what `peer` holds at the time of the call (an event), an export makes `peer` wait for the pack kernel.
#define DGE_STREAM_BLOCKING ((hipStream_t)(intptr_t)-1)
static int partition_copy(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf, bool pack, hipStream_t peer) {
    if (!m || !d_buf || n_parts <= 0 || part < 0 || part >= n_parts || table < 0 || table > 2) DGE_FAIL(DGE_ERR_ARG, "dge_model_%s_partition: bad argument", pack ? "export" : "import");
    if (table == 2 && !m->d_syn1) DGE_FAIL(DGE_ERR_STATE, "dge_model_%s_partition: table 2 (syn1) exists with use_hs only", pack ? "export" : "import");
    DGE_HIP(hipSetDevice(m->device));
    const bool blocking = peer == DGE_STREAM_BLOCKING;
    const bool handshake = !blocking && peer != m->stream;
    if (handshake && !m->ev_peer) DGE_HIP(hipEventCreateWithFlags(&m->ev_peer, hipEventDisableTiming));
    if (!pack) {
        if (blocking) DGE_HIP(hipDeviceSynchronize());          // d_buf comes from the caller's collective, on the caller's stream
        else if (handshake) { DGE_HIP(hipEventRecord(m->ev_peer, peer)); DGE_HIP(hipStreamWaitEvent(m->stream, m->ev_peer, 0)); }
    }
    float* tab = table == 0 ? m->d_syn0 : (table == 1 ? m->d_syn1neg : m->d_syn1);
    const int64_t rows = (m->V + n_parts - 1) / n_parts;
    if (rows > 0) {
        if (pack) hipLaunchKernelGGL(k_partition_pack, dim3(2048), dim3(256), 0, m->stream, tab, d_buf, m->V, m->stride, n_parts, part, rows);
        else hipLaunchKernelGGL(k_partition_unpack, dim3(2048), dim3(256), 0, m->stream, tab, d_buf, m->V, m->stride, n_parts, part, rows);
    }
    DGE_HIP(hipGetLastError());
    if (blocking) DGE_HIP(hipStreamSynchronize(m->stream));
    else if (handshake && pack) { DGE_HIP(hipEventRecord(m->ev_peer, m->stream)); DGE_HIP(hipStreamWaitEvent(peer, m->ev_peer, 0)); }
    return DGE_OK;
}
extern "C" int dge_model_export_partition(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf) { return partition_copy(m, table, n_parts, part, d_buf, true, DGE_STREAM_BLOCKING); }
extern "C" int dge_model_import_partition(dge_model* m, int table, int32_t n_parts, int32_t part, const float* d_buf) { return partition_copy(m, table, n_parts, part, const_cast<float*>(d_buf), false, DGE_STREAM_BLOCKING); }
extern "C" int dge_model_export_partition_async(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf, void* consumer_stream) {
    return partition_copy(m, table, n_parts, part, d_buf, true, (hipStream_t)consumer_stream);
}
extern "C" int dge_model_import_partition_async(dge_model* m, int table, int32_t n_parts, int32_t part, const float* d_buf, void* producer_stream) {
    return partition_copy(m, table, n_parts, part, const_cast<float*>(d_buf), false, (hipStream_t)producer_stream);
}
extern "C" int dge_model_stream(const dge_model* m, void** hip_stream) {
    if (!m || !hip_stream) DGE_FAIL(DGE_ERR_ARG, "dge_model_stream: null argument");
    *hip_stream = (void*)m->stream;
    return DGE_OK;
}

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
#include <rccl/rccl.h>      // types and enum values only (ncclComm_t, ncclUniqueId, ncclFloat32, ncclSum, ncclResult_t): no RCCL symbol is linked
static_assert(sizeof(dge_unique_id) == sizeof(ncclUniqueId), "include/dge.h: dge_unique_id must be the size of ncclUniqueId");
struct dge_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, nranks = 1, device = 0;
    float* d_buf = nullptr; int64_t buf_floats = 0;
};
namespace {
// the entry points are looked up with dlsym at first use; their prototypes are the header's own (decltype), so a change of rccl.h shows at compile time
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.lib) return DGE_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) DGE_FAIL(DGE_ERR_DEVICE, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(&ncclGetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(&ncclCommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(&ncclAllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.AllGather = (decltype(&ncclAllGather))dlsym(h, "ncclAllGather");
    g_rccl.Send = (decltype(&ncclSend))dlsym(h, "ncclSend");
    g_rccl.Recv = (decltype(&ncclRecv))dlsym(h, "ncclRecv");
    g_rccl.GroupStart = (decltype(&ncclGroupStart))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(&ncclGroupEnd))dlsym(h, "ncclGroupEnd");
    g_rccl.CommDestroy = (decltype(&ncclCommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(&ncclGetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy || !g_rccl.Send || !g_rccl.Recv ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd) DGE_FAIL(DGE_ERR_DEVICE, "librccl lacks an expected symbol");
    g_rccl.lib = h;
    return DGE_OK;
}
int rccl_fail(int rc, const char* what) {
    DGE_FAIL(DGE_ERR_DEVICE, "RCCL %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString((ncclResult_t)rc) : "?");
}
}  // namespace

extern "C" int dge_comm_unique_id(dge_unique_id* out) {
    if (!out) DGE_FAIL(DGE_ERR_ARG, "dge_comm_unique_id: null output");
    int rc = rccl_load();
    if (rc) return rc;
    int n = g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(out));
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
    ncclUniqueId nid; memcpy(&nid, id, sizeof(nid));
    int n = g_rccl.CommInitRank(&c->nccl, nranks, nid, rank);
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
    int n = g_rccl.AllReduce(c->d_buf, c->d_buf, (size_t)nfl, ncclFloat32, ncclSum, c->nccl, m->stream);
    if (n) return rccl_fail(n, "ncclAllReduce");
    DGE_HIP(hipStreamSynchronize(m->stream));
    return dge_model_import_delta(m, c->d_buf, 1.0f / (float)c->nranks);
}

// block schedule with RCCL called from the library.  dge_model_ring_pass: after episode `episode` rank g hands the syn1neg partition
// it just trained, (g + episode) % N, to rank g-1 and takes partition (g + 1 + episode) % N — the one it trains next — from rank
// g+1 (ncclSend/ncclRecv in one group).  dge_model_gather_table: every rank publishes partition `rank` of `table` and takes the
// others (all-gather): the end of training, or a checkpoint.
static int comm_buffers(dge_model* m, dge_comm* c, int64_t need) {
    if (c->buf_floats >= need) return DGE_OK;
    dge_dev_free(c->d_buf); c->d_buf = nullptr; c->buf_floats = 0;
    int rc = dge_dev_alloc(&c->d_buf, (size_t)need + 64);
    if (rc) return rc;
    c->buf_floats = need;
    return DGE_OK;
}

extern "C" int dge_model_ring_pass(dge_model* m, dge_comm* c, int32_t episode) {
    if (!m || !c || episode < 0) DGE_FAIL(DGE_ERR_ARG, "dge_model_ring_pass: bad argument");
    if (c->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_ring_pass: communicator and model live on different devices");
    if (c->nranks == 1) return DGE_OK;
    DGE_HIP(hipSetDevice(m->device));
    int64_t pf = 0;
    int rc = dge_model_partition_floats(m, c->nranks, &pf);
    if (rc) return rc;
    const int n_tab = m->d_syn1 ? 2 : 1;                   // with the hierarchical softmax the syn1 partition of the same number travels along
    if ((rc = comm_buffers(m, c, 2 * pf * n_tab))) return rc;
    float* mine = c->d_buf; float* next = c->d_buf + pf * n_tab;
    // everything below is enqueued on the model's stream — pack, ncclSend / ncclRecv, unpack — and the host never waits: the next episode's launches queue up behind
    for (int t = 0; t < n_tab; t++)
        if ((rc = partition_copy(m, 1 + t, c->nranks, (c->rank + episode) % c->nranks, mine + t * pf, true, m->stream))) return rc;
    const int dst = (c->rank + c->nranks - 1) % c->nranks, src = (c->rank + 1) % c->nranks;
    int n = g_rccl.GroupStart();
    if (!n) n = g_rccl.Send(mine, (size_t)(pf * n_tab), ncclFloat32, dst, c->nccl, m->stream);
    if (!n) n = g_rccl.Recv(next, (size_t)(pf * n_tab), ncclFloat32, src, c->nccl, m->stream);
    const int n2 = g_rccl.GroupEnd();
    if (n || n2) return rccl_fail(n ? n : n2, "ncclSend/ncclRecv");
    for (int t = 0; t < n_tab; t++)
        if ((rc = partition_copy(m, 1 + t, c->nranks, (c->rank + 1 + episode) % c->nranks, next + t * pf, false, m->stream))) return rc;
    return DGE_OK;
}

extern "C" int dge_model_gather_table(dge_model* m, dge_comm* c, int table) {
    if (!m || !c || table < 0 || table > 2 || (table == 2 && !m->d_syn1)) DGE_FAIL(DGE_ERR_ARG, "dge_model_gather_table: bad argument");
    if (c->device != m->device) DGE_FAIL(DGE_ERR_ARG, "dge_model_gather_table: communicator and model live on different devices");
    DGE_HIP(hipSetDevice(m->device));
    int64_t pf = 0;
    int rc = dge_model_partition_floats(m, c->nranks, &pf);
    if (rc) return rc;
    if ((rc = comm_buffers(m, c, pf * ((int64_t)c->nranks + 1)))) return rc;
    float* mine = c->d_buf; float* all = c->d_buf + pf;
    if ((rc = partition_copy(m, table, c->nranks, c->rank, mine, true, m->stream))) return rc;
    int n = g_rccl.AllGather(mine, all, (size_t)pf, ncclFloat32, c->nccl, m->stream);
    if (n) return rccl_fail(n, "ncclAllGather");
    for (int r = 0; r < c->nranks; r++)
        if (r != c->rank && (rc = partition_copy(m, table, c->nranks, r, all + (int64_t)r * pf, false, m->stream))) return rc;
    DGE_HIP(hipStreamSynchronize(m->stream));              // (end of training: the caller reads the tables next)
    return DGE_OK;
}
