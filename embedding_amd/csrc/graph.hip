// graph.hip — per-timeslice edge store, alias tables and the walk sampler of libdge.so (gfx950).
//
// Replaces J/LayeredGraph.java (edge store :142-189, alias tables :54-82,195-226, sampler :104-116,
// 232-252) and SpatialGraph.keepNearestKVertices (J/SpatialGraph.java:29-35).  Data layout in HBM:
//   row_ptr int64[V+1] | nbr int32[E] | w f64[E] | outdeg f64[V] | prob f64[E] | alias int32[E]
//   slots {f64 prob, i32 nbr, i32 nbr_alias}[E]   <- what the walk kernel reads: ONE 16-B load per step
// Vertex id = h*R + r for time-sliced graphs, so slice h's rows are contiguous and walks that advance in
// lock-step read one slice at a time.
#include <hipcub/hipcub.hpp>
#include <stdarg.h>

#include "dge_algos.h"
#include "dge_internal.h"

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

void dge_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char* dge_last_error(void) { return g_last_error.c_str(); }
extern "C" int dge_version(void) { return DGE_VERSION; }
uint64_t dge_next_generation() { static uint64_t g = 0; return ++g; }

extern "C" int dge_device_count(int* n) {
    if (!n) DGE_FAIL(DGE_ERR_ARG, "dge_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; DGE_FAIL(DGE_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorName(e)); }
    *n = c;
    return DGE_OK;
}

int dge_require_device(int device) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0)
        DGE_FAIL(DGE_ERR_DEVICE, "libdge has no CPU path: no HIP device visible (%s)", hipGetErrorName(e));
    if (device < 0 || device >= c) DGE_FAIL(DGE_ERR_DEVICE, "device %d out of range (0..%d); libdge has no CPU path", device, c - 1);
    hipDeviceProp_t prop;
    DGE_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        DGE_FAIL(DGE_ERR_DEVICE, "device %d is %s; libdge is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    DGE_HIP(hipSetDevice(device));
    return DGE_OK;
}

// ------------------------------------------------------------------------------------------ small kernels
__global__ void k_iota_u32(uint32_t* p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

__global__ void k_gather_edges(const uint32_t* idx, const int32_t* dst, const double* w, int32_t* nbr, double* wo, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint32_t e = idx[i]; nbr[i] = dst[e]; wo[i] = w[e]; }
}

// row_ptr[v] = first position whose sorted source id is >= v
__global__ void k_row_ptr(const int32_t* sorted_src, int64_t E, int32_t V, int64_t* row_ptr) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > V) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (sorted_src[mid] < (int32_t)v) lo = mid + 1; else hi = mid; }
    row_ptr[v] = lo;
}

// Vertex.addOutEdge: outDegree += weight in insertion order (J/LayeredGraph.java:46-49)
__global__ void k_out_degree(const int64_t* row_ptr, const double* w, int32_t V, double* outdeg) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    double s = 0.0;
    for (int64_t e = row_ptr[v]; e < row_ptr[v + 1]; e++) s += w[e];
    outdeg[v] = s;
}

__global__ void k_max_i32(const int32_t* a, const int32_t* b, int64_t n, int32_t* out) {
    int32_t m = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        m = max(m, a[i]); m = max(m, b[i]);
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__global__ void k_min_i32_edges(const int32_t* a, const int32_t* b, int64_t n, int32_t* out) {
    int32_t m = 0x7fffffff;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        m = min(m, a[i]); m = min(m, b[i]);
    }
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}

__global__ void k_min_degree(const int64_t* row_ptr, int32_t V, unsigned long long* out) {
    unsigned long long m = ~0ULL;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x)
        m = min(m, (unsigned long long)(row_ptr[v + 1] - row_ptr[v]));
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned long long)__shfl_xor((long long)m, o));
    if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}

// keepNearestKVertices: after the stable descending segmented sort, keep slots [0,k) of every vertex and
// recompute outDegree with DoubleStream.sum() (J/SpatialGraph.java:31-33)
__global__ void k_topk_compact(const int64_t* row_ptr, const double* w_sorted, const int32_t* nbr_sorted, int32_t V,
                               int32_t k, int64_t* new_row_ptr, double* w_out, int32_t* nbr_out, double* outdeg) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > V) return;
    new_row_ptr[v] = v * (int64_t)k;
    if (v == V) return;
    int64_t b = row_ptr[v], nb = v * (int64_t)k;
    for (int32_t j = 0; j < k; j++) { w_out[nb + j] = w_sorted[b + j]; nbr_out[nb + j] = nbr_sorted[b + j]; }
    outdeg[v] = dge_java8_stream_sum(w_out + nb, k);
}

// bulk addSourceVertex (J/LayeredGraph.java:180-189).  The weights are gathered in parallel; sourceWeightSum is a running `+=` in
// source order (or DoubleStream.sum(), a Kahan sum) — a sequential double sum by definition, taken by one lane over the gathered,
// contiguous weights (the dependent gathers were what made the one-thread form slow: 14 ms at 100 k sources).
__global__ void k_sources_gather(const int32_t* __restrict__ srcv, int64_t S, const double* __restrict__ outdeg, double* __restrict__ src_w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) src_w[i] = outdeg[srcv[i]];
}
// lane idx's value of a wave-uniform idx, on every lane
__device__ __forceinline__ int32_t wave_pick(int32_t v, int idx) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx)); }
__device__ __forceinline__ double wave_pick(double v, int idx) {
    const int i = __builtin_amdgcn_readfirstlane(idx);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), i), __builtin_amdgcn_readlane(__double2loint(v), i));
}
// sourceWeightSum: the additions are a chain (plain left to right, or DoubleStream.sum's compensated form, dge_java8_stream_sum) and stay
// one; the wave loads 64 terms at a time and every lane runs the chain on registers.
__global__ void __launch_bounds__(64) k_sources_sum(const double* __restrict__ src_w, int64_t S, int stream_sum, double* sum_out) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x & 63;
    double sum = 0.0, comp = 0.0, simple = 0.0;
    for (int64_t c = 0; c < S; c += 64) {
        const double mine = c + lane < S ? src_w[c + lane] : 0.0;
        const int cnt = (int)min((int64_t)64, S - c);
        for (int j = 0; j < cnt; j++) {
            const double x = wave_pick(mine, j);
            if (stream_sum) {
                const double tmp = x - comp;
                const double velvel = sum + tmp;
                comp = (velvel - sum) - tmp;
                sum = velvel;
            }
            simple += x;
        }
    }
    double r = simple;
    if (stream_sum) {
        const double tmp = sum + comp;
        r = (tmp != tmp && (simple - simple) != 0.0 && simple == simple) ? simple : tmp;      // NaN result, infinite simple sum
    }
    if (lane == 0) *sum_out = r;
}
static void launch_sources(hipStream_t st, const int32_t* srcv, int64_t S, const double* outdeg, double* src_w, int stream_sum, double* sum_out) {
    if (S > 0) hipLaunchKernelGGL(k_sources_gather, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, srcv, S, outdeg, src_w);
    hipLaunchKernelGGL(k_sources_sum, dim3(1), dim3(64), 0, st, src_w, S, stream_sum, sum_out);
}

// dge_alias_vose (dge_algos.h) by one WAVE, for tables of thousands of slots (hub vertices of a power-law graph, the source table): the same
// stacks, the same pops and pushes in the same order, the same arithmetic — but every value the serial loop would fetch through two dependent
// loads per step (a stack entry, then its prob) comes from a 64-entry register window over the top of its stack, refilled by all lanes at once.
// What a pop can see that a window cannot hold — the large that has just dropped below 1 and lies on top of the small stack — stays in
// registers (`pend`).  All 64 lanes must call it with the same arguments; control flow is wave-uniform.
#define ALIAS_WAVE_MIN 1024         /* tables from this size on (Vose order) take the wave form */
__device__ void alias_vose_wave(const double* __restrict__ w, int64_t k, double total, double* prob, int32_t* alias, int32_t* scratch) {
    const int lane = threadIdx.x & 63;
    const uint64_t below = (1ull << lane) - 1ull;
    int64_t ns = 0, nl = 0;
    for (int64_t c = 0; c < k; c += 64) {                     // the stacks in slot order: smalls up from scratch[0], larges down from scratch[k-1]
        const int64_t i = c + lane;
        double p = 2.0;
        if (i < k) { p = (double)k * w[i] / total; alias[i] = -1; prob[i] = p; }
        const uint64_t ms = __ballot(i < k && p < 1.0), ml = __ballot(i < k && !(p < 1.0));
        if (i < k) {
            if (p < 1.0) scratch[ns + __popcll(ms & below)] = (int32_t)i;
            else scratch[k - 1 - (nl + __popcll(ml & below))] = (int32_t)i;
        }
        ns += __popcll(ms); nl += __popcll(ml);
    }
    __threadfence_block();
    // windows: lane j of the small window holds stack entry sw_top - j, lane j of the large window entry lw_top + j
    int64_t sw_top = -1, lw_top = -1; int sw_n = 0, lw_n = 0, sw_i = 0, lw_i = 0;
    int32_t sw_id = 0, lw_id = 0; double sw_p = 0.0, lw_p = 0.0;
    bool has_pend = false; int32_t pend_id = 0; double pend_p = 0.0;
    while (ns > 0 && nl > 0) {
        // pop a large
        if (!(lw_i < lw_n && lw_top + lw_i == k - nl)) {
            lw_top = k - nl; lw_n = (int)min((int64_t)64, nl); lw_i = 0;
            if (lane < lw_n) { lw_id = scratch[lw_top + lane]; lw_p = prob[lw_id]; }
        }
        const int32_t l = wave_pick(lw_id, lw_i); double pl = wave_pick(lw_p, lw_i);
        lw_i++; nl--;
        for (;;) {
            // pop a small
            int32_t sid; double sp;
            if (has_pend) { sid = pend_id; sp = pend_p; has_pend = false; }
            else {
                if (!(sw_i < sw_n && sw_top - sw_i == ns - 1)) {
                    sw_top = ns - 1; sw_n = (int)min((int64_t)64, ns); sw_i = 0;
                    if (lane < sw_n) { sw_id = scratch[sw_top - lane]; sw_p = prob[sw_id]; }
                }
                sid = wave_pick(sw_id, sw_i); sp = wave_pick(sw_p, sw_i);
                sw_i++;
            }
            ns--;
            if (lane == 0) alias[sid] = l;
            pl = (pl + sp) - 1.0;
            if (pl < 1.0) {                                  // the large has become a small: on top of the small stack, taken next
                if (lane == 0) { prob[l] = pl; scratch[ns] = l; }
                ns++; has_pend = true; pend_id = l; pend_p = pl;
                break;
            }
            if (ns == 0) {                                   // no small left: back on its stack
                if (lane == 0) { prob[l] = pl; scratch[k - 1 - nl] = l; }
                nl++;
                break;
            }
        }
    }
    __threadfence_block();
    for (int64_t j = lane; j < ns; j += 64) prob[scratch[j]] = 1.0;            // left-overs: full slots
    for (int64_t j = lane; j < nl; j += 64) prob[scratch[k - nl + j]] = 1.0;
}

// Vertex.initiateAliasTable for every vertex: one lane per table (J/LayeredGraph.java:197); hub_min > 0: tables of hub_min slots and more
// are left to k_alias_hubs
__global__ void k_alias_vertices(int32_t V, const int64_t* row_ptr, const double* w, const double* outdeg,
                                 double* prob, int32_t* alias, int exact, uint64_t* bs_scratch, int32_t* vose_scratch, int64_t hub_min) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int64_t b = row_ptr[v], k = row_ptr[v + 1] - b;
    if (k == 0 || (hub_min > 0 && k >= hub_min)) return;
    if (exact) dge_alias_reference(w + b, k, outdeg[v], prob + b, alias + b, bs_scratch + 2 * (b / 32 + 6 * v));
    else       dge_alias_vose(w + b, k, outdeg[v], prob + b, alias + b, vose_scratch + b);
}
__global__ void k_hub_list(int32_t V, const int64_t* row_ptr, int64_t hub_min, int32_t* hubs, int32_t* n_hubs) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V && row_ptr[v + 1] - row_ptr[v] >= hub_min) hubs[atomicAdd(n_hubs, 1)] = (int32_t)v;
}
__global__ void __launch_bounds__(64) k_alias_hubs(const int32_t* hubs, const int64_t* row_ptr, const double* w, const double* outdeg, double* prob,
                                                   int32_t* alias, int32_t* vose_scratch) {
    const int32_t v = hubs[blockIdx.x];
    const int64_t b = row_ptr[v];
    alias_vose_wave(w + b, row_ptr[v + 1] - b, outdeg[v], prob + b, alias + b, vose_scratch + b);
}

// the same pairing over the source vertices, weight = outDegree (J/LayeredGraph.java:199-225)
__global__ void __launch_bounds__(64) k_alias_sources(int64_t S, const double* src_w, double total, double* prob, int32_t* alias, int exact,
                                                      uint64_t* bs_scratch, int32_t* vose_scratch) {
    if (blockIdx.x != 0 || S == 0) return;
    if (!exact && S >= ALIAS_WAVE_MIN) { alias_vose_wave(src_w, S, total, prob, alias, vose_scratch); return; }
    if (threadIdx.x != 0) return;
    if (exact) dge_alias_reference(src_w, S, total, prob, alias, bs_scratch);
    else       dge_alias_vose(src_w, S, total, prob, alias, vose_scratch);
}

// The walk slots of a finished table, one thread per slot (the pairing above is serial per table; this is not): slot e of the table that
// begins at row_ptr[owner[e]] (owner == nullptr: one table from 0, the sources) names its neighbour, its alias' neighbour ("no alias" ->
// stay in slot e) and both neighbours' own row bounds.
__global__ void k_owner_mark(int32_t V, const int64_t* row_ptr, int32_t* owner) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V && row_ptr[v + 1] > row_ptr[v]) owner[row_ptr[v]] = (int32_t)v;
}
__global__ void k_fill_slots(int64_t n, const int32_t* __restrict__ owner, const int64_t* __restrict__ row_ptr, const double* __restrict__ prob,
                             const int32_t* __restrict__ alias, const int32_t* __restrict__ ids, dge_slot* __restrict__ slots) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t b = owner ? row_ptr[owner[e]] : 0;
    const int32_t a = alias[e];
    dge_slot s;
    s.prob = prob[e];
    s.nbr = ids[e];
    s.nbr_alias = a >= 0 ? ids[b + a] : s.nbr;
    s.base = (uint32_t)row_ptr[s.nbr]; s.k = (uint32_t)(row_ptr[s.nbr + 1] - row_ptr[s.nbr]);
    s.base_alias = (uint32_t)row_ptr[s.nbr_alias]; s.k_alias = (uint32_t)(row_ptr[s.nbr_alias + 1] - row_ptr[s.nbr_alias]);
    slots[e] = s;
}
struct MaxI32 { __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; } };

// ------------------------------------------------------------------------------------------ walk sampler
// One draw (J/LayeredGraph.java:104-116): x -> slot -> neighbour, and the neighbour's own row bounds (b, k) for the next draw.
// One 32-byte slot: the step's only memory access.
__device__ __forceinline__ int32_t walk_pick(const dge_slot* __restrict__ slots, int64_t base, int64_t k, double x, int64_t& nb, int64_t& nk) {
    double y;
    int64_t i = dge_alias_slot(x, k, &y);
    const uint4* p = reinterpret_cast<const uint4*>(slots + base + i);
    const uint4 lo = p[0], hi = p[1];
    const double prob = __hiloint2double((int)lo.y, (int)lo.x);
    const bool first = y < prob;
    nb = first ? hi.x : hi.z; nk = first ? hi.y : hi.w;
    return (int32_t)(first ? lo.z : lo.w);
}

// sampleVertexSequence (J/LayeredGraph.java:232-252), one lane per walk.  Walk i consumes the draws
// starting at draw offset (first_index + i) * L of java.util.Random(seed) ("strided" streams), or at
// d_offsets[i] when given.  The block's rows are staged in LDS and leave as one contiguous, coalesced
// store (row length padded to an odd word count -> conflict-free LDS writes).
#define WALK_BLOCK 256
__global__ void __launch_bounds__(WALK_BLOCK)
k_walks(const int64_t* __restrict__ row_ptr, const dge_slot* __restrict__ slots, const dge_slot* __restrict__ src_slots,
        int64_t S, int32_t* __restrict__ out, int64_t n, int32_t L, uint64_t seed0, int64_t base_draw, int64_t draw_stride,
        const int64_t* __restrict__ d_offsets, uint8_t* __restrict__ lens_out, int32_t* deadend_count) {
    extern __shared__ int32_t lds[];
    const int t = threadIdx.x;
    const int64_t blk0 = (int64_t)blockIdx.x * WALK_BLOCK;
    const int64_t i = blk0 + t;
    const int Lp = L | 1;
    if (i < n) {
        uint64_t off = d_offsets ? (uint64_t)d_offsets[i] : (uint64_t)base_draw + (uint64_t)i * (uint64_t)draw_stride;
        uint64_t s = dge_jr_jump(seed0, 2ULL * off);
        int32_t* row = lds + t * Lp;
        int len = 0;
        if (S > 0) {
            double x = dge_jr_next_double(s);
            int64_t b, k;
            int32_t v = walk_pick(src_slots, 0, S, x, b, k);
            row[0] = v;
            len = 1;
            for (; len < L; len++) {
                if (k == 0) break;                      // dead end: no draw (J/LayeredGraph.java:106-107)
                x = dge_jr_next_double(s);
                v = walk_pick(slots, b, k, x, b, k);
                row[len] = v;
            }
        }
        if (len < L && deadend_count) atomicAdd(deadend_count, 1);
        if (lens_out) lens_out[i] = (uint8_t)len;
        for (int j = len; j < L; j++) row[j] = -1;
    }
    if (!out) return;              // length-only pass of the sequential-stream resolver
    __syncthreads();
    const int64_t rows = min((int64_t)WALK_BLOCK, n - blk0);
    const int64_t cnt = rows * L;
    int32_t* dst = out + blk0 * L;
    for (int64_t idx = t; idx < cnt; idx += WALK_BLOCK) {
        int r = (int)(idx / L), c = (int)(idx - (int64_t)r * L);
        dst[idx] = lds[r * Lp + c];
    }
}

// The reference's shared sequential stream when some walk dead-ends (draw count per walk becomes data
// dependent, J/LayeredGraph.java:247-248): walks are chained, so one lane walks them in order.
__global__ void k_walks_sequential(const int64_t* row_ptr, const dge_slot* slots, const dge_slot* src_slots, int64_t S,
                                   int32_t* out, int64_t n, int32_t L, uint64_t seed0, int64_t first_draw, int64_t* draws_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t s = dge_jr_jump(seed0, 2ULL * (uint64_t)first_draw);
    int64_t draws = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t* row = out + i * L;
        int len = 0;
        if (S > 0 && L > 0) {
            double x = dge_jr_next_double(s); draws++;
            int64_t b, k;
            int32_t v = walk_pick(src_slots, 0, S, x, b, k);
            row[0] = v; len = 1;
            for (; len < L; len++) {
                if (k == 0) break;
                x = dge_jr_next_double(s); draws++;
                v = walk_pick(slots, b, k, x, b, k);
                row[len] = v;
            }
        }
        for (int j = len; j < L; j++) row[j] = -1;
    }
    *draws_out = draws;
}

// ---- the reference's shared sequential stream with dead ends, in parallel -------------------------------------------
// Walk i starts at draw o_i with o_0 = first_draw and o_{i+1} = o_i + len(o_i), where len(o) = draws a walk started at
// draw o consumes (1..L, data dependent: J/LayeredGraph.java:247-248).  (1) len(o) for EVERY o in [0, n*L) (k_walks in
// length-only mode, draw stride 1); (2) the offset range is cut into blocks; from each of the L possible entry points
// of a block one lane follows o -> o+len(o) to the block's end: exit point and hop count; (3) one lane chains the
// blocks (entry of block b+1 = exit of block b); (4) one lane per block re-follows its chain and writes the start
// draw of every walk; (5) k_walks with explicit offsets.  All exact: the walks are those of the sequential loop.
#define SEQ_BLOCK 4096
__global__ void k_seq_block_exits(const uint8_t* __restrict__ lens, int64_t N, int32_t L, int64_t n_blocks, int32_t* exit_e, int32_t* hops) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks * L) return;
    const int64_t b = t / L; const int e = (int)(t - b * L);
    const int64_t end = min((b + 1) * (int64_t)SEQ_BLOCK, N);
    int64_t o = b * (int64_t)SEQ_BLOCK + e; int32_t cnt = 0;
    while (o < end) { o += lens[o]; cnt++; }
    exit_e[t] = (int32_t)(o - end); hops[t] = cnt;
}
__global__ void k_seq_chain(const int32_t* exit_e, const int32_t* hops, int32_t L, int64_t n_blocks, int32_t* start_e, int64_t* start_idx) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int e = 0; int64_t idx = 0;
    for (int64_t b = 0; b < n_blocks; b++) {
        start_e[b] = e; start_idx[b] = idx;
        idx += hops[b * L + e]; e = exit_e[b * L + e];
    }
}
__global__ void k_seq_starts(const uint8_t* __restrict__ lens, int64_t N, int64_t n_blocks, const int32_t* start_e, const int64_t* start_idx,
                             int64_t first_draw, int64_t n_walks, int64_t* starts, int64_t* draws_out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const int64_t end = min((b + 1) * (int64_t)SEQ_BLOCK, N);
    int64_t o = b * (int64_t)SEQ_BLOCK + start_e[b], idx = start_idx[b];
    while (o < end && idx < n_walks) {
        starts[idx] = first_draw + o;
        if (idx == n_walks - 1) *draws_out = o + lens[o];
        o += lens[o]; idx++;
    }
}

__global__ void k_sample_next(const int64_t* row_ptr, const dge_slot* slots, int32_t v, double x, int32_t* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int64_t b = row_ptr[v], k = row_ptr[v + 1] - b;
    int64_t nb, nk;
    *out = k == 0 ? -1 : walk_pick(slots, b, k, x, nb, nk);
}

__global__ void k_position_prefix(int32_t* walks, int64_t n, int32_t L, int32_t R) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * L) return;
    int32_t j = (int32_t)(i % L);
    int32_t t = walks[i];
    if (t >= 0) walks[i] = j * R + t;
}

// ------------------------------------------------------------------------------------------ host side
static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

extern "C" int dge_graph_create(dge_graph** out, int device) {
    if (!out) DGE_FAIL(DGE_ERR_ARG, "dge_graph_create: null output");
    *out = nullptr;
    int rc = dge_require_device(device);
    if (rc) return rc;
    dge_graph* g = new dge_graph();
    g->device = device;
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { delete g; DGE_FAIL(DGE_ERR_DEVICE, "dge_graph_create: hipStreamCreate failed"); }
    g->own_stream = true;
    *out = g;
    return DGE_OK;
}

static void free_csr(dge_graph* g) {
    dge_dev_free(g->d_row_ptr); dge_dev_free(g->d_nbr); dge_dev_free(g->d_w); dge_dev_free(g->d_outdeg);
    dge_dev_free(g->d_prob); dge_dev_free(g->d_alias); dge_dev_free(g->d_slots);
    g->d_row_ptr = nullptr; g->d_nbr = nullptr; g->d_w = nullptr; g->d_outdeg = nullptr;
    g->d_prob = nullptr; g->d_alias = nullptr; g->d_slots = nullptr;
    g->csr_built = false; g->alias_built = false;
}
static void free_sources(dge_graph* g) {
    dge_dev_free(g->d_srcv); dge_dev_free(g->d_src_w); dge_dev_free(g->d_src_prob); dge_dev_free(g->d_src_alias);
    dge_dev_free(g->d_src_slots);
    g->d_srcv = nullptr; g->d_src_w = nullptr; g->d_src_prob = nullptr; g->d_src_alias = nullptr; g->d_src_slots = nullptr;
    g->S = 0;
}

extern "C" void dge_graph_free(dge_graph* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    free_csr(g); free_sources(g);
    dge_dev_free(g->d_coo_src); dge_dev_free(g->d_coo_dst); dge_dev_free(g->d_coo_w);
    if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

extern "C" int dge_graph_set_stream(dge_graph* g, void* hip_stream) {
    if (!g) DGE_FAIL(DGE_ERR_ARG, "dge_graph_set_stream: null graph");
    DGE_HIP(hipSetDevice(g->device));
    DGE_HIP(hipStreamSynchronize(g->stream));
    if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
    g->stream = (hipStream_t)hip_stream;
    g->own_stream = false;
    return DGE_OK;
}

static int coo_reserve(dge_graph* g, int64_t extra) {
    if (g->n_coo + extra <= g->cap_coo) return DGE_OK;
    int64_t nc = (g->n_coo + extra) + (g->n_coo + extra) / 2 + 1024;
    dge_tmp<int32_t> ns, nd; dge_tmp<double> nw;
    int rc;
    if ((rc = ns.alloc((size_t)nc))) return rc;
    if ((rc = nd.alloc((size_t)nc))) return rc;
    if ((rc = nw.alloc((size_t)nc))) return rc;
    if (g->n_coo) {
        DGE_HIP(hipMemcpyAsync(ns, g->d_coo_src, g->n_coo * sizeof(int32_t), hipMemcpyDeviceToDevice, g->stream));
        DGE_HIP(hipMemcpyAsync(nd, g->d_coo_dst, g->n_coo * sizeof(int32_t), hipMemcpyDeviceToDevice, g->stream));
        DGE_HIP(hipMemcpyAsync(nw, g->d_coo_w, g->n_coo * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
    }
    dge_dev_free(g->d_coo_src); dge_dev_free(g->d_coo_dst); dge_dev_free(g->d_coo_w);
    g->d_coo_src = ns.release(); g->d_coo_dst = nd.release(); g->d_coo_w = nw.release(); g->cap_coo = nc;
    return DGE_OK;
}

static int add_edges_common(dge_graph* g, const int32_t* src, const int32_t* dst, const double* w, int64_t n, hipMemcpyKind kind) {
    if (!g || n < 0 || (n > 0 && (!src || !dst || !w))) DGE_FAIL(DGE_ERR_ARG, "dge_graph_add_edges: bad argument");
    if (n == 0) return DGE_OK;
    if (g->csr_built && g->E != g->n_coo)
        DGE_FAIL(DGE_ERR_STATE, "dge_graph_add_edges: edges cannot be added after keep_top_k pruned the store");
    DGE_HIP(hipSetDevice(g->device));
    // device sources come from the caller's streams (e.g. torch's): wait for whatever produced them — this handle's
    // stream is non-blocking and would otherwise race with the producer
    if (kind == hipMemcpyDeviceToDevice) DGE_HIP(hipDeviceSynchronize());
    int rc = coo_reserve(g, n);
    if (rc) return rc;
    DGE_HIP(hipMemcpyAsync(g->d_coo_src + g->n_coo, src, n * sizeof(int32_t), kind, g->stream));
    DGE_HIP(hipMemcpyAsync(g->d_coo_dst + g->n_coo, dst, n * sizeof(int32_t), kind, g->stream));
    DGE_HIP(hipMemcpyAsync(g->d_coo_w + g->n_coo, w, n * sizeof(double), kind, g->stream));
    // id range check + vertex count (vertex ids are insertion ordinals: V = max id + 1)
    dge_tmp<int32_t> d_mm;
    if ((rc = d_mm.alloc(2))) return rc;
    int32_t init[2] = {-1, 0x7fffffff};
    DGE_HIP(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, g->stream));
    unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_max_i32, dim3(blocks), dim3(256), 0, g->stream, g->d_coo_src + g->n_coo, g->d_coo_dst + g->n_coo, n, d_mm);
    hipLaunchKernelGGL(k_min_i32_edges, dim3(blocks), dim3(256), 0, g->stream, g->d_coo_src + g->n_coo, g->d_coo_dst + g->n_coo, n, d_mm + 1);
    int32_t mm[2];
    DGE_HIP(hipMemcpyAsync(mm, d_mm, sizeof(mm), hipMemcpyDeviceToHost, g->stream));
    DGE_HIP(hipStreamSynchronize(g->stream));
    if (mm[1] < 0) DGE_FAIL(DGE_ERR_RANGE, "dge_graph_add_edges: negative vertex id %d", mm[1]);
    if (mm[0] == 0x7fffffff) DGE_FAIL(DGE_ERR_RANGE, "dge_graph_add_edges: vertex id overflow");
    g->max_id = std::max(g->max_id, mm[0]);
    g->n_coo += n;
    free_csr(g);
    return DGE_OK;
}

extern "C" int dge_graph_add_edges(dge_graph* g, const int32_t* src, const int32_t* dst, const double* w, int64_t n) {
    return add_edges_common(g, src, dst, w, n, hipMemcpyHostToDevice);
}
extern "C" int dge_graph_add_edges_device(dge_graph* g, const int32_t* src, const int32_t* dst, const double* w, int64_t n) {
    return add_edges_common(g, src, dst, w, n, hipMemcpyDeviceToDevice);
}

// COO (insertion order) -> CSR with insertion order kept inside every vertex: stable radix sort on src
int dge_graph_ensure_csr(dge_graph* g) {
    if (g->csr_built) return DGE_OK;
    DGE_HIP(hipSetDevice(g->device));
    const int64_t E = g->n_coo;
    const int32_t V = g->max_id + 1;
    int rc;
    free_csr(g);
    if ((rc = dge_dev_alloc(&g->d_row_ptr, (size_t)V + 1))) return rc;
    if ((rc = dge_dev_alloc(&g->d_nbr, (size_t)E))) return rc;
    if ((rc = dge_dev_alloc(&g->d_w, (size_t)E))) return rc;
    if ((rc = dge_dev_alloc(&g->d_outdeg, (size_t)V))) return rc;
    if (E > 0) {
        if (E >= (int64_t)0xFFFFFFFFLL) DGE_FAIL(DGE_ERR_ARG, "edge count %lld exceeds 2^32-1", (long long)E);
        dge_tmp<int32_t> d_keys; dge_tmp<uint32_t> d_idx, d_idx_sorted; dge_tmp<char> d_tmp;
        if ((rc = d_keys.alloc((size_t)E))) return rc;
        if ((rc = d_idx.alloc((size_t)E))) return rc;
        if ((rc = d_idx_sorted.alloc((size_t)E))) return rc;
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(E, 256)), dim3(256), 0, g->stream, d_idx.p, E);
        int end_bit = 1;
        while (end_bit < 31 && (1LL << end_bit) < (int64_t)V) end_bit++;
        size_t tmp_bytes = 0;
        DGE_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, g->d_coo_src, d_keys.p, d_idx.p, d_idx_sorted.p, E, 0, end_bit, g->stream));
        if ((rc = d_tmp.alloc(tmp_bytes))) return rc;
        DGE_HIP(hipcub::DeviceRadixSort::SortPairs((void*)d_tmp.p, tmp_bytes, g->d_coo_src, d_keys.p, d_idx.p, d_idx_sorted.p, E, 0, end_bit, g->stream));
        hipLaunchKernelGGL(k_gather_edges, dim3(grid_for(E, 256)), dim3(256), 0, g->stream, d_idx_sorted.p, g->d_coo_dst, g->d_coo_w, g->d_nbr, g->d_w, E);
        hipLaunchKernelGGL(k_row_ptr, dim3(grid_for((int64_t)V + 1, 256)), dim3(256), 0, g->stream, d_keys.p, E, V, g->d_row_ptr);
        hipLaunchKernelGGL(k_out_degree, dim3(grid_for(V, 256)), dim3(256), 0, g->stream, g->d_row_ptr, g->d_w, V, g->d_outdeg);
        DGE_HIP(hipStreamSynchronize(g->stream));
    } else {
        DGE_HIP(hipMemsetAsync(g->d_row_ptr, 0, ((size_t)V + 1) * sizeof(int64_t), g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
    }
    DGE_HIP(hipGetLastError());
    g->V = V; g->E = E; g->csr_built = true; g->alias_built = false;
    return DGE_OK;
}

extern "C" int dge_graph_set_sources(dge_graph* g, const int32_t* v, int64_t n, int stream_sum) {
    if (!g || n < 0 || (n > 0 && !v)) DGE_FAIL(DGE_ERR_ARG, "dge_graph_set_sources: bad argument");
    int rc = dge_graph_ensure_csr(g);
    if (rc) return rc;
    for (int64_t i = 0; i < n; i++)
        if (v[i] < 0 || v[i] >= g->V) DGE_FAIL(DGE_ERR_RANGE, "dge_graph_set_sources: vertex %d is not in the graph (V=%d)", v[i], g->V);
    free_sources(g);
    if ((rc = dge_dev_alloc(&g->d_srcv, (size_t)n))) return rc;
    if ((rc = dge_dev_alloc(&g->d_src_w, (size_t)n))) return rc;
    dge_tmp<double> d_sum;
    if ((rc = d_sum.alloc(1))) return rc;
    if (n) DGE_HIP(hipMemcpyAsync(g->d_srcv, v, n * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
    launch_sources(g->stream, g->d_srcv, n, g->d_outdeg, g->d_src_w, stream_sum, d_sum.p);
    DGE_HIP(hipMemcpyAsync(&g->src_weight_sum, d_sum.p, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    DGE_HIP(hipStreamSynchronize(g->stream));
    g->S = n;
    g->src_stream_sum = stream_sum ? 1 : 0; g->src_sum_fixed = false;
    g->alias_built = false;
    return DGE_OK;
}

extern "C" int dge_graph_reserve_vertices(dge_graph* g, int32_t n) {
    if (!g || n < 0) DGE_FAIL(DGE_ERR_ARG, "dge_graph_reserve_vertices: bad argument");
    if (n - 1 > g->max_id) {
        if (g->csr_built && g->E != g->n_coo) DGE_FAIL(DGE_ERR_STATE, "dge_graph_reserve_vertices: the store was pruned by keep_top_k");
        DGE_HIP(hipSetDevice(g->device));
        g->max_id = n - 1;
        free_csr(g);
    }
    return DGE_OK;
}

// Vertex.outDegree is a public field of the reference (J/LayeredGraph.java:35; J/SpatialGraph.java:33 assigns it): a host that
// keeps that field hands its values over instead of having the running sums recomputed
extern "C" int dge_graph_set_out_degree(dge_graph* g, const double* out_degree, int32_t n) {
    if (!g || !out_degree || n < 0) DGE_FAIL(DGE_ERR_ARG, "dge_graph_set_out_degree: bad argument");
    int rc = dge_graph_ensure_csr(g);
    if (rc) return rc;
    if (n != g->V) DGE_FAIL(DGE_ERR_ARG, "dge_graph_set_out_degree: %d values for %d vertices", n, g->V);
    if (n) DGE_HIP(hipMemcpyAsync(g->d_outdeg, out_degree, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g->stream));
    g->alias_built = false;
    if (g->S > 0) {              // source weights are outDegree values: refresh them (and their sum, unless the host fixed it)
        dge_tmp<double> d_sum;
        if ((rc = d_sum.alloc(1))) return rc;
        double sum = 0.0;
        launch_sources(g->stream, g->d_srcv, g->S, g->d_outdeg, g->d_src_w, g->src_stream_sum, d_sum.p);
        DGE_HIP(hipMemcpyAsync(&sum, d_sum.p, sizeof(double), hipMemcpyDeviceToHost, g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
        if (!g->src_sum_fixed) g->src_weight_sum = sum;
    }
    DGE_HIP(hipStreamSynchronize(g->stream));
    return DGE_OK;
}

// LayeredGraph.sourceWeightSum is a protected field that subclasses assign (J/SpatialGraph.java:57,83)
extern "C" int dge_graph_set_source_weight_sum(dge_graph* g, double sum) {
    if (!g) DGE_FAIL(DGE_ERR_ARG, "dge_graph_set_source_weight_sum: null graph");
    g->src_weight_sum = sum; g->src_sum_fixed = true; g->alias_built = false;
    return DGE_OK;
}

extern "C" int dge_graph_get_csr(const dge_graph* gc, int64_t* row_ptr, int32_t* nbr, double* weight, double* prob, int32_t* alias,
                                 double* out_degree, int32_t cap_vertices, int64_t cap_edges) {
    dge_graph* g = const_cast<dge_graph*>(gc);
    if (!g) DGE_FAIL(DGE_ERR_ARG, "dge_graph_get_csr: null graph");
    int rc = dge_graph_ensure_csr(g);
    if (rc) return rc;
    if ((row_ptr || out_degree) && cap_vertices < g->V) DGE_FAIL(DGE_ERR_CAP, "dge_graph_get_csr: %d vertices exceed cap %d", g->V, cap_vertices);
    if ((nbr || weight || prob || alias) && cap_edges < g->E) DGE_FAIL(DGE_ERR_CAP, "dge_graph_get_csr: %lld edges exceed cap %lld", (long long)g->E, (long long)cap_edges);
    if ((prob || alias) && !g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_graph_get_csr: alias tables not built");
    DGE_HIP(hipSetDevice(g->device));
    DGE_HIP(hipStreamSynchronize(g->stream));
    if (row_ptr) DGE_HIP(hipMemcpy(row_ptr, g->d_row_ptr, ((size_t)g->V + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (out_degree && g->V) DGE_HIP(hipMemcpy(out_degree, g->d_outdeg, (size_t)g->V * sizeof(double), hipMemcpyDeviceToHost));
    if (g->E) {
        if (nbr) DGE_HIP(hipMemcpy(nbr, g->d_nbr, (size_t)g->E * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (weight) DGE_HIP(hipMemcpy(weight, g->d_w, (size_t)g->E * sizeof(double), hipMemcpyDeviceToHost));
        if (prob) DGE_HIP(hipMemcpy(prob, g->d_prob, (size_t)g->E * sizeof(double), hipMemcpyDeviceToHost));
        if (alias) DGE_HIP(hipMemcpy(alias, g->d_alias, (size_t)g->E * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return DGE_OK;
}

extern "C" int dge_graph_keep_top_k(dge_graph* g, int32_t k) {
    if (!g || k < 0) DGE_FAIL(DGE_ERR_ARG, "dge_graph_keep_top_k: bad argument");
    int rc = dge_graph_ensure_csr(g);
    if (rc) return rc;
    const int32_t V = g->V; const int64_t E = g->E;
    if (V == 0) return DGE_OK;
    dge_tmp<unsigned long long> d_min;
    if ((rc = d_min.alloc(1))) return rc;
    DGE_HIP(hipMemsetAsync(d_min, 0xFF, sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(k_min_degree, dim3(std::min<unsigned>(grid_for(V, 256), 2048u)), dim3(256), 0, g->stream, g->d_row_ptr, V, d_min.p);
    unsigned long long mind = 0;
    DGE_HIP(hipMemcpyAsync(&mind, d_min.p, sizeof(mind), hipMemcpyDeviceToHost, g->stream));
    DGE_HIP(hipStreamSynchronize(g->stream));
    if (mind < (unsigned long long)k)
        DGE_FAIL(DGE_ERR_TOPK, "keepNearestKVertices(%d): a vertex has only %llu out-edges (the reference throws IndexOutOfBoundsException)", k, mind);
    dge_tmp<double> d_ws; dge_tmp<int32_t> d_ns; dge_tmp<char> d_tmp;
    if ((rc = d_ws.alloc((size_t)E))) return rc;
    if ((rc = d_ns.alloc((size_t)E))) return rc;
    size_t tmp_bytes = 0;
    DGE_HIP(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, tmp_bytes, g->d_w, d_ws.p, g->d_nbr, d_ns.p, E, V,
                                                                  g->d_row_ptr, g->d_row_ptr + 1, 0, 64, g->stream));
    if ((rc = d_tmp.alloc(tmp_bytes))) return rc;
    DGE_HIP(hipcub::DeviceSegmentedRadixSort::SortPairsDescending((void*)d_tmp.p, tmp_bytes, g->d_w, d_ws.p, g->d_nbr, d_ns.p, E, V,
                                                                  g->d_row_ptr, g->d_row_ptr + 1, 0, 64, g->stream));
    dge_tmp<int64_t> nrp; dge_tmp<double> nw; dge_tmp<int32_t> nn;
    const int64_t NE = (int64_t)V * k;
    if ((rc = nrp.alloc((size_t)V + 1))) return rc;
    if ((rc = nw.alloc((size_t)NE))) return rc;
    if ((rc = nn.alloc((size_t)NE))) return rc;
    hipLaunchKernelGGL(k_topk_compact, dim3(grid_for((int64_t)V + 1, 256)), dim3(256), 0, g->stream, g->d_row_ptr, d_ws.p, d_ns.p, V, k,
                       nrp.p, nw.p, nn.p, g->d_outdeg);
    DGE_HIP(hipStreamSynchronize(g->stream));
    DGE_HIP(hipGetLastError());
    dge_dev_free(g->d_row_ptr); dge_dev_free(g->d_w); dge_dev_free(g->d_nbr);
    dge_dev_free(g->d_prob); dge_dev_free(g->d_alias); dge_dev_free(g->d_slots);
    g->d_prob = nullptr; g->d_alias = nullptr; g->d_slots = nullptr;
    g->d_row_ptr = nrp.release(); g->d_w = nw.release(); g->d_nbr = nn.release(); g->E = NE;
    g->alias_built = false;
    // source weights depend on outDegree: refresh them if sources were already set
    if (g->S > 0) {
        dge_tmp<double> d_sum;
        if ((rc = d_sum.alloc(1))) return rc;
        launch_sources(g->stream, g->d_srcv, g->S, g->d_outdeg, g->d_src_w, 1, d_sum.p);
        DGE_HIP(hipMemcpyAsync(&g->src_weight_sum, d_sum.p, sizeof(double), hipMemcpyDeviceToHost, g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
        g->src_stream_sum = 1; g->src_sum_fixed = false;
    }
    return DGE_OK;
}

extern "C" int dge_graph_build_alias(dge_graph* g, int exact) {
    if (!g) DGE_FAIL(DGE_ERR_ARG, "dge_graph_build_alias: null graph");
    int rc = dge_graph_ensure_csr(g);
    if (rc) return rc;
    const int32_t V = g->V; const int64_t E = g->E; const int64_t S = g->S;
    dge_dev_free(g->d_prob); dge_dev_free(g->d_alias); dge_dev_free(g->d_slots);
    dge_dev_free(g->d_src_prob); dge_dev_free(g->d_src_alias); dge_dev_free(g->d_src_slots);
    g->d_prob = nullptr; g->d_alias = nullptr; g->d_slots = nullptr;
    g->d_src_prob = nullptr; g->d_src_alias = nullptr; g->d_src_slots = nullptr;
    if ((rc = dge_dev_alloc(&g->d_prob, (size_t)E))) return rc;
    if ((rc = dge_dev_alloc(&g->d_alias, (size_t)E))) return rc;
    if ((rc = dge_dev_alloc(&g->d_slots, (size_t)E))) return rc;
    if ((rc = dge_dev_alloc(&g->d_src_prob, (size_t)S))) return rc;
    if ((rc = dge_dev_alloc(&g->d_src_alias, (size_t)S))) return rc;
    if ((rc = dge_dev_alloc(&g->d_src_slots, (size_t)S))) return rc;
    dge_tmp<uint64_t> d_bs; dge_tmp<int32_t> d_vs;
    if (exact) {
        size_t words = (size_t)std::max<int64_t>(2 * (E / 32 + 6 * (int64_t)V) + 64, 2 * dge_bs_words(S) + 64);
        if ((rc = d_bs.alloc(words))) return rc;
    } else {
        if ((rc = d_vs.alloc((size_t)std::max<int64_t>(E, S) + 1))) return rc;
    }
    if (V > 0) {
        const int64_t hub_min = exact ? 0 : ALIAS_WAVE_MIN;
        hipLaunchKernelGGL(k_alias_vertices, dim3(grid_for(V, 64)), dim3(64), 0, g->stream, V, g->d_row_ptr, g->d_w,
                           g->d_outdeg, g->d_prob, g->d_alias, exact, d_bs.p, d_vs.p, hub_min);
        if (hub_min > 0 && E >= hub_min) {
            const int64_t max_hubs = E / hub_min;
            dge_tmp<int32_t> d_hubs;
            if ((rc = d_hubs.alloc((size_t)max_hubs + 1))) return rc;
            int32_t n_hubs = 0;
            DGE_HIP(hipMemsetAsync(d_hubs.p + max_hubs, 0, sizeof(int32_t), g->stream));
            hipLaunchKernelGGL(k_hub_list, dim3(grid_for(V, 256)), dim3(256), 0, g->stream, V, g->d_row_ptr, hub_min, d_hubs.p, d_hubs.p + max_hubs);
            DGE_HIP(hipMemcpyAsync(&n_hubs, d_hubs.p + max_hubs, sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
            DGE_HIP(hipStreamSynchronize(g->stream));
            if (n_hubs > 0)
                hipLaunchKernelGGL(k_alias_hubs, dim3((unsigned)n_hubs), dim3(64), 0, g->stream, d_hubs.p, g->d_row_ptr, g->d_w, g->d_outdeg, g->d_prob,
                                   g->d_alias, d_vs.p);
            DGE_HIP(hipStreamSynchronize(g->stream));
        }
    }
    if (S > 0)
        hipLaunchKernelGGL(k_alias_sources, dim3(1), dim3(64), 0, g->stream, S, g->d_src_w, g->src_weight_sum,
                           g->d_src_prob, g->d_src_alias, exact, d_bs.p, d_vs.p);
    if (E > 0) {
        // the table an edge slot belongs to: the vertices mark their first slot, a running maximum carries the mark along the row
        dge_tmp<int32_t> d_owner_own; dge_tmp<char> d_tmp;
        int32_t* d_owner = d_vs.p;                                        // (Vose order: the stacks are done with once the stream gets here — E + 1 ints)
        if (!d_owner) { if ((rc = d_owner_own.alloc((size_t)E))) return rc; d_owner = d_owner_own.p; }
        DGE_HIP(hipMemsetAsync(d_owner, 0, (size_t)E * sizeof(int32_t), g->stream));
        hipLaunchKernelGGL(k_owner_mark, dim3(grid_for(V, 256)), dim3(256), 0, g->stream, V, g->d_row_ptr, d_owner);
        for (int64_t e0 = 0; e0 < E; e0 += (1ll << 30)) {                 // (the scan counts in 32 bits; a later piece starts from its predecessor's last mark)
            const int64_t ne = std::min<int64_t>(E - e0, 1ll << 30);
            const int64_t lo = e0 > 0 ? e0 - 1 : 0;
            size_t tb = 0;
            DGE_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, tb, d_owner + lo, d_owner + lo, MaxI32(), (int)(e0 + ne - lo), g->stream));
            if (!d_tmp.p && (rc = d_tmp.alloc(tb + 256))) return rc;
            DGE_HIP(hipcub::DeviceScan::InclusiveScan((void*)d_tmp.p, tb, d_owner + lo, d_owner + lo, MaxI32(), (int)(e0 + ne - lo), g->stream));
        }
        hipLaunchKernelGGL(k_fill_slots, dim3(grid_for(E, 256)), dim3(256), 0, g->stream, E, (const int32_t*)d_owner, g->d_row_ptr, g->d_prob, g->d_alias,
                           g->d_nbr, g->d_slots);
        DGE_HIP(hipStreamSynchronize(g->stream));
    }
    if (S > 0)
        hipLaunchKernelGGL(k_fill_slots, dim3(grid_for(S, 256)), dim3(256), 0, g->stream, S, (const int32_t*)nullptr, g->d_row_ptr, g->d_src_prob,
                           g->d_src_alias, g->d_srcv, g->d_src_slots);
    DGE_HIP(hipStreamSynchronize(g->stream));
    DGE_HIP(hipGetLastError());
    g->alias_built = true;
    return DGE_OK;
}

extern "C" int dge_graph_num_vertices(const dge_graph* g, int32_t* n) {
    if (!g || !n) DGE_FAIL(DGE_ERR_ARG, "dge_graph_num_vertices: bad argument");
    *n = g->csr_built ? g->V : g->max_id + 1;
    return DGE_OK;
}
extern "C" int dge_graph_num_edges(const dge_graph* g, int64_t* n) {
    if (!g || !n) DGE_FAIL(DGE_ERR_ARG, "dge_graph_num_edges: bad argument");
    *n = g->csr_built ? g->E : g->n_coo;
    return DGE_OK;
}

extern "C" int dge_graph_get_alias(const dge_graph* gc, int32_t v, double* prob, int32_t* alias, int32_t* nbr, double* weight,
                                   int32_t cap, int32_t* k, double* out_degree) {
    dge_graph* g = const_cast<dge_graph*>(gc);
    if (!g) DGE_FAIL(DGE_ERR_ARG, "dge_graph_get_alias: null graph");
    int rc = dge_graph_ensure_csr(g);
    if (rc) return rc;
    if (v < 0 || v >= g->V) DGE_FAIL(DGE_ERR_RANGE, "dge_graph_get_alias: vertex %d not in graph (V=%d)", v, g->V);
    DGE_HIP(hipSetDevice(g->device));
    int64_t rp[2];
    DGE_HIP(hipMemcpy(rp, g->d_row_ptr + v, sizeof(rp), hipMemcpyDeviceToHost));
    int64_t d = rp[1] - rp[0];
    if (k) *k = (int32_t)d;
    if (out_degree) DGE_HIP(hipMemcpy(out_degree, g->d_outdeg + v, sizeof(double), hipMemcpyDeviceToHost));
    if (d > cap) {
        if (prob || alias || nbr || weight) DGE_FAIL(DGE_ERR_CAP, "dge_graph_get_alias: degree %lld exceeds cap %d", (long long)d, cap);
        return DGE_OK;
    }
    if (d == 0) return DGE_OK;
    if ((prob || alias) && !g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_graph_get_alias: alias tables not built");
    if (prob) DGE_HIP(hipMemcpy(prob, g->d_prob + rp[0], d * sizeof(double), hipMemcpyDeviceToHost));
    if (alias) DGE_HIP(hipMemcpy(alias, g->d_alias + rp[0], d * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nbr) DGE_HIP(hipMemcpy(nbr, g->d_nbr + rp[0], d * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (weight) DGE_HIP(hipMemcpy(weight, g->d_w + rp[0], d * sizeof(double), hipMemcpyDeviceToHost));
    return DGE_OK;
}

extern "C" int dge_graph_get_source_alias(const dge_graph* g, double* prob, int32_t* alias, int32_t* src, int32_t cap, int32_t* k,
                                          double* weight_sum) {
    if (!g) DGE_FAIL(DGE_ERR_ARG, "dge_graph_get_source_alias: null graph");
    if (k) *k = (int32_t)g->S;
    if (weight_sum) *weight_sum = g->src_weight_sum;
    if (g->S > cap) {
        if (prob || alias || src) DGE_FAIL(DGE_ERR_CAP, "dge_graph_get_source_alias: %lld sources exceed cap %d", (long long)g->S, cap);
        return DGE_OK;
    }
    if (g->S == 0) return DGE_OK;
    DGE_HIP(hipSetDevice(g->device));
    if ((prob || alias) && !g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_graph_get_source_alias: alias tables not built");
    if (prob) DGE_HIP(hipMemcpy(prob, g->d_src_prob, g->S * sizeof(double), hipMemcpyDeviceToHost));
    if (alias) DGE_HIP(hipMemcpy(alias, g->d_src_alias, g->S * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (src) DGE_HIP(hipMemcpy(src, g->d_srcv, g->S * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DGE_OK;
}

extern "C" int dge_graph_sample_next(const dge_graph* g, int32_t v, double x, int32_t* next) {
    if (!g || !next) DGE_FAIL(DGE_ERR_ARG, "dge_graph_sample_next: bad argument");
    if (!g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_graph_sample_next: call dge_graph_build_alias first");
    if (v < 0 || v >= g->V) DGE_FAIL(DGE_ERR_RANGE, "dge_graph_sample_next: vertex %d not in graph (V=%d)", v, g->V);
    if (!(x >= 0.0 && x < 1.0)) DGE_FAIL(DGE_ERR_ARG, "dge_graph_sample_next: x must be in [0,1)");
    DGE_HIP(hipSetDevice(g->device));
    dge_tmp<int32_t> d_out;
    int rc = d_out.alloc(1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sample_next, dim3(1), dim3(64), 0, g->stream, g->d_row_ptr, g->d_slots, v, x, d_out.p);
    DGE_HIP(hipMemcpyAsync(next, d_out.p, sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    DGE_HIP(hipStreamSynchronize(g->stream));
    return DGE_OK;
}

// ------------------------------------------------------------------------------------------ walks API
int dge_launch_walks_strided(const dge_graph* g, hipStream_t stream, int32_t* d_out, int64_t n, int32_t L, int64_t seed,
                             int64_t first_index, int32_t* d_deadend_count) {
    if (n == 0) return DGE_OK;
    size_t lds = (size_t)WALK_BLOCK * (size_t)(L | 1) * sizeof(int32_t);
    if (lds > 64 * 1024) DGE_FAIL(DGE_ERR_ARG, "walk length %d too long (max %d)", L, (int)(64 * 1024 / WALK_BLOCK / 4 - 1));
    hipLaunchKernelGGL(k_walks, dim3(grid_for(n, WALK_BLOCK)), dim3(WALK_BLOCK), lds, stream, g->d_row_ptr, g->d_slots, g->d_src_slots,
                       g->S, d_out, n, L, dge_jr_scramble(seed), first_index * (int64_t)L, (int64_t)L, (const int64_t*)nullptr, (uint8_t*)nullptr, d_deadend_count);
    DGE_HIP(hipGetLastError());
    return DGE_OK;
}

static int sample_walks_impl(const dge_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode, int64_t first_index,
                             int32_t* d_out, int64_t* draws_consumed) {
    if (!g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_sample_walks: call dge_graph_build_alias first (J/LayeredGraph.java:195)");
    if (rng_mode != 0 && rng_mode != 1) DGE_FAIL(DGE_ERR_ARG, "dge_sample_walks: rng_mode must be 0 or 1");
    DGE_HIP(hipSetDevice(g->device));
    if (n_walks == 0) { if (draws_consumed) *draws_consumed = 0; return DGE_OK; }
    dge_tmp<int32_t> d_dead;
    int rc = d_dead.alloc(1);
    if (rc) return rc;
    DGE_HIP(hipMemsetAsync(d_dead.p, 0, sizeof(int32_t), g->stream));
    int64_t walk0 = first_index;
    if (rng_mode == 0) {
        // sequential stream: if no walk dead-ends every walk takes exactly max_len draws and the stream
        // position of walk i is first_index + i*max_len — which is the strided layout shifted by first_index.
        if (first_index % max_len != 0) walk0 = -1;
        else walk0 = first_index / max_len;
    }
    int32_t dead = 1;
    if (walk0 >= 0) {
        rc = dge_launch_walks_strided(g, g->stream, d_out, n_walks, max_len, seed, walk0, d_dead.p);
        if (rc) return rc;
        DGE_HIP(hipMemcpyAsync(&dead, d_dead.p, sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
    }
    if (rng_mode == 1) {
        if (draws_consumed) *draws_consumed = n_walks * (int64_t)max_len;   // stride, not data-dependent
        return DGE_OK;
    }
    if (dead == 0) { if (draws_consumed) *draws_consumed = n_walks * (int64_t)max_len; return DGE_OK; }
    // some walk dead-ended (or the stream offset is unaligned): draw counts are data dependent
    int64_t draws = 0;
    if (g->S == 0) {                  // no source: every walk is empty and consumes nothing
        DGE_HIP(hipMemsetAsync(d_out, 0xFF, (size_t)(n_walks * max_len) * sizeof(int32_t), g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
    } else if (n_walks < 2048) {      // short corpora: one lane chains the walks
        dge_tmp<int64_t> d_draws;
        if ((rc = d_draws.alloc(1))) return rc;
        hipLaunchKernelGGL(k_walks_sequential, dim3(1), dim3(64), 0, g->stream, g->d_row_ptr, g->d_slots, g->d_src_slots, g->S, d_out,
                           n_walks, max_len, dge_jr_scramble(seed), first_index, d_draws.p);
        DGE_HIP(hipMemcpyAsync(&draws, d_draws.p, sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
        DGE_HIP(hipGetLastError());
    } else {                          // parallel resolver (see k_seq_block_exits)
        const int64_t N = n_walks * (int64_t)max_len;
        const int64_t n_blocks = (N + SEQ_BLOCK - 1) / SEQ_BLOCK;
        dge_tmp<uint8_t> d_lens; dge_tmp<int32_t> d_exit, d_hops, d_se; dge_tmp<int64_t> d_si, d_starts, d_draws;
        if ((rc = d_lens.alloc((size_t)N + 256))) return rc;
        if ((rc = d_exit.alloc((size_t)(n_blocks * max_len)))) return rc;
        if ((rc = d_hops.alloc((size_t)(n_blocks * max_len)))) return rc;
        if ((rc = d_se.alloc((size_t)n_blocks))) return rc;
        if ((rc = d_si.alloc((size_t)n_blocks))) return rc;
        if ((rc = d_starts.alloc((size_t)n_walks))) return rc;
        if ((rc = d_draws.alloc(1))) return rc;
        const size_t lds = (size_t)WALK_BLOCK * (size_t)(max_len | 1) * sizeof(int32_t);
        hipLaunchKernelGGL(k_walks, dim3(grid_for(N, WALK_BLOCK)), dim3(WALK_BLOCK), lds, g->stream, g->d_row_ptr, g->d_slots, g->d_src_slots, g->S,
                           (int32_t*)nullptr, N, max_len, dge_jr_scramble(seed), first_index, (int64_t)1, (const int64_t*)nullptr, d_lens.p, (int32_t*)nullptr);
        hipLaunchKernelGGL(k_seq_block_exits, dim3(grid_for(n_blocks * max_len, 256)), dim3(256), 0, g->stream, d_lens.p, N, max_len, n_blocks, d_exit.p, d_hops.p);
        hipLaunchKernelGGL(k_seq_chain, dim3(1), dim3(64), 0, g->stream, d_exit.p, d_hops.p, max_len, n_blocks, d_se.p, d_si.p);
        hipLaunchKernelGGL(k_seq_starts, dim3(grid_for(n_blocks, 256)), dim3(256), 0, g->stream, d_lens.p, N, n_blocks, d_se.p, d_si.p, first_index, n_walks, d_starts.p, d_draws.p);
        hipLaunchKernelGGL(k_walks, dim3(grid_for(n_walks, WALK_BLOCK)), dim3(WALK_BLOCK), lds, g->stream, g->d_row_ptr, g->d_slots, g->d_src_slots, g->S,
                           d_out, n_walks, max_len, dge_jr_scramble(seed), (int64_t)0, (int64_t)0, (const int64_t*)d_starts.p, (uint8_t*)nullptr, (int32_t*)nullptr);
        DGE_HIP(hipMemcpyAsync(&draws, d_draws.p, sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
        DGE_HIP(hipStreamSynchronize(g->stream));
        DGE_HIP(hipGetLastError());
    }
    if (draws_consumed) *draws_consumed = draws;
    return DGE_OK;
}

extern "C" int dge_sample_walks_device(const dge_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode,
                                       int64_t first_index, dge_walks** out, int64_t* draws_consumed) {
    if (!g || !out || n_walks < 0 || max_len <= 0 || first_index < 0) DGE_FAIL(DGE_ERR_ARG, "dge_sample_walks_device: bad argument");
    *out = nullptr;
    DGE_HIP(hipSetDevice(g->device));
    dge_walks* w = new dge_walks();
    w->device = g->device; w->n = n_walks; w->L = max_len; w->gen = dge_next_generation();
    int rc = dge_dev_alloc(&w->d, (size_t)(n_walks * max_len));
    if (rc) { delete w; return rc; }
    rc = sample_walks_impl(g, n_walks, max_len, seed, rng_mode, first_index, w->d, draws_consumed);
    if (rc) { dge_walks_free(w); return rc; }
    *out = w;
    return DGE_OK;
}

extern "C" int dge_sample_walks(const dge_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode, int64_t first_index,
                                int32_t* out, int64_t* draws_consumed) {
    if (!out && n_walks > 0) DGE_FAIL(DGE_ERR_ARG, "dge_sample_walks: null output");
    dge_walks* w = nullptr;
    int rc = dge_sample_walks_device(g, n_walks, max_len, seed, rng_mode, first_index, &w, draws_consumed);
    if (rc) return rc;
    rc = dge_walks_to_host(w, out, n_walks * (int64_t)max_len);
    dge_walks_free(w);
    return rc;
}

extern "C" int dge_sample_walks_into(const dge_graph* g, dge_walks* w, int64_t row0, int64_t n_walks, int64_t seed, int64_t first_index) {
    if (!g || !w || row0 < 0 || n_walks < 0 || row0 + n_walks > w->n || first_index < 0) DGE_FAIL(DGE_ERR_ARG, "dge_sample_walks_into: bad argument");
    if (!g->alias_built) DGE_FAIL(DGE_ERR_STATE, "dge_sample_walks_into: alias tables not built");
    if (g->device != w->device) DGE_FAIL(DGE_ERR_ARG, "dge_sample_walks_into: graph and corpus live on different devices");
    DGE_HIP(hipSetDevice(g->device));
    w->gen = dge_next_generation();
    int rc = dge_launch_walks_strided(g, g->stream, w->d + row0 * w->L, n_walks, w->L, seed, first_index, nullptr);
    if (rc) return rc;
    DGE_HIP(hipStreamSynchronize(g->stream));
    return DGE_OK;
}

extern "C" int dge_walks_from_host(int device, const int32_t* walks, int64_t n_walks, int32_t max_len, dge_walks** out) {
    if (!out || n_walks < 0 || max_len <= 0 || (n_walks > 0 && !walks)) DGE_FAIL(DGE_ERR_ARG, "dge_walks_from_host: bad argument");
    *out = nullptr;
    int rc = dge_require_device(device);
    if (rc) return rc;
    dge_walks* w = new dge_walks();
    w->device = device; w->n = n_walks; w->L = max_len; w->gen = dge_next_generation();
    rc = dge_dev_alloc(&w->d, (size_t)(n_walks * max_len));
    if (rc) { delete w; return rc; }
    if (n_walks && hipMemcpy(w->d, walks, (size_t)(n_walks * max_len) * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        dge_walks_free(w);
        DGE_FAIL(DGE_ERR_DEVICE, "dge_walks_from_host: copy to the device failed");
    }
    *out = w;
    return DGE_OK;
}

extern "C" int dge_walks_to_host(const dge_walks* w, int32_t* out, int64_t cap_elems) {
    if (!w || (!out && w->n > 0)) DGE_FAIL(DGE_ERR_ARG, "dge_walks_to_host: bad argument");
    if (cap_elems < w->n * w->L) DGE_FAIL(DGE_ERR_CAP, "dge_walks_to_host: buffer holds %lld of %lld tokens", (long long)cap_elems, (long long)(w->n * w->L));
    DGE_HIP(hipSetDevice(w->device));
    if (w->n) DGE_HIP(hipMemcpy(out, w->d, (size_t)(w->n * w->L) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DGE_OK;
}

extern "C" int dge_walks_info(const dge_walks* w, int64_t* n_walks, int32_t* max_len, const int32_t** d_ptr) {
    if (!w) DGE_FAIL(DGE_ERR_ARG, "dge_walks_info: null corpus");
    if (n_walks) *n_walks = w->n;
    if (max_len) *max_len = w->L;
    if (d_ptr) *d_ptr = w->d;
    return DGE_OK;
}

extern "C" int dge_walks_add_position_prefix(dge_walks* w, int32_t region_count) {
    if (!w || region_count <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_walks_add_position_prefix: bad argument");
    if ((int64_t)region_count * w->L > 0x7fffffffLL) DGE_FAIL(DGE_ERR_RANGE, "position prefix overflows int32 ids");
    DGE_HIP(hipSetDevice(w->device));
    w->gen = dge_next_generation();
    if (w->n) hipLaunchKernelGGL(k_position_prefix, dim3(grid_for(w->n * w->L, 256)), dim3(256), 0, 0, w->d, w->n, w->L, region_count);
    DGE_HIP(hipDeviceSynchronize());
    return DGE_OK;
}

extern "C" void dge_walks_free(dge_walks* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    dge_dev_free(w->d);
    delete w;
}
