// knn.hip — cosine k-nearest-neighbour lists on the GPU (gfx950): the quality metric of the reference's evaluation
// scripts, P/embeddingEvaluation_tract.py:169-196 (pairwiseEstimator: per region the other regions sorted by
// scipy-cosine distance, NaN -> 2).  A "next" row of SURVEY.md §8f, not part of the training hot path; it serves as the
// statistical parity check between training schedules at sizes where the Python loop (O(n^2) scipy calls) is hopeless.
//
// One workgroup = 64 query rows (4 waves x 16 rows, the query fragments stay in registers for the whole strip).  The strip of
// similarities S[64 x n] = Xq . X^T is produced 64 columns at a time with the exact-f32 matrix instruction v_mfma_f32_16x16x4_f32
// (a wave: its 16 rows x the tile's 4 x 16 columns, four accumulators); the column tile comes from LDS, the NEXT tile's 16-byte global
// loads are issued before the matrix loop and land in LDS after it.  The similarities never leave the accumulator registers: every
// lane tests its 16 values against the current k-th best of their rows (a threshold that only tightens, so nothing that belongs in
// a list is missed) and the few that pass are inserted into the row's k best in LDS under a per-row lock, ordered by (distance,
// index) — the order of insertion does not matter.  Nothing of size n^2 ever reaches HBM.
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "dge_internal.h"

#define KNN_MAX_K 64
#define KNN_MAX_D 256

__global__ void k_row_norms(const float* __restrict__ x, int n, int D, float* __restrict__ inv_norm) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s = 0.0;
    for (int j = 0; j < D; j++) { double v = x[(size_t)r * D + j]; s += v * v; }
    inv_norm[r] = s > 0.0 ? (float)(1.0 / sqrt(s)) : 0.0f;       // 0 marks a zero vector: cosine undefined -> distance 2
}

// rows scaled to unit norm and zero-padded to Dp floats (a zero vector stays zero: inv_norm 0 marks it)
__global__ void k_normalise(const float* __restrict__ x, const float* __restrict__ inv_norm, int n, int D, int Dp, float* __restrict__ xn) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * Dp) return;
    const int r = (int)(i / Dp), c = (int)(i % Dp);
    xn[i] = c < D ? x[(size_t)r * D + c] * inv_norm[r] : 0.0f;
}

// (d, col) < (d2, col2): distance ascending, smaller index first among equals
__device__ __forceinline__ bool knn_less(float d, int col, float d2, int col2) { return d < d2 || (d == d2 && (unsigned)col < (unsigned)col2); }

// 16 query rows per wave, not 32 (v_mfma_f32_32x32x2_f32, round 1): a wave is bound to one of the device's 1 024 SIMDs, so 32-row waves put
// ceil(n / 32 / 1024) whole wave strips on the busiest SIMD — 2 at n = 41 667 where the mean is 1.27 (measured 50.7 TFLOP/s against 86.9 at
// n = 65 536, where every SIMD has 2) — and 16-row waves ceil(n / 16 / 1024) half-size ones (3 at n = 41 667: 74.5 TFLOP/s); the query fragment
// is half as many registers (three workgroups a compute unit instead of two), the columns are streamed from L2 / the Infinity Cache twice as often.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int DH>   // DH = D_padded / 2
__global__ void __launch_bounds__(256, DH >= 128 ? 2 : 3)      // (D = 256: the column tile is 66 KB, two workgroups a compute unit anyway)
k_knn_strip(const float* __restrict__ xn, const float* __restrict__ inv_norm, int n, int k, int32_t* __restrict__ out_idx,
              float* __restrict__ out_dist) {
    extern __shared__ float lds[];
    constexpr int Dp = DH * 2, P = Dp + 4;        // pitch of a column in LDS: 16-byte rows, and 4*col + kk is a different bank for every lane
    constexpr int NPF = Dp / 16;                  // float4 loads per thread and tile: 64 columns x Dp floats / 256 threads
    constexpr int NS = Dp / 4;                    // MFMA steps per tile and column block
    float* Bs = lds;                              // [64][P]        column tile
    float* Ld = Bs + 64 * P;                      // [64][k]        running best distances
    int32_t* Li = (int32_t*)(Ld + 64 * k);        // [64][k]        their indices (-1: empty, sorts last)
    int* Lk = (int*)(Li + 64 * k);                // [64]           per-row locks
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q0 = blockIdx.x * 64;
    const int ce = n;

    // query fragment: lane holds A[row = lane%16][kk = 4*j + lane/16] for j = 0..NS-1
    float a[NS];
    {
        const int row = q0 + wave * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NS; j++) a[j] = row < n ? xn[(size_t)row * Dp + 4 * j + (lane >> 4)] : 0.0f;
    }
    for (int i = t; i < 64 * k; i += 256) { Ld[i] = 3.0f; Li[i] = -1; }      // 3 > any cosine distance
    if (t < 64) Lk[t] = 0;

    float4 pf[NPF];
    auto fetch = [&](int c0) {                    // tile c0 .. c0+63: thread takes float4 number t + 256*i
#pragma unroll
        for (int i = 0; i < NPF; i++) {
            const int f = t + 256 * i, c = f / (Dp / 4), kq = f % (Dp / 4), col = c0 + c;
            pf[i] = col < ce ? *(const float4*)(xn + (size_t)col * Dp + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; i++) {
            const int f = t + 256 * i, c = f / (Dp / 4), kq = f % (Dp / 4);
            *(float4*)(Bs + c * P + 4 * kq) = pf[i];
        }
    };
    fetch(0);
    __syncthreads();
    stash();
    __syncthreads();

    for (int c0 = 0; c0 < ce; c0 += 64) {
        if (c0 + 64 < ce) fetch(c0 + 64);                         // in flight during the matrix loop
        v4f acc[4] = {{0}, {0}, {0}, {0}};
        const float* b = Bs + (lane & 15) * P + (lane >> 4);      // B[kk = 4*j + lane/16][col = 16*cb + lane%16]
#pragma unroll
        for (int j = 0; j < NS; j++) {
#pragma unroll
            for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[cb * 16 * P + 4 * j], acc[cb], 0, 0, 0);
            if ((j & 7) == 7) asm volatile("" ::: "memory");      // keeps the scheduler from hoisting every fragment read at once
        }
        // accumulator register v of column block cb is S[4*(lane/16) + v][16*cb + lane%16].  Common path: 16 compares against the rows' k-th
        // best -> a bit mask (bit 4*cb + v); nothing else is touched.
        uint32_t hits = 0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int rl = wave * 16 + 4 * (lane >> 4) + v, q = q0 + rl;
            const float thr_d = Ld[rl * k + k - 1]; const int thr_i = Li[rl * k + k - 1];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) {
                const int col = c0 + 16 * cb + (lane & 15);
                if (q < n && col < ce && col != q && knn_less(1.0f - acc[cb][v], col, thr_d, thr_i)) hits |= 1u << (4 * cb + v);
            }
        }
        // Rare path (after the lists have warmed up: a handful per tile): one candidate per lane and trip, inserted under its row's lock
        while (__any(hits != 0)) {
            bool pending = hits != 0;
            const int bit = pending ? __builtin_ctz(hits) : 0;
            hits &= hits - 1;
            const int v = bit & 3, cb = bit >> 2;
            float sv = 0.f;
#pragma unroll
            for (int x = 0; x < 16; x++) if (x == bit) sv = acc[x >> 2][x & 3];
            const int rl = wave * 16 + 4 * (lane >> 4) + v, q = q0 + rl;
            const int col = c0 + 16 * cb + (lane & 15);
            float d = 1.0f - sv;
            if (pending && (inv_norm[q] == 0.0f || inv_norm[col] == 0.0f)) d = 2.0f;      // cosine undefined -> distance 2
            while (__any(pending)) {
                if (pending && atomicCAS(&Lk[rl], 0, 1) == 0) {
                    float* ld = Ld + rl * k; int32_t* li = Li + rl * k;
                    if (knn_less(d, col, ld[k - 1], li[k - 1])) {
                        int pos = k - 1;
                        while (pos > 0 && knn_less(d, col, ld[pos - 1], li[pos - 1])) { ld[pos] = ld[pos - 1]; li[pos] = li[pos - 1]; pos--; }
                        ld[pos] = d; li[pos] = col;
                    }
                    __threadfence_block();
                    atomicExch(&Lk[rl], 0);
                    pending = false;
                }
            }
        }
        __syncthreads();                                          // every wave has finished with this tile
        if (c0 + 64 < ce) stash();
        __syncthreads();
    }
    for (int i = t; i < 64 * k; i += 256) {
        const int r = i / k, j = i - r * k;
        if (q0 + r < n) { out_idx[(size_t)(q0 + r) * k + j] = Li[r * k + j]; out_dist[(size_t)(q0 + r) * k + j] = Ld[r * k + j]; }
    }
}

// (Cutting a strip into column segments for more work units than the 326 row blocks of a 41 667-row slice was tried and is slower, 18 against
// 11.5 ms: every segment starts with empty lists, and a row's insertions — ~k ln(columns / k) per scan, each under the row's lock — multiply.
// So was giving the waves of a 64-row workgroup one half of every column tile each: one accumulator per wave, 41.7 against 50.7 TFLOP/s.)
template <int DH>
static int launch_knn(const float* xn, const float* inv, int n, int k, int32_t* oi, float* od, hipStream_t st) {
    const size_t lds = (size_t)(64 * (2 * DH + 4) + 64 * k * 2 + 64) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    (void)hipFuncSetAttribute((const void*)k_knn_strip<DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_knn_strip<DH>), dim3((n + 63) / 64), dim3(256), lds, st, xn, inv, n, k, oi, od);
    return 0;
}

// device form: x [n x D] in device memory -> lists in device memory
static int knn_device(const float* d_x, int n, int D, int k, float* d_inv, float* d_xn, int32_t* d_oi, float* d_od, double* ms_kernel) {
    const int dh = (D + 1) / 2 <= 16 ? 16 : ((D + 1) / 2 <= 32 ? 32 : ((D + 1) / 2 <= 64 ? 64 : 128));
    const int Dp = 2 * dh;
    hipLaunchKernelGGL(k_row_norms, dim3((n + 255) / 256), dim3(256), 0, 0, d_x, n, D, d_inv);
    hipLaunchKernelGGL(k_normalise, dim3((unsigned)(((size_t)n * Dp + 255) / 256)), dim3(256), 0, 0, d_x, d_inv, n, D, Dp, d_xn);
    hipEvent_t e0, e1;
    DGE_HIP(hipEventCreate(&e0)); DGE_HIP(hipEventCreate(&e1));
    DGE_HIP(hipEventRecord(e0, 0));
    int rc;
    if (dh == 16) rc = launch_knn<16>(d_xn, d_inv, n, k, d_oi, d_od, 0);
    else if (dh == 32) rc = launch_knn<32>(d_xn, d_inv, n, k, d_oi, d_od, 0);
    else if (dh == 64) rc = launch_knn<64>(d_xn, d_inv, n, k, d_oi, d_od, 0);
    else rc = launch_knn<128>(d_xn, d_inv, n, k, d_oi, d_od, 0);
    DGE_HIP(hipEventRecord(e1, 0));
    if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); DGE_FAIL(DGE_ERR_ARG, "dge_knn_cosine: k = %d with dim = %d needs more than the 160 KB of LDS", k, D); }
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipDeviceSynchronize());
    float ms = 0.f;
    DGE_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms_kernel) *ms_kernel = ms;
    return DGE_OK;
}

// features: host float32 [n x D]; out_idx / out_dist: host [n x k].  Slots beyond n-1 neighbours hold -1 / 3.0.
// ms_kernel (optional): HIP-event time of the strip kernel.
extern "C" int dge_knn_cosine(int device, const float* features, int32_t n, int32_t D, int32_t k, int32_t* out_idx, float* out_dist,
                              double* ms_kernel) {
    if (!features || !out_idx || !out_dist || n <= 0 || D <= 0 || k <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_knn_cosine: bad argument");
    if (k > KNN_MAX_K) DGE_FAIL(DGE_ERR_ARG, "dge_knn_cosine: k = %d exceeds %d", k, KNN_MAX_K);
    if (D > KNN_MAX_D) DGE_FAIL(DGE_ERR_ARG, "dge_knn_cosine: dim = %d exceeds %d", D, KNN_MAX_D);
    int rc = dge_require_device(device);
    if (rc) return rc;
    dge_tmp<float> d_x, d_inv, d_od, d_xn; dge_tmp<int32_t> d_oi;
    if ((rc = d_x.alloc((size_t)n * D))) return rc;
    if ((rc = d_xn.alloc((size_t)n * 256))) return rc;
    if ((rc = d_inv.alloc((size_t)n))) return rc;
    if ((rc = d_od.alloc((size_t)n * k))) return rc;
    if ((rc = d_oi.alloc((size_t)n * k))) return rc;
    DGE_HIP(hipMemcpy(d_x.p, features, (size_t)n * D * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = knn_device(d_x.p, n, D, k, d_inv.p, d_xn.p, d_oi.p, d_od.p, ms_kernel))) return rc;
    DGE_HIP(hipMemcpy(out_idx, d_oi.p, (size_t)n * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    DGE_HIP(hipMemcpy(out_dist, d_od.p, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost));
    return DGE_OK;
}

// nDCG@k of the reference's evaluation (P/embeddingEvaluation_tract.py:249-260) on the device: relevance of neighbour j of region r =
// 1 - gnd_dist[r][j] (cosine distance in the GROUND features, a zero vector at distance 2), DCG = sum_i relv_i / log2(i + 1) over the k
// nearest neighbours of r in the ESTIMATED features, normalised by the DCG of the ground truth's own k nearest; mean over the regions.
__global__ void k_ndcg(const float* __restrict__ gn, const float* __restrict__ ginv, int Dp, int n, int k, const int32_t* __restrict__ est_idx,
                       const float* __restrict__ gnd_dist, double* __restrict__ ratio) {
    const int r = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    double dcg = 0.0, dmax = 0.0;
    for (int i = 0; i < k; i++) {
        const int j = est_idx[(size_t)r * k + i];
        if (j < 0) break;
        float acc = 0.f;
        for (int c = lane; c < Dp; c += 64) acc = fmaf(gn[(size_t)r * Dp + c], gn[(size_t)j * Dp + c], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        const double dist = (ginv[r] == 0.0f || ginv[j] == 0.0f) ? 2.0 : 1.0 - (double)acc;
        dcg += (1.0 - dist) / log2((double)(i + 2));
        dmax += (1.0 - (double)gnd_dist[(size_t)r * k + i]) / log2((double)(i + 2));
    }
    if (lane == 0) ratio[r] = dmax != 0.0 ? dcg / dmax : 0.0;
}

extern "C" int dge_ndcg_at_k(int device, const float* features, int32_t dim, const float* gnd_features, int32_t gnd_dim, int32_t n, int32_t k,
                             double* ndcg, double* ms_kernels) {
    if (!features || !gnd_features || !ndcg || n <= 1 || dim <= 0 || gnd_dim <= 0 || k <= 0) DGE_FAIL(DGE_ERR_ARG, "dge_ndcg_at_k: bad argument");
    if (k > KNN_MAX_K || k > n - 1) DGE_FAIL(DGE_ERR_ARG, "dge_ndcg_at_k: k = %d must be <= %d and < n", k, KNN_MAX_K);
    if (dim > KNN_MAX_D || gnd_dim > KNN_MAX_D) DGE_FAIL(DGE_ERR_ARG, "dge_ndcg_at_k: dim exceeds %d", KNN_MAX_D);
    int rc = dge_require_device(device);
    if (rc) return rc;
    dge_tmp<float> d_x, d_g, d_inv, d_ginv, d_xn, d_gn, d_od, d_god; dge_tmp<int32_t> d_oi, d_goi; dge_tmp<double> d_ratio;
    if ((rc = d_x.alloc((size_t)n * dim)) || (rc = d_g.alloc((size_t)n * gnd_dim)) || (rc = d_inv.alloc(n)) || (rc = d_ginv.alloc(n)) ||
        (rc = d_xn.alloc((size_t)n * 256)) || (rc = d_gn.alloc((size_t)n * 256)) || (rc = d_od.alloc((size_t)n * k)) || (rc = d_god.alloc((size_t)n * k)) ||
        (rc = d_oi.alloc((size_t)n * k)) || (rc = d_goi.alloc((size_t)n * k)) || (rc = d_ratio.alloc(n))) return rc;
    DGE_HIP(hipMemcpy(d_x.p, features, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice));
    DGE_HIP(hipMemcpy(d_g.p, gnd_features, (size_t)n * gnd_dim * sizeof(float), hipMemcpyHostToDevice));
    double ms1 = 0, ms2 = 0;
    if ((rc = knn_device(d_x.p, n, dim, k, d_inv.p, d_xn.p, d_oi.p, d_od.p, &ms1))) return rc;
    if ((rc = knn_device(d_g.p, n, gnd_dim, k, d_ginv.p, d_gn.p, d_goi.p, d_god.p, &ms2))) return rc;
    const int gdh = (gnd_dim + 1) / 2 <= 16 ? 16 : ((gnd_dim + 1) / 2 <= 32 ? 32 : ((gnd_dim + 1) / 2 <= 64 ? 64 : 128));
    hipLaunchKernelGGL(k_ndcg, dim3((n + 3) / 4), dim3(256), 0, 0, d_gn.p, d_ginv.p, 2 * gdh, n, k, d_oi.p, d_god.p, d_ratio.p);
    DGE_HIP(hipGetLastError());
    std::vector<double> ratio((size_t)n);
    DGE_HIP(hipMemcpy(ratio.data(), d_ratio.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    double s = 0.0;
    for (int r = 0; r < n; r++) s += ratio[(size_t)r];
    *ndcg = s / (double)n;
    if (ms_kernels) *ms_kernels = ms1 + ms2;
    return DGE_OK;
}
