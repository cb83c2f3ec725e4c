// knn.hip — cosine k-nearest-neighbour lists on the GPU (gfx950): the quality metric of the reference's evaluation
// scripts, P/embeddingEvaluation_tract.py:169-196 (pairwiseEstimator: per region the other regions sorted by
// scipy-cosine distance, NaN -> 2).  A "next" row of SURVEY.md §8f, not part of the training hot path; it serves as the
// statistical parity check between training schedules at sizes where the Python loop (O(n^2) scipy calls) is hopeless.
//
// One workgroup = 64 query rows.  The strip of similarities S[64 x n] = Xq . X^T is produced tile by tile (64 columns)
// with the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (4 waves = 2x2 sub-tiles of 32x32; the query fragment of
// a wave stays in registers for the whole strip, the column tile is staged in LDS), and is consumed immediately: all 256
// threads test the tile's 64 x 64 similarities against each row's current k-th best (a threshold that only tightens, so
// nothing that belongs in the list is missed), and one lane per query row then inserts the few columns that passed, in
// column order, into that row's k best (distance ascending, index ascending among equals) in LDS.
// Nothing of size n^2 ever reaches HBM.
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "dge_internal.h"

typedef float v16f __attribute__((ext_vector_type(16)));
#define KNN_MAX_K 64
#define KNN_MAX_D 256

__global__ void k_row_norms(const float* __restrict__ x, int n, int D, float* __restrict__ inv_norm) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s = 0.0;
    for (int j = 0; j < D; j++) { double v = x[(size_t)r * D + j]; s += v * v; }
    inv_norm[r] = s > 0.0 ? (float)(1.0 / sqrt(s)) : 0.0f;       // 0 marks a zero vector: cosine undefined -> distance 2
}

template <int DH>   // DH = D_padded / 2 = MFMA steps per tile
__global__ void __launch_bounds__(256)
k_knn_strip(const float* __restrict__ x, const float* __restrict__ inv_norm, int n, int D, int k, int32_t* __restrict__ out_idx,
            float* __restrict__ out_dist) {
    extern __shared__ float lds[];
    const int Dp = DH * 2;
    float* Bs = lds;                              // [64][Dp + 1]   column tile, row-major per column vector
    float* Ss = Bs + 64 * (Dp + 1);               // [64][65]       similarities of the tile
    float* Ld = Ss + 64 * 65;                     // [64][k]        running best distances (pitch k: two workgroups fit a CU and overlap their phases)
    int32_t* Li = (int32_t*)(Ld + 64 * k);        // [64][k]        their indices
    uint32_t* Ms = (uint32_t*)(Li + 64 * k);      // [64][4]     per row and 16-column quarter: columns that beat the row's threshold
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
    const int q0 = blockIdx.x * 64;

    // query fragment: lane holds A[row = lane%32][kk = 2*j + lane/32] for j = 0..DH-1, scaled to unit norm
    float a[DH];
    {
        const int row = q0 + wr * 32 + (lane & 31);
        const float sc = row < n ? inv_norm[row] : 0.0f;
#pragma unroll
        for (int j = 0; j < DH; j++) {
            const int kk = 2 * j + (lane >> 5);
            a[j] = (row < n && kk < D) ? x[(size_t)row * D + kk] * sc : 0.0f;
        }
    }
    for (int i = t; i < 64 * k; i += 256) { Ld[i] = 3.0f; Li[i] = -1; }     // 3 > any cosine distance

    for (int c0 = 0; c0 < n; c0 += 64) {
        __syncthreads();                                          // previous tile fully consumed
        for (int i = t; i < 64 * Dp; i += 256) {                   // stage the normalised column vectors
            const int c = i / Dp, kk = i - c * Dp, col = c0 + c;
            Bs[c * (Dp + 1) + kk] = (col < n && kk < D) ? x[(size_t)col * D + kk] * inv_norm[col] : 0.0f;
        }
        __syncthreads();
        v16f acc = {0};
        const float* bcol = Bs + (wc * 32 + (lane & 31)) * (Dp + 1) + (lane >> 5);
#pragma unroll
        for (int j = 0; j < DH; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bcol[2 * j], acc, 0, 0, 0);
        // accumulator register v of a lane is S[8*(v/4) + 4*(lane/32) + v%4][lane%32] of the wave's 32x32 sub-tile
#pragma unroll
        for (int v = 0; v < 16; v++)
            Ss[(wr * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3)) * 65 + wc * 32 + (lane & 31)] = acc[v];
        __syncthreads();
        {   // filter: thread (row r, quarter qd) tests 16 columns against the row's k-th best as it stands before this tile
            const int r = t & 63, qd = t >> 6, q = q0 + r;
            uint32_t m = 0;
            if (q < n) {
                const bool qzero = inv_norm[q] == 0.0f;
                const float thr = Ld[r * k + k - 1];
                for (int cc = 0; cc < 16; cc++) {
                    const int c = qd * 16 + cc, col = c0 + c;
                    if (col >= n || col == q) continue;
                    const float d = (qzero || inv_norm[col] == 0.0f) ? 2.0f : 1.0f - Ss[r * 65 + c];
                    if (d < thr) m |= 1u << cc;
                }
            }
            Ms[r * 4 + qd] = m;
        }
        __syncthreads();
        if (t < 64 && q0 + t < n) {                               // one lane per query row inserts what passed, in column order
            const int q = q0 + t;
            const bool qzero = inv_norm[q] == 0.0f;
            float* ld = Ld + t * k; int32_t* li = Li + t * k;
            uint64_t mask = (uint64_t)Ms[t * 4] | ((uint64_t)Ms[t * 4 + 1] << 16) | ((uint64_t)Ms[t * 4 + 2] << 32) | ((uint64_t)Ms[t * 4 + 3] << 48);
            while (mask) {
                const int c = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int col = c0 + c;
                float d = (qzero || inv_norm[col] == 0.0f) ? 2.0f : 1.0f - Ss[t * 65 + c];
                if (!(d < ld[k - 1])) continue;                   // ties keep the earlier (smaller) index: stable order
                int p = k - 1;
                while (p > 0 && d < ld[p - 1]) { ld[p] = ld[p - 1]; li[p] = li[p - 1]; p--; }
                ld[p] = d; li[p] = col;
            }
        }
    }
    __syncthreads();
    for (int i = t; i < 64 * k; i += 256) {
        const int r = i / k, j = i - r * k;
        if (q0 + r < n) { out_idx[(size_t)(q0 + r) * k + j] = Li[r * k + j]; out_dist[(size_t)(q0 + r) * k + j] = Ld[r * k + j]; }
    }
}

template <int DH>
static void launch_knn(const float* x, const float* inv, int n, int D, int k, int32_t* oi, float* od, hipStream_t st) {
    const size_t lds = (size_t)(64 * (2 * DH + 1) + 64 * 65 + 64 * k * 2 + 64 * 4) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_knn_strip<DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_knn_strip<DH>), dim3((n + 63) / 64), dim3(256), lds, st, x, inv, n, D, k, oi, od);
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
    dge_tmp<float> d_x, d_inv, d_od; dge_tmp<int32_t> d_oi;
    if ((rc = d_x.alloc((size_t)n * D))) return rc;
    if ((rc = d_inv.alloc((size_t)n))) return rc;
    if ((rc = d_od.alloc((size_t)n * k))) return rc;
    if ((rc = d_oi.alloc((size_t)n * k))) return rc;
    DGE_HIP(hipMemcpy(d_x.p, features, (size_t)n * D * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_row_norms, dim3((n + 255) / 256), dim3(256), 0, 0, d_x.p, n, D, d_inv.p);
    hipEvent_t e0, e1;
    DGE_HIP(hipEventCreate(&e0)); DGE_HIP(hipEventCreate(&e1));
    DGE_HIP(hipEventRecord(e0, 0));
    const int dh = (D + 1) / 2;
    if (dh <= 16) launch_knn<16>(d_x.p, d_inv.p, n, D, k, d_oi.p, d_od.p, 0);
    else if (dh <= 32) launch_knn<32>(d_x.p, d_inv.p, n, D, k, d_oi.p, d_od.p, 0);
    else if (dh <= 64) launch_knn<64>(d_x.p, d_inv.p, n, D, k, d_oi.p, d_od.p, 0);
    else launch_knn<128>(d_x.p, d_inv.p, n, D, k, d_oi.p, d_od.p, 0);
    DGE_HIP(hipEventRecord(e1, 0));
    DGE_HIP(hipGetLastError());
    DGE_HIP(hipDeviceSynchronize());
    float ms = 0.f;
    DGE_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms_kernel) *ms_kernel = ms;
    DGE_HIP(hipMemcpy(out_idx, d_oi.p, (size_t)n * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    DGE_HIP(hipMemcpy(out_dist, d_od.p, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost));
    return DGE_OK;
}
