// Microbenchmark for the reference's own row width (D = 20 on a 6 408-row vocabulary, J/DeepWalk.java:62-66): how fast do row updates by float
// atomics complete on a table of that size when a row update is (a) two instructions of 16 + 4 contiguous lanes (the trainer's atomics layout: a
// 16-lane group, lane j holds floats j and 16 + j), (b) one instruction of twenty contiguous lanes (32-lane groups), (c) two instructions of ten lanes
// (float2 a lane, 16-lane groups)?
// Each group first reads the row (agent-scope load, as the trainer does), then adds to it.
// hipcc --offload-arch=gfx950 -O3 small_row_atomics.hip -o small_row_atomics && ./small_row_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void add_agent(float* p, float v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// FORM 0: 16-lane groups, floats j and 16 + j a lane; FORM 1: 32-lane groups, a float a lane; FORM 2: 16-lane groups, float2 a lane
template <int FORM>
__global__ void __launch_bounds__(256) k_rows(float* table, int n_rows, int stride, int D, int iters, int rows_per_it, int skew, float* sink) {
    constexpr int G = FORM == 1 ? 32 : 16;
    const int lane = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        for (int r = 0; r < rows_per_it; r++) {
            uint64_t h = mix64((uint64_t)(group * 1000003 + it * 131 + r));
            int row = (int)(h % (uint64_t)n_rows);
            if (skew) { const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0); row = (int)((double)n_rows * u * u * u); }   // cubic: the first 10 % of the rows take 46 %
            float* p = table + (int64_t)row * stride;
            if (FORM == 0) {
                float x0 = ld_agent(p + lane), x1 = lane + 16 < D ? ld_agent(p + 16 + lane) : 0.f;
                acc += x0 + x1;
                atomicAdd(p + lane, 1e-9f); if (lane + 16 < D) atomicAdd(p + 16 + lane, 1e-9f);
            } else if (FORM == 1) {
                if (lane < D) { acc += ld_agent(p + lane); atomicAdd(p + lane, 1e-9f); }
            } else {
                if (lane * 2 < D) { acc += ld_agent(p + lane * 2) + ld_agent(p + lane * 2 + 1); atomicAdd(p + lane * 2, 1e-9f); atomicAdd(p + lane * 2 + 1, 1e-9f); }
            }
        }
        // the dependent part of a pair: the next pair starts when this one's reads are back
        acc = __shfl_xor(acc, 1) + acc;
    }
    if (acc == 12345.f) sink[0] = acc;
}
int main() {
    const int n_rows = 6408, stride = 64, D = 20;
    float* d; hipMalloc(&d, (size_t)n_rows * stride * sizeof(float)); hipMemset(d, 0, (size_t)n_rows * stride * sizeof(float));
    float* sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int skew = 0; skew < 2; skew++)
    for (int groups : {3204, 6408, 9612, 16384, 32768})
    for (int form = 0; form < 3; form++) {
        const int G = form == 1 ? 32 : 16, iters = 2000, rows_per_it = 7;
        const unsigned blocks = (unsigned)(((int64_t)groups * G + 255) / 256);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (form == 0) hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, d, n_rows, stride, D, iters, rows_per_it, skew, sink);
            if (form == 1) hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(256), 0, 0, d, n_rows, stride, D, iters, rows_per_it, skew, sink);
            if (form == 2) hipLaunchKernelGGL(k_rows<2>, dim3(blocks), dim3(256), 0, 0, d, n_rows, stride, D, iters, rows_per_it, skew, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double groups_run = (double)blocks * 256 / G;
        printf("%s rows, %5d groups, %s: %8.2f ms  %.3e row updates/s = %.3e pairs/s of 7 rows\n", skew ? "skewed " : "uniform", groups,
               form == 0 ? "16 + 4 lanes (trainer)" : form == 1 ? "1 x 20 lanes (float) " : "2 x 10 lanes (float2)", best,
               groups_run * iters * rows_per_it / best * 1e3, groups_run * iters / best * 1e3);
    }
    return 0;
}
