// Microbenchmark: memory-side float atomic throughput as a function of how many CONTIGUOUS bytes one instruction covers per
// group of lanes (16 lanes = 64 B, 32 lanes = 128 B, 64 lanes = 256 B), rows picked at random from a table much larger than the L2s.
// hipcc --offload-arch=gfx950 -O3 atomic_width.hip -o atomic_width && ./atomic_width
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
template <int G>   // lanes per group
__global__ void __launch_bounds__(256) k_atomic(float* table, int64_t n_rows, int row_floats, int iters, int chunks) {
    const int lane = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    for (int it = 0; it < iters; it++) {
        const int64_t row = (int64_t)(mix64((uint64_t)(group * 1000003 + it)) % (uint64_t)n_rows);
        float* p = table + row * row_floats + lane;
        // one row update = row_floats floats = row_floats/G instructions of G contiguous lanes
        for (int c = 0; c < chunks; c++) atomicAdd(p + c * G, 1.0f);
    }
}
int main() {
    const int64_t n_rows = 1 << 20; const int row_floats = 128;     // 512-B rows, 512 MB table
    float* d; hipMalloc(&d, n_rows * row_floats * sizeof(float)); hipMemset(d, 0, n_rows * row_floats * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++)
    for (int G : {16, 32, 64}) {
        const int chunks = row_floats / G, iters = 400;
        const int64_t groups = 16384LL * 16 / G * 4;                // same number of lanes in flight for every G
        const unsigned blocks = (unsigned)(groups * G / 256);
        hipEventRecord(a);
        if (G == 16) hipLaunchKernelGGL(k_atomic<16>, dim3(blocks), dim3(256), 0, 0, d, n_rows, row_floats, iters, chunks);
        if (G == 32) hipLaunchKernelGGL(k_atomic<32>, dim3(blocks), dim3(256), 0, 0, d, n_rows, row_floats, iters, chunks);
        if (G == 64) hipLaunchKernelGGL(k_atomic<64>, dim3(blocks), dim3(256), 0, 0, d, n_rows, row_floats, iters, chunks);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)groups * iters * row_floats * 4;
        printf("lanes/group %2d (%3d contiguous B per instruction): %8.2f ms, %6.1f GB/s of atomic bytes, %.2e rows/s\n", G, G * 4, ms, bytes / ms / 1e6, groups * (double)iters / ms * 1e3);
    }
    return 0;
}
