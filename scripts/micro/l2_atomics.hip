// Microbenchmark: float atomics performed in the XCD's own L2 (workgroup scope: no sc1 bit) against the memory-side form
// (agent scope, sc1) that HIP's atomicAdd emits.  Rows are owned by XCDs (row % 8 == HW_REG_XCC_ID of the adding workgroup), so an
// L2-local atomic is never raced from another XCD; the kernel boundary writes the dirty lines back.  Checks conservation (every
// element must equal the number of additions its row received, counted with integer atomics) and prints the byte rates.
// hipcc --offload-arch=gfx950 -O3 l2_atomics.hip -o l2_atomics && ./l2_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <cmath>
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7; }

// MODE 0: agent-scope atomics (memory side)   1: workgroup-scope atomics (L2)   2: sc1 load of the row + L2 atomics (the SGNS shape)
// 3: sc1 load + plain store (lossy reference point for the data path)
template <int MODE>
__global__ void __launch_bounds__(256) k_add(float* table, unsigned* counts, int64_t rows_per_xcd, int row_floats, int iters, int* census) {
    const int lane = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&census[x], 1);
    float sink = 0.f;
    for (int it = 0; it < iters; it++) {
        const int64_t r = (int64_t)(mix64((uint64_t)(group * 1000003 + it)) % (uint64_t)rows_per_xcd);
        const int64_t row = r * 8 + x;                                  // a row of THIS XCD's partition
        float* p = table + row * row_floats + lane;
        if (MODE >= 2) for (int c = 0; c < row_floats / 16; c++) sink += __builtin_nontemporal_load(p + c * 16) * 0.f + __hip_atomic_load(p + c * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int c = 0; c < row_floats / 16; c++) {
            if (MODE == 0) atomicAdd(p + c * 16, 1.0f);
            else if (MODE == 3) p[c * 16] = sink * 0.f + 1.0f;
            else (void)__hip_atomic_fetch_add(p + c * 16, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (lane == 0 && counts) atomicAdd(&counts[row], 1u);
    }
    if (sink == 12345.f) table[0] = sink;
}

int main() {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { int64_t rows; int row_floats; const char* name; } cfgs[] = {{100000, 64, "100k x 256 B (25.6 MB: cache resident)"}, {1 << 20, 128, "1M x 512 B (512 MB)"}};
    for (auto& c : cfgs) {
        const int64_t rows_per_xcd = c.rows / 8, n_rows = rows_per_xcd * 8;
        float* d; unsigned* cnt; int* census;
        hipMalloc(&d, n_rows * c.row_floats * sizeof(float)); hipMalloc(&cnt, n_rows * sizeof(unsigned)); hipMalloc(&census, 8 * sizeof(int));
        for (int mode : {0, 1, 2, 3}) {
            for (int with_counts = 1; with_counts >= 0; with_counts--) {
                hipMemset(d, 0, n_rows * c.row_floats * sizeof(float)); hipMemset(cnt, 0, n_rows * sizeof(unsigned)); hipMemset(census, 0, 8 * sizeof(int));
                const int iters = 200; const int64_t groups = 16384; const unsigned blocks = (unsigned)(groups * 16 / 256);
                hipDeviceSynchronize();
                hipEventRecord(a);
                unsigned* cp = with_counts ? cnt : nullptr;
                if (mode == 0) hipLaunchKernelGGL(k_add<0>, dim3(blocks), dim3(256), 0, 0, d, cp, rows_per_xcd, c.row_floats, iters, census);
                if (mode == 1) hipLaunchKernelGGL(k_add<1>, dim3(blocks), dim3(256), 0, 0, d, cp, rows_per_xcd, c.row_floats, iters, census);
                if (mode == 2) hipLaunchKernelGGL(k_add<2>, dim3(blocks), dim3(256), 0, 0, d, cp, rows_per_xcd, c.row_floats, iters, census);
                if (mode == 3) hipLaunchKernelGGL(k_add<3>, dim3(blocks), dim3(256), 0, 0, d, cp, rows_per_xcd, c.row_floats, iters, census);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double bytes = (double)groups * iters * c.row_floats * 4;
                if (with_counts) {      // conservation: read back through the host (every XCD's L2 was written back at the kernel boundary)
                    std::vector<float> h((size_t)n_rows * c.row_floats); std::vector<unsigned> hc((size_t)n_rows); int cs[8];
                    hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost); hipMemcpy(hc.data(), cnt, hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
                    hipMemcpy(cs, census, sizeof(cs), hipMemcpyDeviceToHost);
                    double worst = 0; int64_t bad = 0;
                    for (int64_t r = 0; r < n_rows; r++) for (int e = 0; e < c.row_floats; e++) { double dlt = fabs((double)h[(size_t)r * c.row_floats + e] - (double)hc[(size_t)r]); if (dlt > 0) bad++; if (dlt > worst) worst = dlt; }
                    printf("%-40s mode %d: conservation worst |err| %.0f, %lld bad elements; blocks per XCD %d %d %d %d %d %d %d %d\n", c.name, mode, worst, (long long)bad, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], cs[6], cs[7]);
                } else
                    printf("%-40s mode %d: %8.3f ms  %7.1f GB/s of row bytes  %.2e rows/s\n", c.name, mode, ms, bytes / ms / 1e6, groups * (double)iters / ms * 1e3);
            }
        }
        hipFree(d); hipFree(cnt); hipFree(census);
    }
    // box calibration: float4 copy of 1 GiB
    { float4 *x, *y; size_t n = (1u << 30) / 16; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMemset(x, 1, n * 16);
      for (int r = 0; r < 3; r++) { hipEventRecord(a); hipMemcpyAsync(y, x, n * 16, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); printf("copy 1 GiB: %.3f ms = %.0f GB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e6); } }
    return 0;
}
