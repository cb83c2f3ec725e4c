// Does it matter that a block of the N-rank schedule reads the rows of ONE partition of a table laid out row-major (partition g = rows g, g + N, g + 2N, ...:
// 512-byte rows 4 KiB apart at N = 8) instead of a compact array of those rows?  Random 512-byte rows (16 lanes x 2 x 16 B, plain cached loads as the owner-computes
// phases issue them) are read from (a) a compact array of R rows, (b) rows g + N k of an N-times larger array, for working sets that fit the Infinity Cache and
// that do not.  Channels interleave every 256 bytes: if the interleave is plain address bits, the strided rows of one partition live on 1/8 of the channels.
// hipcc --offload-arch=gfx950 -O3 strided_rows.hip -o strided_rows && ./strided_rows
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorName(e_), __LINE__); return 1; } } while (0)

template <int WRITE>
__global__ void __launch_bounds__(256) k_rows(char* base, uint64_t n_rows, uint64_t stride_rows, uint64_t first, int row_bytes, int iters, float* sink) {
    const int lane = threadIdx.x & 15;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = 0x9E3779B97F4A7C15ull * (group + 1);
    float acc = 0.f;
    for (int i = 0; i < iters; i += 4) {
        v4u v[4][2]; char* p[4];
#pragma unroll
        for (int z = 0; z < 4; z++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t r = (s >> 20) % n_rows;
            p[z] = base + (first + r * stride_rows) * (uint64_t)row_bytes + lane * 16;
            v[z][0] = *(v4u*)p[z];
            v[z][1] = row_bytes > 256 ? *(v4u*)(p[z] + 256) : v[z][0];
        }
#pragma unroll
        for (int z = 0; z < 4; z++) {
            acc += __uint_as_float(v[z][0].x ^ v[z][1].y);
            if (WRITE) { v[z][0].x += 1u; *(v4u*)p[z] = v[z][0]; if (row_bytes > 256) *(v4u*)(p[z] + 256) = v[z][1]; }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static double run(char* buf, uint64_t n_rows, uint64_t stride_rows, uint64_t first, int row_bytes, bool write, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 256; const uint64_t groups = 65536; const unsigned blocks = (unsigned)(groups * 16 / 256);
    double best = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        if (write) hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, stride_rows, first, row_bytes, iters, sink);
        else hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, stride_rows, first, row_bytes, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        best = std::max(best, (double)groups * iters * row_bytes * (write ? 2 : 1) / (ms * 1e-3) / 1e9);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint64_t MiB = 1ull << 20;
    float* sink; CK(hipMalloc(&sink, 16));
    char* buf; CK(hipMalloc(&buf, 4096 * MiB)); CK(hipMemset(buf, 1, 4096 * MiB));
    printf("random rows read (GB/s) / read + written back (GB/s, both directions counted): working set of R rows, compact against every N-th row of an N-fold array\n");
    for (int row_bytes : {512, 256, 1024})
        for (uint64_t ws_mib : {16, 64, 128, 512}) {
            const uint64_t R = ws_mib * MiB / row_bytes;
            printf("rows of %4d B, working set %4llu MiB:", row_bytes, (unsigned long long)ws_mib);
            for (uint64_t N : {1, 2, 4, 8}) {
                if (R * N * row_bytes > 4096 * MiB) continue;
                const double r = run(buf, R, N, N > 1 ? 3 % N : 0, row_bytes, false, sink), w = run(buf, R, N, N > 1 ? 3 % N : 0, row_bytes, true, sink);
                printf("  N=%llu %6.0f / %6.0f", (unsigned long long)N, r, w);
            }
            printf("\n");
        }
    return 0;
}
