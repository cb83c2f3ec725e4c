// (row_offset.hip: the row kernels of scatter_bw.hip / size_class.hip, another main)
// Microbenchmark (profiles/r03_placement.txt): does PHYSICAL SCATTER of an array decide what random-row traffic on it reaches?
// A 1 GiB buffer built six ways — hipMalloc; hipDeviceMallocContiguous; virtual-memory chunks of 2 MiB created one after the other (compact);
// the same number of 2 MiB chunks picked at random from 48 GiB worth of chunks (the rest released: scattered over 48 GiB of physical memory);
// 64 MiB chunks compact and picked from 48 GiB — and random 512-byte rows (agent-scope loads, write-through stores, as the trainer's) read, or
// read and written back, over the whole buffer.
// hipcc --offload-arch=gfx950 -O3 scatter_bw.hip -o scatter_bw && ./scatter_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorName(e_), __LINE__); return 1; } } while (0)

template <int WRITE>
__global__ void __launch_bounds__(256) k_rows(char* base, uint64_t rows, int iters, float* sink) {
    const int lane = threadIdx.x & 15;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = 0x9E3779B97F4A7C15ull * (group + 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(uint32_t)(rows * 512), 0x00020000);
    float acc = 0.f;
    for (int i = 0; i < iters; i += 8) {
        v4u v[8][2]; uint32_t off[8];
#pragma unroll
        for (int z = 0; z < 8; z++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            off[z] = (uint32_t)((s >> 20) % rows) * 512u + (uint32_t)lane * 16u;
            v[z][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[z], 0, 16);
            v[z][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[z] + 256u), 0, 16);
        }
#pragma unroll
        for (int z = 0; z < 8; z++) {
            acc += __uint_as_float(v[z][0].x ^ v[z][1].y);
            if (WRITE) { v[z][0].x += 1u; __builtin_amdgcn_raw_buffer_store_b128(v[z][0], rs, (int)off[z], 0, 16); __builtin_amdgcn_raw_buffer_store_b128(v[z][1], rs, (int)(off[z] + 256u), 0, 16); }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
static double run(char* buf, uint64_t bytes, bool write, float* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 256; const uint64_t groups = 65536; const unsigned blocks = (unsigned)(groups * 16 / 256);
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        if (write) hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 512, iters, sink);
        else hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 512, iters, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        best = std::max(best, (double)groups * iters * 512.0 * (write ? 2 : 1) / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint64_t MiB = 1ull << 20;
    float* sink; CK(hipMalloc(&sink, 16));
    // does the rate of an allocation (its "class") depend on where inside it the rows begin?  12 hipMalloc allocations of 489 MiB + 64 KiB, all held;
    // random 512-byte rows read and written back over 488 MiB starting at byte offset 0, 256, 512, 768, 1024, 2048, 4096, 65536
    const uint64_t total = 488 * MiB;
    std::vector<char*> bufs;
    for (int k = 0; k < 12; k++) {
        char* buf = nullptr;
        CK(hipMalloc((void**)&buf, total + 2097152 + 4096));
        CK(hipMemset(buf, 0, total + 2097152 + 4096)); CK(hipDeviceSynchronize());
        printf("allocation %2d:", k);
        for (uint64_t off : {0, 256, 0, 65536, 0, 2097152, 0}) printf("  +%llu: %4.0f", (unsigned long long)off, run(buf + off, total, true, sink));
        printf("\n");
        bufs.push_back(buf);
    }
    for (char* b : bufs) (void)hipFree(b);
    return 0;
}
