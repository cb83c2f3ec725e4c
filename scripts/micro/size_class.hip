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
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    // which allocation sizes ever land in fast memory?  16 allocations of every size and kind, all held; rate of random rows read and written back
    for (uint64_t mib : {256, 384, 448, 480, 487, 496, 504, 510, 511, 512, 513, 520, 544, 640, 768, 1024}) {
        for (int kind = 0; kind < 2; kind++) {
            const uint64_t total = mib * MiB;
            std::vector<char*> bufs; std::vector<hipMemGenericAllocationHandle_t> hs;
            printf("%4llu MiB %s:", (unsigned long long)mib, kind ? "virtual-memory" : "hipMalloc     ");
            for (int k = 0; k < 16; k++) {
                char* buf = nullptr;
                if (kind == 0) CK(hipMalloc((void**)&buf, total));
                else {
                    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, total, &prop, 0)); hs.push_back(h);
                    CK(hipMemAddressReserve((void**)&buf, total, 2 * MiB, nullptr, 0));
                    CK(hipMemMap(buf, total, 0, h, 0)); CK(hipMemSetAccess(buf, total, &acc, 1));
                }
                CK(hipMemset(buf, 0, total)); CK(hipDeviceSynchronize());
                printf(" %4.0f", run(buf, total, true, sink));
                bufs.push_back(buf);
            }
            printf("\n");
            for (size_t k = 0; k < bufs.size(); k++) {
                if (kind == 0) (void)hipFree(bufs[k]);
                else { (void)hipMemUnmap(bufs[k], total); (void)hipMemRelease(hs[k]); (void)hipMemAddressFree(bufs[k], total); }
            }
        }
    }
    return 0;
}
