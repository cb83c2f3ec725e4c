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
    const uint64_t MiB = 1ull << 20, total = 1024 * MiB;
    float* sink; CK(hipMalloc(&sink, 16));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int round = 0; round < 2; round++)
    for (int how = 0; how < 6; how++) {
        char* buf = nullptr; std::vector<hipMemGenericAllocationHandle_t> hs;
        const char* names[6] = {"hipMalloc", "contiguous", "2 MiB chunks, compact", "2 MiB chunks out of 48 GiB", "64 MiB chunks, compact", "64 MiB chunks out of 48 GiB"};
        if (how == 0) CK(hipMalloc((void**)&buf, total));
        else if (how == 1) { if (hipExtMallocWithFlags((void**)&buf, total, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); continue; } }
        else {
            const size_t chunk = how <= 3 ? 2 * MiB : 64 * MiB; const bool scatter = how == 3 || how == 5;
            const size_t need = total / chunk, pool_n = scatter ? need * 48 : need;
            std::vector<hipMemGenericAllocationHandle_t> pool(pool_n);
            for (size_t i = 0; i < pool_n; i++) CK(hipMemCreate(&pool[i], chunk, &prop, 0));
            uint64_t rs = 777 + how; std::vector<size_t> idx(pool_n); for (size_t i = 0; i < pool_n; i++) idx[i] = i;
            if (scatter) for (size_t k = pool_n; k > 1; k--) { rs = rs * 6364136223846793005ull + 1442695040888963407ull; std::swap(idx[k - 1], idx[(rs >> 33) % k]); }
            for (size_t i = 0; i < need; i++) hs.push_back(pool[idx[i]]);
            for (size_t i = need; i < pool_n; i++) (void)hipMemRelease(pool[idx[i]]);
            CK(hipMemAddressReserve((void**)&buf, total, 1ull << 30, nullptr, 0));
            for (size_t i = 0; i < need; i++) CK(hipMemMap(buf + i * chunk, chunk, 0, hs[i], 0));
            CK(hipMemSetAccess(buf, total, &acc, 1));
        }
        CK(hipMemset(buf, 0, total)); CK(hipDeviceSynchronize());
        printf("round %d  %-28s rows read %5.0f GB/s   read + written back %5.0f GB/s\n", round, names[how], run(buf, total, false, sink), run(buf, total, true, sink));
        if (how >= 2) { (void)hipMemUnmap(buf, total); for (auto& h : hs) (void)hipMemRelease(h); (void)hipMemAddressFree(buf, total); } else (void)hipFree(buf);
    }
    return 0;
}
