// Microbenchmark behind profiles/r03_placement.txt: what does the PHYSICAL placement of an array do to random-row traffic?
// A 4 GiB buffer is obtained three ways — hipMalloc, hipExtMallocWithFlags(hipDeviceMallocContiguous), and 2 MiB chunks of the
// virtual-memory API mapped in shuffled order — and random 512-byte rows (16 lanes x 2 x 16 B, agent-scope loads as the trainer's) are
// read (mode r) or read and written back write-through (mode w) inside windows of the buffer:
//   one window of S bytes at offset O            -> is the memory behind a window served by all channels, whatever S and O?
//   two windows of 256 MiB at offsets O1 and O2  -> do two arrays interfere depending on their distance (a power of two or not)?
// hipcc --offload-arch=gfx950 -O3 window_bw.hip -o window_bw && ./window_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorName(e_), __LINE__); return 1; } } while (0)

template <int WRITE>
__global__ void __launch_bounds__(256) k_rows(char* base, uint64_t off1, uint64_t off2, uint64_t rows_per_window, int iters, float* sink) {
    const int lane = threadIdx.x & 15;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = 0x9E3779B97F4A7C15ull * (group + 1);
    float acc = 0.f;
    for (int i = 0; i < iters; i += 8) {
        v4u v[8][2]; char* p[8];
#pragma unroll
        for (int z = 0; z < 8; z++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t r = (s >> 20) % rows_per_window;
            p[z] = base + ((z & 1) ? off2 : off1) + r * 512 + lane * 16;
            v[z][0] = __builtin_nontemporal_load((v4u*)p[z]) ;
            v[z][1] = __builtin_nontemporal_load((v4u*)(p[z] + 256));
        }
#pragma unroll
        for (int z = 0; z < 8; z++) {
            acc += __uint_as_float(v[z][0].x ^ v[z][1].y);
            if (WRITE) { v[z][0].x += 1u; v[z][1].y += 1u; __builtin_nontemporal_store(v[z][0], (v4u*)p[z]); __builtin_nontemporal_store(v[z][1], (v4u*)(p[z] + 256)); }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static double run(char* buf, uint64_t o1, uint64_t o2, uint64_t window, bool write, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 256; const uint64_t groups = 65536; const unsigned blocks = (unsigned)(groups * 16 / 256);
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (write) hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(256), 0, 0, buf, o1, o2, window / 512, iters, sink);
        else hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, buf, o1, o2, window / 512, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        best = std::max(best, (double)groups * iters * 512.0 * (write ? 2 : 1) / (ms * 1e-3) / 1e9);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint64_t MiB = 1ull << 20, total = 4096 * MiB;
    float* sink; CK(hipMalloc(&sink, 16));
    for (int how = 0; how < 5; how++) {
        char* buf = nullptr; std::vector<hipMemGenericAllocationHandle_t> hs;
        const char* name = how == 0 ? "hipMalloc" : (how == 1 ? "contiguous" : (how == 2 ? "2 MiB chunks, shuffled" : (how == 3 ? "2 MiB chunks, in order" : "64 MiB chunks, shuffled")));
        printf("allocating: %s\n", name);
        if (how == 0) CK(hipMalloc((void**)&buf, total));
        else if (how == 1) { if (hipExtMallocWithFlags((void**)&buf, total, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous: refused\n"); (void)hipGetLastError(); continue; } }
        else {
            hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            printf("allocation granularity %zu\n", gran);
            gran = how == 4 ? 64 * MiB : 2 * MiB;
            const size_t n = total / gran; hs.resize(n);
            for (size_t i = 0; i < n; i++) CK(hipMemCreate(&hs[i], gran, &prop, 0));
            uint64_t rs = 12345; if (how != 3) for (size_t k = n; k > 1; k--) { rs = rs * 6364136223846793005ull + 1442695040888963407ull; std::swap(hs[k - 1], hs[(rs >> 33) % k]); }
            CK(hipMemAddressReserve((void**)&buf, total, 1ull << 30, nullptr, 0));
            for (size_t i = 0; i < n; i++) CK(hipMemMap(buf + i * gran, gran, 0, hs[i], 0));
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(buf, total, &acc, 1));
        }
        CK(hipMemset(buf, 0, total)); CK(hipDeviceSynchronize());
        printf("== %s (virtual address %p)\n", name, (void*)buf);
        for (int w = 0; w < 2; w++) {
            printf("  %s, one window: ", w ? "read+write" : "read");
            for (uint64_t S : {64 * MiB, 512 * MiB, 2048 * MiB, 4096 * MiB}) printf(" %4llu MiB: %5.0f", (unsigned long long)(S / MiB), run(buf, 0, 0, S, w, sink));
            printf("  | 512 MiB at 1 GiB %5.0f, at 2.5 GiB %5.0f, at 3 GiB + 34 MiB %5.0f GB/s\n", run(buf, 1024 * MiB, 1024 * MiB, 512 * MiB, w, sink),
                   run(buf, 2560 * MiB, 2560 * MiB, 512 * MiB, w, sink), run(buf, 3106 * MiB, 3106 * MiB, 512 * MiB, w, sink));
            printf("  %s, two 512 MiB windows, second at +", w ? "read+write" : "read");
            for (uint64_t d : {512 * MiB, 514 * MiB, 1024 * MiB, 1026 * MiB, 1090 * MiB, 2048 * MiB, 2050 * MiB, 3000 * MiB})
                printf(" %llu MiB: %5.0f", (unsigned long long)(d / MiB), run(buf, 0, d, 512 * MiB, w, sink));
            printf(" GB/s\n");
        }
        if (how >= 2) { hipMemUnmap(buf, total); for (auto& h : hs) hipMemRelease(h); hipMemAddressFree(buf, total); } else hipFree(buf);
    }
    return 0;
}
