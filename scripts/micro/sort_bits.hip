// radix sort of (int32 key, uint64 value) items as update_policy 8 sorts them: rocprim's default digit width (8 bits) against wider digits
// (fewer passes over the items).  hipcc --offload-arch=gfx950 -O3 -o /tmp/sort_bits scripts/micro/sort_bits.hip && /tmp/sort_bits
#include <hip/hip_runtime.h>

#include <cstring>
#include <cstdint>
#include <cstdio>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(int32_t* k, uint64_t* v, int64_t n, int bits) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = (uint64_t)i * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    k[i] = (int32_t)(s & ((1u << bits) - 1)); v[i] = (uint64_t)i;
}
__global__ void k_check(const int32_t* k, const uint64_t* v, int64_t n, int* bad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    if (k[i] > k[i + 1] || (k[i] == k[i + 1] && v[i] > v[i + 1])) atomicAdd(bad, 1);      // sorted and stable
}

template <class Config>
static int run(const char* name, int64_t n, int bits, int32_t* k0, int32_t* k1, uint64_t* v0, uint64_t* v1, int* d_bad) {
    size_t b = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, b, k0, k1, v0, v1, (size_t)n, 0, bits, 0)));
    void* tmp; CK(hipMalloc(&tmp, b));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, 0));
        CK((rocprim::radix_sort_pairs<Config>(tmp, b, k0, k1, v0, v1, (size_t)n, 0, bits, 0)));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k1, v1, n, d_bad);
    int bad = 0; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("n %10lld bits %2d %-26s %7.3f ms  %6.2f Gitems/s  %s\n", (long long)n, bits, name, best, n / best / 1e6, bad ? "NOT SORTED/STABLE" : "ok");
    CK(hipFree(tmp));
    return 0;
}

int main() {
    using namespace rocprim;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<1024, 8>, 9, block_radix_rank_algorithm::match>> C9;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<1024, 8>, 10, block_radix_rank_algorithm::match>> C10;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<512, 12>, 9, block_radix_rank_algorithm::match>> C9b;
    // bigger tiles (gfx950 has 160 KB of LDS a workgroup may use): more items per digit and block, longer runs in the scatter
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<512, 16>, 9, block_radix_rank_algorithm::match>> C9_512x16;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<512, 20>, 9, block_radix_rank_algorithm::match>> C9_512x20;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<1024, 12>, 9, block_radix_rank_algorithm::match>> C9_1024x12;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<1024, 16>, 9, block_radix_rank_algorithm::match>> C9_1024x16;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<256, 24>, 9, block_radix_rank_algorithm::match>> C9_256x24;
    const int64_t sizes[] = {12800000, 48000000};
    for (int64_t n : sizes) {
        int32_t *k0, *k1; uint64_t *v0, *v1; int* d_bad;
        CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8)); CK(hipMalloc(&d_bad, 4));
        for (int bits : {17, 18}) {
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k0, v0, n, bits);
            if (run<default_config>("default (8-bit digits)", n, bits, k0, k1, v0, v1, d_bad)) return 1;
            if (run<C9b>("9-bit, 512x12", n, bits, k0, k1, v0, v1, d_bad)) return 1;
            if (run<C9_512x16>("9-bit, 512x16", n, bits, k0, k1, v0, v1, d_bad)) return 1;
            if (run<C9_512x20>("9-bit, 512x20", n, bits, k0, k1, v0, v1, d_bad)) return 1;
            if (run<C9_1024x12>("9-bit, 1024x12", n, bits, k0, k1, v0, v1, d_bad)) return 1;
            if (run<C9_1024x16>("9-bit, 1024x16", n, bits, k0, k1, v0, v1, d_bad)) return 1;
            if (run<C9_256x24>("9-bit, 256x24", n, bits, k0, k1, v0, v1, d_bad)) return 1;
        }
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(d_bad);
    }
    return 0;
}
