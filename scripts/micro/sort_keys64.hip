// What would an 8-byte item buy the owner-computes schedule's sorts?  (int32 key, uint64 value) pairs — 12 bytes an item, as shipped — against ONE packed
// uint64 whose top bits are the key, sorted on that bit range only (stable: the low bits ride along).  9-bit digits, 17 / 18 / 20-bit keys.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/sort_keys64 scripts/micro/sort_keys64.hip && /tmp/sort_keys64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <rocprim/device/device_radix_sort.hpp>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(int32_t* k, uint64_t* v, uint64_t* p, int64_t n, int bits, int shift) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = (uint64_t)i * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    const uint64_t key = s & ((1ull << bits) - 1);
    k[i] = (int32_t)key; v[i] = (uint64_t)i; p[i] = (key << shift) | ((uint64_t)i & ((1ull << shift) - 1));
}
int main() {
    using namespace rocprim;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<512, 12>, 9, block_radix_rank_algorithm::match>> C9;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<512, 16>, 9, block_radix_rank_algorithm::match>> C9w;
    typedef radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<512, 16>, 10, block_radix_rank_algorithm::match>> C10w;
    const int shift = 40;
    for (int64_t n : {12800000ll, 48000000ll}) {
        int32_t *k0, *k1; uint64_t *v0, *v1, *p0, *p1;
        CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8)); CK(hipMalloc(&p0, n * 8)); CK(hipMalloc(&p1, n * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int bits : {17, 18, 20}) {
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k0, v0, p0, n, bits, shift);
            size_t b1 = 0, b2 = 0, b3 = 0, b4 = 0;
            CK((radix_sort_pairs<C9>(nullptr, b1, k0, k1, v0, v1, (size_t)n, 0, bits, 0)));
            CK((radix_sort_keys<C9>(nullptr, b2, p0, p1, (size_t)n, shift, shift + bits, 0)));
            CK((radix_sort_keys<C9w>(nullptr, b3, p0, p1, (size_t)n, shift, shift + bits, 0)));
            CK((radix_sort_keys<C10w>(nullptr, b4, p0, p1, (size_t)n, shift, shift + bits, 0)));
            size_t b = b1 > b2 ? b1 : b2; b = b > b3 ? b : b3; b = b > b4 ? b : b4;
            void* tmp; CK(hipMalloc(&tmp, b));
            float t[4] = {1e9f, 1e9f, 1e9f, 1e9f};
            for (int r = 0; r < 5; r++)
                for (int w = 0; w < 4; w++) {
                    size_t bb = b;
                    CK(hipEventRecord(e0, 0));
                    if (w == 0) CK((radix_sort_pairs<C9>(tmp, bb, k0, k1, v0, v1, (size_t)n, 0, bits, 0)));
                    if (w == 1) CK((radix_sort_keys<C9>(tmp, bb, p0, p1, (size_t)n, shift, shift + bits, 0)));
                    if (w == 2) CK((radix_sort_keys<C9w>(tmp, bb, p0, p1, (size_t)n, shift, shift + bits, 0)));
                    if (w == 3) CK((radix_sort_keys<C10w>(tmp, bb, p0, p1, (size_t)n, shift, shift + bits, 0)));
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t[w]) t[w] = ms;
                }
            printf("n %9lld  %2d-bit keys:  pairs 4+8 B (shipped) %6.3f ms | packed uint64, 9-bit digits 512x12 %6.3f ms (%.2fx) | 512x16 %6.3f ms (%.2fx) | 10-bit digits 512x16 %6.3f ms (%.2fx)\n",
                   (long long)n, bits, t[0], t[1], t[0] / t[1], t[2], t[0] / t[2], t[3], t[0] / t[3]);
            CK(hipFree(tmp));
        }
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(p0); hipFree(p1);
    }
    return 0;
}
