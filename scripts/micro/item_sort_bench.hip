// The owner-computes schedule's own item sort (embedding_amd/csrc/item_sort.h) against rocPRIM's radix_sort_keys on the same inputs: identical output (both are stable sorts
// of 64-bit words on a bit range) on every size / key width / shift tried — edge sizes around the tile, 1 .. 31 key bits —, and the time of both on the sizes the schedule sorts.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/item_sort_bench.hip -o scripts/micro/item_sort_bench.bin && ./scripts/micro/item_sort_bench.bin
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <stdarg.h>
#include "item_sort.h"
std::atomic<int64_t> g_dge_host_syncs{0};
void dge_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); printf("\n"); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(uint64_t* p, int64_t n, int bits, int shift, int skew) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    uint64_t key = s & ((1ull << bits) - 1);
    if (skew && (s >> 40) % 3 == 0) key &= 7;            // a third of the items on eight keys: long runs
    p[i] = (key << shift) | (s & ((1ull << shift) - 1)) | ((s >> 7) << (shift + bits));      // (bits above the key range must not matter)
}
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<512, 16>, 9, rocprim::block_radix_rank_algorithm::match>> SortWide;
static hipError_t ref_sort(void* tmp, size_t& b, const uint64_t* in, uint64_t* out, int64_t n, int shift, int bits) {
    if (bits > 16 && bits <= 18) return rocprim::radix_sort_keys<SortWide>(tmp, b, in, out, (size_t)n, (unsigned)shift, (unsigned)(shift + bits), 0);
    return rocprim::radix_sort_keys(tmp, b, in, out, (size_t)n, (unsigned)shift, (unsigned)(shift + bits), 0);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int64_t cap = 48000000;
    uint64_t *a, *b, *c; CK(hipMalloc(&a, cap * 8)); CK(hipMalloc(&b, cap * 8)); CK(hipMalloc(&c, cap * 8));
    size_t tb = 0; CK(ref_sort(nullptr, tb, a, b, cap, 20, 31)); { size_t t2 = 0; CK(ref_sort(nullptr, t2, a, b, cap, 20, 18)); tb = std::max(tb, t2); }
    void* tmp; CK(hipMalloc(&tmp, tb));
    ItemSorter srt; if (srt.ensure(cap, 0)) return 1;
    int bad = 0, cases = 0;
    std::vector<uint64_t> hb, hc;
#ifdef RS_TIMING_ONLY
    if (0)
#endif
    for (int64_t n : {1ll, 2ll, 63ll, 64ll, 65ll, 1023ll, 1025ll, 4095ll, 4096ll, 4097ll, 8191ll, 100000ll, 1000003ll})
        for (int bits : {1, 2, 5, 8, 9, 10, 11, 17, 18, 19, 20, 21, 24, 30, 31})
            for (int shift : {12, 30})
                for (int skew : {0, 1}) {
                    if (shift + bits > 63) continue;
                    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, n, bits, shift, skew);
                    size_t bb = tb; CK(ref_sort(tmp, bb, a, b, n, shift, bits));
                    CK(hipMemsetAsync(c, 0xFF, n * 8, 0));
                    if (srt.sort(a, c, n, shift, bits, 0)) return 1;
                    hb.resize(n); hc.resize(n);
                    CK(hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost));
                    cases++;
                    if (hb != hc) { bad++; if (bad < 10) printf("MISMATCH n %lld bits %d shift %d skew %d\n", (long long)n, bits, shift, skew); }
                }
    printf("edge cases: %d compared with rocPRIM, %d mismatches\n", cases, bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int64_t n : {1600000ll, 7200000ll, 12800000ll, 48000000ll})
        for (int bits : {17, 18, 20}) {
            const int shift = 21;
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, n, bits, shift, 0);
            float t[2] = {1e9f, 1e9f};
            for (int r = 0; r < 5; r++)
                for (int w = 0; w < 2; w++) {
                    CK(hipEventRecord(e0, 0));
                    if (w == 0) { size_t bb = tb; CK(ref_sort(tmp, bb, a, b, n, shift, bits)); }
                    else if (srt.sort(a, c, n, shift, bits, 0)) return 1;
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[w] = std::min(t[w], ms);
                }
            hb.resize(n); hc.resize(n);
            CK(hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost));
            printf("n %9lld  %2d key bits: rocPRIM %7.3f ms | item_sort.h %7.3f ms (%.2fx)  %s\n", (long long)n, bits, t[0], t[1], t[0] / t[1], hb == hc ? "identical" : "MISMATCH");
            if (hb != hc) bad++;
        }
    return bad ? 2 : 0;
}
