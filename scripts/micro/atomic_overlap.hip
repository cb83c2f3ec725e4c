// Do memory-side float atomics (the head rows of the mixed policy) and plain row traffic (the tail under locks) overlap in the memory system,
// or do their times add up as they do inside k_sgns_train_locked on cfg5?  Kernel P: random 512-byte rows of a 1 GiB
// table read and stored back write-through.  Kernel A: float atomics on random rows of a 10 MB table (20 480 rows: a head).  Each alone, then
// both at once on two streams.      hipcc --offload-arch=gfx950 -O3 -o /tmp/ao scripts/micro/atomic_overlap.hip && /tmp/ao
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void __launch_bounds__(256) k_plain(float* t, int64_t rows, int64_t per_group) {
    const int lane = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = mix(group + 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(t, 0, (int)(uint32_t)(rows * 512), 0x00020000);
    for (int64_t i = 0; i < per_group; i += 8) {
        v4u a[8], b[8]; uint32_t off[8];
#pragma unroll
        for (int z = 0; z < 8; z++) {
            s = s * 25214903917ull + 11; off[z] = (uint32_t)((s >> 16) % (uint64_t)rows) * 512u + lane * 16u;
            a[z] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[z], 0, 16); b[z] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[z] + 256), 0, 16);
        }
#pragma unroll
        for (int z = 0; z < 8; z++) { __builtin_amdgcn_raw_buffer_store_b128(a[z], rs, (int)off[z], 0, 16); __builtin_amdgcn_raw_buffer_store_b128(b[z], rs, (int)(off[z] + 256), 0, 16); }
    }
}
__global__ void __launch_bounds__(256) k_atomic(float* t, int64_t rows, int64_t per_group) {
    const int lane = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t s = mix(group + 77);
    for (int64_t i = 0; i < per_group; i++) {
        s = s * 25214903917ull + 11;
        float* p = t + ((s >> 16) % (uint64_t)rows) * 128 + lane;
#pragma unroll
        for (int m = 0; m < 8; m++) atomicAdd(p + 16 * m, 1e-9f);
    }
}
int main() {
    const int64_t rows_p = 1 << 21, rows_a = 20480;
    float *tp, *ta; CK(hipMalloc(&tp, rows_p * 512)); CK(hipMalloc(&ta, rows_a * 512));
    CK(hipMemset(tp, 0, rows_p * 512)); CK(hipMemset(ta, 0, rows_a * 512));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    // half the device each (128 CUs x 3 blocks x 16 groups), as the two kinds of work share one kernel in the trainer
    const int blocks = 128 * 3;
    const int64_t per_p = 3000, per_a = 420;
    for (int rep = 0; rep < 3; rep++) {
        float tP, tA, tB1, tB2;
        CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, s1, tp, rows_p, per_p); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tP, e0, e1));
        CK(hipEventRecord(f0, s2)); hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, s2, ta, rows_a, per_a); CK(hipEventRecord(f1, s2)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tA, f0, f1));
        CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, s1, tp, rows_p, per_p); CK(hipEventRecord(e1, s1));
        CK(hipEventRecord(f0, s2)); hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, s2, ta, rows_a, per_a); CK(hipEventRecord(f1, s2));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tB1, e0, e1)); CK(hipEventElapsedTime(&tB2, f0, f1));
        const double gp = (double)blocks * 16 * per_p * 1024 / 1e9, ga = (double)blocks * 16 * per_a * 512 / 1e9;
        printf("plain alone %.2f ms (%.2f TB/s read+write) | atomics alone %.2f ms (%.2f TB/s) | together: plain %.2f ms, atomics %.2f ms (sum of the two alone %.2f)\n",
               tP, gp / tP, tA, ga / tA, tB1, tB2, tP + tA);
    }
    return 0;
}
