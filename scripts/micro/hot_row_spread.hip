// Microbenchmark: float atomics of EVERY wave on ONE 512-byte row (the Huffman root's situation) — the row's eight 64-byte lines contiguous, or spaced
// `stride` bytes apart (different memory channels?).  Prints row updates per second; one row update = 8 wave-wide instructions... here: one 16-lane group adds
// 16 bytes per lane to each of the row's lines it owns (lane l: line l / 2... see below), as the trainers do (a 16-lane group covers 256 B per instruction).
// hipcc --offload-arch=gfx950 -O3 hot_row_spread.hip -o hot_row_spread && ./hot_row_spread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// a row = 128 floats = 8 lines of 16 floats; line j of the row lives at base + j * stride_floats
__global__ void __launch_bounds__(256) k_hot(float* base, int64_t stride_floats, int iters, int n_rows, int64_t row_pitch_floats, int mapping) {
    const int lane = threadIdx.x & 15;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    for (int it = 0; it < iters; it++) {
        float* row = base + (int64_t)((grp + it) % n_rows) * row_pitch_floats;
        if (mapping == 0) {
            // the trainers' 16-byte-per-lane register layout: instruction (c, e) touches floats 64c + 4l + e — four quarter-filled lines per instruction, 32 requests a row
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int f = 64 * c + 4 * lane + e;
                    atomicAdd(row + (int64_t)(f >> 4) * stride_floats + (f & 15), 1.0f);
                }
        } else {
            // element order: instruction j covers line j (16 lanes x 4 bytes = one 64-byte request), 8 requests a row
#pragma unroll
            for (int j = 0; j < 8; j++) atomicAdd(row + (int64_t)j * stride_floats + lane, 1.0f);
        }
    }
}
int main() {
    float* d; const size_t bytes = 1ull << 30; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 2048, iters = 200;
    const int64_t groups = (int64_t)blocks * 16;
    for (int mapping : {1, 0})
    for (int n_rows : {1, 2, 8}) {
        for (int64_t stride : {64ll, 128ll, 256ll, 4096ll, 1048576ll}) {
            // rows spaced so that they never overlap: pitch = 8 * stride (contiguous rows when stride = 64)
            const int64_t pitch = stride == 64 ? 128 : (stride / 4) * 8 + 16;
            if ((size_t)(n_rows * pitch + 8 * stride / 4) * 4 > bytes) continue;
            hipMemset(d, 0, bytes);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k_hot, dim3(blocks), dim3(256), 0, 0, d, stride / 4, iters, n_rows, pitch, mapping);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double upd = (double)groups * iters;
            printf("mapping %d  rows %2d  line stride %8lld B: %8.3f ms  %.3e row updates/s  (%.1f ns per row update, %.2f ns per 64-B line)\n", mapping, n_rows, (long long)stride, ms, upd / ms * 1e3, ms * 1e6 / upd * 1.0, ms * 1e6 / upd / 8.0);
        }
    }
    // conservation of the last configuration is implied by atomics; print one element as a sanity value
    float v; hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost); printf("sample element %.0f\n", v);
    return 0;
}
