// Microbenchmark: device-scope fp32 atomicAdd throughput into an L2/MALL-resident array (the "other-side" gradient of a
// single-sweep loss backward), and LDS ds_add_f32 throughput.
//   mode 0: every workgroup adds 32 x 104 floats (rows of 416 B, coalesced 64-lane segments) at pseudo-random row blocks of a
//           [rows x 104] array, no-return global atomics;  mode 1: same addresses, plain stores (ceiling);
//   mode 2: same, but 128 x 104 per visit;  mode 3: LDS ds_add_f32 into a 32 x 112 tile, 56 adds per lane per visit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k(int mode, float* arr, int rows, int iters, float* sink) {
    __shared__ float tile[32 * 112];
    const int tid = threadIdx.x;
    unsigned s = blockIdx.x * 2654435761u + 12345u;
    float v = 1e-9f * tid;
    for (int t = tid; t < 32 * 112; t += 256) tile[t] = 0.f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        if (mode == 3) {
#pragma unroll
            for (int q = 0; q < 14; ++q) atomicAdd(&tile[(q * 256 + tid) % (32 * 112)], v);   // 14 adds/lane = 32x112/256
            continue;
        }
        const int nrow = mode == 2 ? 128 : 32;
        const int r0 = (s >> 8) % (rows / nrow) * nrow;
        float* base = arr + (size_t)r0 * 104;
        const int n = nrow * 104;
        if (mode == 1) { for (int e = tid; e < n; e += 256) base[e] = v; }
        else { for (int e = tid; e < n; e += 256) atomicAdd(base + e, v); }
    }
    if (mode == 3 && tile[tid] == 123.f) sink[0] = 1.f;
}

int main(int argc, char** argv) {
    const int rows = 46080;                  // configs[1]: negatives rows per table
    float* arr; hipMalloc(&arr, (size_t)rows * 104 * 4 * 3); hipMemset(arr, 0, (size_t)rows * 104 * 4 * 3);
    float* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2)
        for (int mode = 0; mode < 4; ++mode) {
            const int grid = 256 * wg_per_cu;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, mode, arr, rows * 3, 10, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, mode, arr, rows * 3, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per = mode == 2 ? 128.0 * 104 : (mode == 3 ? 14.0 * 256 : 32.0 * 104);
            const double ops = (double)grid * iters * per;
            printf("wg/cu %d mode %d: %8.3f ms  %.3e float-ops/s  (%.2f TB/s payload)\n", wg_per_cu, mode, ms, ops / (ms * 1e-3), ops * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
