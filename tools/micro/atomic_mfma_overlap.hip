// Microbenchmark: do fire-and-forget global fp32 atomics (the "other-side" gradient flush of a single-sweep loss backward)
// hide under fp32 MFMA work?  Workgroups of 512 threads (8 waves = 128 owner rows), one per CU.  Per step a wave issues
// NMFMA v_mfma_f32_16x16x4_f32 and NATOM no-return atomicAdd wave-instructions (64 consecutive floats each) into
// pseudo-random rows of a [138240 x 104] array (3 tables of configs[1] negatives).
//   ./atomic_mfma_overlap  -> table of (NMFMA, NATOM) combinations
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NMFMA, int NATOM>
__global__ __launch_bounds__(512) void k(float* arr, int rows, int iters, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float a = tid * 1e-3f, b = 1.0001f;
    unsigned s = (blockIdx.x * 8 + wave) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int j = 0; j < NMFMA; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 7], 0, 0, 0);
        s = s * 1664525u + 1013904223u;
        float* base = arr + (size_t)((s >> 8) % (rows - 64)) * 104;
#pragma unroll
        for (int q = 0; q < NATOM; ++q) atomicAdd(base + q * 64 + lane, acc[q & 7][0] * 1e-30f);
    }
    float t = 0;
    for (int j = 0; j < 8; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (t == 123.456f) sink[0] = t;
}

template <int NMFMA, int NATOM>
void run(float* arr, int rows, float* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400;
    hipLaunchKernelGGL((k<NMFMA, NATOM>), dim3(256), dim3(512), 0, 0, arr, rows, 4, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NMFMA, NATOM>), dim3(256), dim3(512), 0, 0, arr, rows, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double atoms = 256.0 * 8 * iters * NATOM * 64;
    printf("NMFMA %4d NATOM %3d: %8.3f ms  (%.1f us/step; %.2e atomics/s; %.1f MFMA TFLOP/s)\n", NMFMA, NATOM, ms, ms * 1e3 / iters,
           atoms / (ms * 1e-3), 256.0 * 8 * iters * NMFMA * 2048.0 / (ms * 1e-3) / 1e12);
}

int main() {
    const int rows = 138240;
    float* arr; (void)hipMalloc(&arr, (size_t)rows * 104 * 4); (void)hipMemset(arr, 0, (size_t)rows * 104 * 4);
    float* sink; (void)hipMalloc(&sink, 4);
    // single-sweep step of a wave (M = 3, 32-row other tile): 486 MFMAs; flush of 32 x 104 x 3 floats by 8 waves = 20 wave-atomics each
    run<486, 0>(arr, rows, sink);
    run<486, 10>(arr, rows, sink);
    run<486, 20>(arr, rows, sink);
    run<486, 40>(arr, rows, sink);
    run<243, 20>(arr, rows, sink);
    run<0, 20>(arr, rows, sink);
    return 0;
}
