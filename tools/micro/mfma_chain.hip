// Microbenchmark: cost of dependent fp32 MFMA chains on gfx950.  One wave per SIMD (256 threads, 256 workgroups).
// NACC independent accumulators used round-robin; NACC = 1 is a fully dependent chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int N_IT = 20000;

template <int NACC, bool BIG>
__global__ __launch_bounds__(256) void k(float* out) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float s = 0;
    if (BIG) {
        f32x16 acc[NACC];
        for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0;
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
        }
        for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[NACC];
        for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0;
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % NACC], 0, 0, 0);
        }
        for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NACC, bool BIG>
void run(float* out, const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(256), 0, 0, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(256), 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s NACC=%d  %8.3f ms  %.1f ns per MFMA\n", name, NACC, ms, ms * 1e6 / (N_IT * 8.0));
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    run<1, true>(out, "32x32x2 f32");  run<2, true>(out, "32x32x2 f32");  run<4, true>(out, "32x32x2 f32");
    run<1, false>(out, "16x16x4 f32"); run<2, false>(out, "16x16x4 f32"); run<4, false>(out, "16x16x4 f32"); run<8, false>(out, "16x16x4 f32");
    return 0;
}
