// Microbenchmark: issue rate of v_mfma_f32_4x4x1_16B_f32 vs v_mfma_f32_16x16x4_f32 on gfx950, alone and mixed 6:1 (the
// shape of the loss sweep's gradient step).  One wave per SIMD (256 threads, 256 workgroups), 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N_IT = 20000;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = in[i];
    __syncthreads();
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0;
    for (int i = 0; i < N_IT; ++i) {
        if (MODE >= 3) { a = a * 1.0001f + 0.5f; }
        float bb[7];
        if (MODE >= 3) { for (int j = 0; j < 7; ++j) bb[j] = sh[(i * 7 + j * 16 + (threadIdx.x & 15)) & 4095]; }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (MODE >= 3) { if (MODE == 3 || j < 6) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb[j], acc[j], 0, 0, 0); else acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb[j], acc[j], 0, 0, 0); }
            else if (MODE == 0 || (MODE == 2 && j < 6)) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* name, int threads) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, out + 1024);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, out + 1024);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s threads %d %8.3f ms  %.1f ns per group of 7\n", name, threads, ms, ms * 1e6 / N_IT);
}

int main() {
    float* out; (void)hipMalloc(&out, 65536); (void)hipMemset(out, 0, 65536);
    for (int t = 256; t <= 512; t += 256) {
        run<0>(out, "7 x 16x16x4", t);
        run<1>(out, "7 x 4x4x1", t);
        run<2>(out, "6 x 16x16x4 + 1 x 4x4x1", t);
        run<3>(out, "7 x 16x16x4, B from LDS", t);
        run<4>(out, "6 + 1 x 4x4x1, B from LDS", t);
    }
    return 0;
}
