// Dependent-issue latency of v_mfma_f32_16x16x32_f16 on gfx950: NCH independent accumulator chains issued round-robin, one wave per SIMD.
// With NCH chains a dependent MFMA is NCH issues (16 cycles each) behind its producer.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int N_IT = 20000;
template <int NCH>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc) {
    f16x8 a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 1e-3f + e); b8[e] = (_Float16)1.0f; }
    f32x4 c[NCH];
    for (int j = 0; j < NCH; ++j) c[j] = f32x4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) c[j % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j % NCH], 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < NCH; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> void run(float* out, long long* cyc) {
    hipLaunchKernelGGL(k<NCH>, dim3(8), dim3(256), 0, 0, out, cyc);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d chains: %.2f cycles per MFMA\n", NCH, (double)h / N_IT / 12);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    long long* cyc; (void)hipMalloc(&cyc, 64);
    run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<6>(out, cyc);
    return 0;
}
