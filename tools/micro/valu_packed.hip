// Microbenchmark: is packed fp32 VALU (v_pk_fma_f32: two FMAs per lane per instruction) issued at the same rate as
// scalar v_fma_f32 on gfx950?  8 waves per workgroup (2 per SIMD), pure VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int N_IT = 20000;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float s = 0;
    if (MODE == 0) {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = a + j;
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], b, a);
        }
        for (int j = 0; j < 16; ++j) s += v[j];
    } else {
        f32x2 v[8], bb = {b, b}, aa = {a, a};
        for (int j = 0; j < 8; ++j) v[j] = f32x2{a + j, a - j};
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], bb, aa);
        }
        for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
    }
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* name, double flops_per_it) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms   %.1f TFLOP/s\n", name, ms, flops_per_it * N_IT * 256.0 * 512.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    run<0>(out, "16 x v_fma_f32 per iteration", 32.0);
    run<1>(out, "8 x v_pk_fma_f32 per iteration", 32.0);
    return 0;
}
