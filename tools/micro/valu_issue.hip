// Issue cost of single VALU instructions on gfx950, one wave per SIMD, exact instructions by inline asm (16 independent destination registers,
// 64 instructions per loop iteration): v_fma_f32, v_pk_fma_f32, v_mul_f32 (VOP2, SGPR operand), v_exp_f32, v_cvt_pk_f16_f32, v_fma_mixlo_f16,
// v_accvgpr_read, s_nop 0, and the same VALU streams with one v_mfma_f32_16x16x32_f16 after every 4 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int N_IT = 5000;
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE, bool WITH_MFMA>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float sc) {
    float v[16]; f32x2 w[16];
    for (int j = 0; j < 16; ++j) { v[j] = threadIdx.x * 1e-4f + j * 0.01f; w[j] = f32x2{v[j], v[j] + 1.f}; }
    f16x8 a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 1e-3f + e); b8[e] = (_Float16)1.0f; }
    f32x4 c[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    const f32x2 k2 = f32x2{0.999f, 0.998f};
    unsigned u = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = j4 * 4 + q;
                    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(sc));
                    if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[j]) : "v"(k2));
                    if (MODE == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[j]) : "s"(sc));
                    if (MODE == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
                    if (MODE == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(v[j]), "v"(v[(j + 1) & 15]));
                    if (MODE == 5) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(u) : "v"(u), "v"(v[j]));
                    if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[j]) : "v"(k2));
                    if (MODE == 7) asm volatile("s_nop 0");
                    if (MODE == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 1) & 15]));
                }
                if (WITH_MFMA) c[j4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j4], 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += v[j] + w[j][0] + w[j][1];
    for (int j = 0; j < 4; ++j) s += c[j][0];
    out[threadIdx.x + blockIdx.x * 256] = s + u;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, float* out, long long* cyc) {
    long long h0, h1;
    hipLaunchKernelGGL((k<MODE, false>), dim3(256), dim3(256), 0, 0, out, cyc, 0.999f);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h0, cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<MODE, true>), dim3(256), dim3(256), 0, 0, out, cyc, 0.999f);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h1, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s alone: %6.2f cycles each | 4 of them + 1 MFMA: %6.2f cycles per group (MFMA alone 16.2) -> %6.2f each on top of the MFMA\n", name,
           (double)h0 / N_IT / 64, (double)h1 / N_IT / 16, ((double)h1 / N_IT / 16 - 16.2) / 4);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    long long* cyc; (void)hipMalloc(&cyc, 64);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<2>("v_mul_f32 (SGPR operand)", out, cyc);
    run<3>("v_exp_f32", out, cyc);
    run<4>("v_cvt_pk_f16_f32", out, cyc);
    run<5>("v_fma_mixlo_f16", out, cyc);
    run<6>("v_pk_mul_f32", out, cyc);
    run<7>("s_nop 0", out, cyc);
    run<8>("v_mov_b32", out, cyc);
    return 0;
}
