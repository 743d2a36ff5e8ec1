// Do f16 MFMA (v_mfma_f32_16x16x32_f16) and fp32 VALU overlap on one gfx950 SIMD?  (tools/micro/mfma_valu_overlap.hip asked the same for the
// fp32 MFMA: there they do NOT.)  512-thread workgroups = 2 waves per SIMD (wave w and w + 4 share a SIMD).
//   mode 0: waves 0-3 MFMA only          1: waves 4-7 VALU (fma) only        2: both (one of each per SIMD)
//   mode 3: waves 4-7 VALU (exp2) only   4: waves 0-3 MFMA + waves 4-7 exp2
//   mode 5+k (k = 0..6): every one of 4 waves (one per SIMD) interleaves 1 MFMA : 2k fma in one stream (k = 0: MFMA only)
//   mode 12+k (k=1..4): every one of 4 waves: 1 MFMA : k exp2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef BIG
typedef f32x16 acc_t;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef f32x4 acc_t;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
constexpr int N_IT = 20000;

template <int NV, bool EXP>
__device__ __forceinline__ void stream(acc_t* c, float* v, f16x8 a8, f16x8 b8) {
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = MFMA(a8, b8, c[j]);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int x = (j * NV + q) & 15;
                v[x] = EXP ? __builtin_amdgcn_exp2f(v[x]) : fmaf(v[x], 0.999f, 0.5f);
            }
        }
    }
}

__global__ __launch_bounds__(512) void k(int mode, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const long long t0 = __builtin_readcyclecounter();
    f16x8 a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 1e-3f + e); b8[e] = (_Float16)1.0f; }
    acc_t c[4]; for (int j = 0; j < 4; ++j) for (int e = 0; e < (int)(sizeof(acc_t) / 4); ++e) c[j][e] = 0.f;
    float v[16]; for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-4f + j * 0.01f;
    const bool mf = (mode == 0 || mode == 2 || mode == 4) && wave < 4;
    const bool vf = (mode == 1 || mode == 2) && wave >= 4;
    const bool ve = (mode == 3 || mode == 4) && wave >= 4;
    if (mf) {
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = MFMA(a8, b8, c[j]);
    } else if (vf) {
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], 0.999f, 0.5f);
    } else if (ve) {
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
    } else if (mode >= 5 && wave < 4) {
        switch (mode) {
            case 5: stream<0, false>(c, v, a8, b8); break;
            case 6: stream<2, false>(c, v, a8, b8); break;
            case 7: stream<4, false>(c, v, a8, b8); break;
            case 8: stream<6, false>(c, v, a8, b8); break;
            case 9: stream<8, false>(c, v, a8, b8); break;
            case 10: stream<10, false>(c, v, a8, b8); break;
            case 11: stream<12, false>(c, v, a8, b8); break;
            case 13: stream<1, true>(c, v, a8, b8); break;
            case 14: stream<2, true>(c, v, a8, b8); break;
            case 15: stream<3, true>(c, v, a8, b8); break;
            case 16: stream<4, true>(c, v, a8, b8); break;
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < (int)(sizeof(acc_t) / 4); ++e) s += c[j][e];
    for (int j = 0; j < 16; ++j) s += v[j];
    if (s == 123.456f) out[threadIdx.x] = s;
    const long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    float* out; (void)hipMalloc(&out, 4096);
    long long* cyc; (void)hipMalloc(&cyc, 64); long long hc[8];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"waves 0-3: 4 MFMA / it", "waves 4-7: 16 fma / it", "waves 0-3 MFMA | waves 4-7 fma (one of each per SIMD)", "waves 4-7: 8 exp2 / it",
                           "waves 0-3 MFMA | waves 4-7 exp2", "1 wave/SIMD: 4 x (1 MFMA)", "1 wave/SIMD: 4 x (1 MFMA + 2 fma)", "1 wave/SIMD: 4 x (1 MFMA + 4 fma)",
                           "1 wave/SIMD: 4 x (1 MFMA + 6 fma)", "1 wave/SIMD: 4 x (1 MFMA + 8 fma)", "1 wave/SIMD: 4 x (1 MFMA + 10 fma)", "1 wave/SIMD: 4 x (1 MFMA + 12 fma)",
                           "-", "1 wave/SIMD: 4 x (1 MFMA + 1 exp2)", "1 wave/SIMD: 4 x (1 MFMA + 2 exp2)", "1 wave/SIMD: 4 x (1 MFMA + 3 exp2)", "1 wave/SIMD: 4 x (1 MFMA + 4 exp2)"};
    for (int mode = 0; mode <= 16; ++mode) {
        if (mode == 12) continue;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, mode, out, cyc);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, mode, out, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
        printf("mode %2d  %-60s %8.3f ms   %7.1f cycles per iteration @2.4GHz   s_memtime per iteration: wave0 %.1f wave4 %.1f\n", mode, names[mode], ms, ms * 1e-3 * 2.4e9 / N_IT,
               (double)hc[0] / N_IT, (double)hc[4] / N_IT);
    }
    return 0;
}
