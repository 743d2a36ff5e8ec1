// Probe for the split-fp16 (hi + lo, 2^12 pre-scale) form of the loss sweeps on gfx950:
//  part 1  accuracy of S = <x, y> (104 columns, unit rows) from three v_mfma_f32_16x16x32_f16 products (hi.hi + hi.lo + lo.hi, the tail's
//          three products packed into the k slots of ONE MFMA) against the exact-fp32 MFMA chain and an fp64 host sum;
//  part 2  does the f16 MFMA honour fp16 subnormal inputs?
//  part 3  issue rates: 16x16x32_f16, 32x32x16_f16, the legacy 16x16x16f16; an MFMA-only wave beside an MFMA + VALU (exp2) wave on one SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int DP = 104;
__device__ inline void split16(float v, _Float16& h, _Float16& l) { h = (_Float16)v; l = (_Float16)(v - (float)h); }

// one wave per 16 x 16 tile: S[i][j] = <X[i], Y[j]>
__global__ __launch_bounds__(64) void acc_kernel(const float* X, const float* Y, float* S32, float* S16, int ntiles) {
    const int t = blockIdx.x, lane = threadIdx.x, l15 = lane & 15, g4 = lane >> 4;
    const float* x = X + (size_t)(t * 16 + l15) * DP;      // A rows
    const float* y = Y + (size_t)(t * 16 + l15) * DP;      // B columns
    f32x4 a32 = {0, 0, 0, 0};
    for (int k = 0; k < DP; k += 4) a32 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[k + g4], y[k + g4], a32, 0, 0, 0);
    const float sc = 4096.f;
    f32x4 a16 = {0, 0, 0, 0};
    for (int q = 0; q < 3; ++q) {
        f16x8 ah, al, bh, bl;
        for (int e = 0; e < 8; ++e) {
            _Float16 h, l;
            split16(x[32 * q + 8 * g4 + e] * sc, h, l); ah[e] = h; al[e] = l;
            split16(y[32 * q + 8 * g4 + e] * sc, h, l); bh[e] = h; bl[e] = l;
        }
        a16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a16, 0, 0, 0);
        a16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a16, 0, 0, 0);
        a16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a16, 0, 0, 0);
    }
    {   // tail columns 96..103: k group 0: hi.hi, 1: hi.lo, 2: lo.hi, 3: lo.lo
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            _Float16 xh, xl, yh, yl;
            split16(x[96 + e] * sc, xh, xl); split16(y[96 + e] * sc, yh, yl);
            a[e] = (g4 < 2) ? xh : xl;
            b[e] = (g4 & 1) ? yl : yh;
        }
        a16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, a16, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
        const size_t o = ((size_t)t * 16 + 4 * g4 + r) * 16 + l15;
        S32[o] = a32[r];
        S16[o] = a16[r] * (1.f / (sc * sc));
    }
}

__global__ __launch_bounds__(64) void denorm_kernel(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
    if ((lane >> 4) == 0) { a[0] = (_Float16)3.0e-6f; b[0] = (_Float16)1024.f; }      // 3e-6 is subnormal in fp16 (min normal 6.1e-5)
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) { out[0] = c[0]; out[1] = (float)a[0] * 1024.f; }
}

constexpr int N_IT = 20000;
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out) {
    const int wave = threadIdx.x >> 6;
    f16x8 a8, b8; f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 1e-3f + e); b8[e] = (_Float16)1.0f; }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    f32x4 c[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f32x16 d[2]; for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) d[j][e] = 0.f;
    float v[8]; for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-4f + j;
    if (MODE == 0) {            // 8 waves: 16x16x32_f16, 4 independent accumulators
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j], 0, 0, 0);
    } else if (MODE == 1) {     // 8 waves: 32x32x16_f16, 2 accumulators (4 per iteration)
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[j & 1], 0, 0, 0);
    } else if (MODE == 2) {     // 8 waves: legacy 16x16x16f16
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[j], 0, 0, 0);
    } else if (MODE == 3) {     // waves 0-3: 4 x 16x16x32 per iteration; waves 4-7: 2 x 32x32x16 + 24 VALU (8 of them exp2) per iteration
        if (wave < 4) {
            for (int i = 0; i < N_IT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j], 0, 0, 0);
        } else {
            for (int i = 0; i < N_IT; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    d[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = __builtin_amdgcn_exp2f(v[q] * 0.5f); v[4 + q] = fmaf(v[4 + q], 0.999f, v[q]); }
                }
            }
        }
    } else if (MODE == 4) {     // 4 waves only (0-3): 16x16x32
        if (wave < 4)
            for (int i = 0; i < N_IT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[j], 0, 0, 0);
    } else if (MODE == 5) {     // 4 waves only (4-7): 2 x 32x32x16 + 24 VALU
        if (wave >= 4)
            for (int i = 0; i < N_IT; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    d[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = __builtin_amdgcn_exp2f(v[q] * 0.5f); v[4 + q] = fmaf(v[4 + q], 0.999f, v[q]); }
                }
            }
    } else if (MODE == 6) {     // every wave: 4 x 16x16x32 + 2 x 32x32x16 + 24 VALU in ONE stream (what lockstep waves would do, but interleaved)
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                c[2 * j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[2 * j], 0, 0, 0);
                c[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[2 * j + 1], 0, 0, 0);
                d[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = __builtin_amdgcn_exp2f(v[q] * 0.5f); v[4 + q] = fmaf(v[4 + q], 0.999f, v[q]); }
            }
        }
    } else if (MODE == 7) {     // 8 waves VALU only: 24 VALU (8 exp2) per iteration
        for (int i = 0; i < N_IT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = __builtin_amdgcn_exp2f(v[q] * 0.5f); v[4 + q] = fmaf(v[4 + q], 0.999f, v[q]); }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += d[j][e];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
void run_rate(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(512), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(512), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rate mode %d  %-86s %8.3f ms  %8.1f cycles/iteration @2.4GHz\n", MODE, name, ms, ms * 1e-3 * 2.4e9 / N_IT);
}

int main() {
    const int ntiles = 8192, n = ntiles * 16;
    for (int corr10 = 0; corr10 <= 9; corr10 += 9) {
        const double corr = corr10 / 10.0;
        std::vector<float> X((size_t)n * DP, 0.f), Y((size_t)n * DP, 0.f);
        std::vector<double> base(100);
        srand(1 + corr10);
        auto rnd = []() { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
        for (auto& b : base) b = rnd();
        for (int which = 0; which < 2; ++which) {
            std::vector<float>& Z = which ? Y : X;
            for (int i = 0; i < n; ++i) {
                double r[100], nn = 0;
                for (int k = 0; k < 100; ++k) { r[k] = rnd() * (1 - corr) + base[k] * corr; nn += r[k] * r[k]; }
                nn = 1.0 / std::sqrt(nn);
                for (int k = 0; k < 100; ++k) Z[(size_t)i * DP + k] = (float)(r[k] * nn);
            }
        }
        float *dX, *dY, *d32, *d16;
        hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4); hipMalloc(&d32, (size_t)n * 16 * 4); hipMalloc(&d16, (size_t)n * 16 * 4);
        hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(ntiles), dim3(64), 0, 0, dX, dY, d32, d16, ntiles);
        std::vector<float> S32((size_t)n * 16), S16((size_t)n * 16);
        hipMemcpy(S32.data(), d32, S32.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(S16.data(), d16, S16.size() * 4, hipMemcpyDeviceToHost);
        double m32 = 0, m16 = 0, r32 = 0, r16 = 0, b32 = 0, b16 = 0, ms = 0;
        for (int t = 0; t < ntiles; ++t)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0;
                    for (int k = 0; k < 100; ++k) s += (double)X[(size_t)(t * 16 + i) * DP + k] * (double)Y[(size_t)(t * 16 + j) * DP + k];
                    const size_t o = ((size_t)t * 16 + i) * 16 + j;
                    const double e32 = S32[o] - s, e16 = S16[o] - s;
                    m32 = std::fmax(m32, std::fabs(e32)); m16 = std::fmax(m16, std::fabs(e16));
                    r32 += e32 * e32; r16 += e16 * e16; b32 += e32; b16 += e16; ms += std::fabs(s);
                }
        const double cnt = (double)n * 16;
        printf("accuracy corr %.1f  mean|S| %.3f   fp32 MFMA: max %.3e rms %.3e bias %+.3e    split-fp16 x3 MFMA: max %.3e rms %.3e bias %+.3e\n", corr, ms / cnt,
               m32, std::sqrt(r32 / cnt), b32 / cnt, m16, std::sqrt(r16 / cnt), b16 / cnt);
        hipFree(dX); hipFree(dY); hipFree(d32); hipFree(d16);
    }
    float* out; hipMalloc(&out, 4096);
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out);
    float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("denormal input: mfma gives %.6e, exact product of the fp16 value %.6e  -> %s\n", h[0], h[1], h[0] != 0.f ? "subnormals honoured" : "FLUSHED");
    run_rate<0>("8 waves: 4 x 16x16x32_f16", out);
    run_rate<1>("8 waves: 4 x 32x32x16_f16", out);
    run_rate<2>("8 waves: 4 x legacy 16x16x16f16", out);
    run_rate<3>("waves 0-3: 4 x 16x16x32 | waves 4-7: 2 x 32x32x16 + 24 VALU (8 exp2)", out);
    run_rate<4>("waves 0-3 only: 4 x 16x16x32", out);
    run_rate<5>("waves 4-7 only: 2 x 32x32x16 + 24 VALU (8 exp2)", out);
    run_rate<6>("8 waves, one stream each: 4 x 16x16x32 + 2 x 32x32x16 + 24 VALU", out);
    run_rate<7>("8 waves: 24 VALU (8 exp2) only", out);
    return 0;
}
