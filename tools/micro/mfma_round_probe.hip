// Does the matrix core's fp32 accumulation round to nearest, or does it truncate?  One wave accumulates N dependent MFMAs of positive
// bf16-representable operands (so a chopping adder shows up as a NEGATIVE mean error that grows like N, a rounding one as a random walk
// like sqrt(N)), with v_mfma_f32_16x16x32_bf16 (32 products per step), v_mfma_f32_16x16x4_f32 (4 per step; the same values as fp32) and
// v_mfma_f32_16x16x32_f16 (same values: they are also fp16-representable), against an fp64 sum of the same products.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_round_probe.hip -o /tmp/mfma_round_probe && /tmp/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A [N][16 rows][32 k], B [N][16 cols][32 k] floats (values with <= 8 significant bits)
__global__ __launch_bounds__(64) void probe(const float* A, const float* B, int N, float* out_bf, float* out_f32, float* out_f16, int small_first) {
    const int lane = threadIdx.x, l15 = lane & 15, g4 = lane >> 4;
    f32x4 cb = {0, 0, 0, 0}, cf = {0, 0, 0, 0}, ch = {0, 0, 0, 0};
    for (int n = 0; n < N; ++n) {
        const float* a = A + ((size_t)n * 16 + l15) * 32 + 8 * g4;
        const float* b = B + ((size_t)n * 16 + l15) * 32 + 8 * g4;
        bf16x8 ab, bb; f16x8 ah, bh;
        for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a[e]; bb[e] = (__bf16)b[e]; ah[e] = (_Float16)a[e]; bh[e] = (_Float16)b[e]; }
        cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, cb, 0, 0, 0);
        ch = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, ch, 0, 0, 0);
        // the fp32 MFMA takes k = g4 per step: 8 steps cover the same 32 k slots (k = 4 s + g4 <-> element (k >> 3, k & 7))
        for (int s = 0; s < 8; ++s) {
            const int k = 4 * s + g4;
            const float av = A[((size_t)n * 16 + l15) * 32 + k], bv = B[((size_t)n * 16 + l15) * 32 + k];
            cf = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, cf, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; ++r) {
        const int o = (4 * g4 + r) * 16 + l15;
        out_bf[o] = cb[r]; out_f32[o] = cf[r]; out_f16[o] = ch[r];
    }
}

int main() {
    for (int variant = 0; variant < 5; ++variant) {
        const int N = 4096;
        std::vector<float> A((size_t)N * 16 * 32), B((size_t)N * 16 * 32);
        srand(1 + variant);
        auto r8 = [](double lo, double hi) {                    // a value with 8 significant bits in [lo, hi)
            double v = lo + (hi - lo) * (rand() / (RAND_MAX + 1.0));
            int e; double f = frexp(v, &e);
            return (float)ldexp(floor(f * 256.0) / 256.0, e);
        };
        for (size_t i = 0; i < A.size(); ++i) {
            if (variant == 0) { A[i] = r8(0.5, 1.0); B[i] = r8(0.5, 1.0); }                       // all positive, similar size
            else if (variant == 1) { A[i] = r8(0.5, 1.0) * ((rand() & 1) ? 1.f : -1.f); B[i] = r8(0.5, 1.0); }   // random signs
            else if (variant == 2) { A[i] = r8(0.5, 1.0) * (float)ldexp(1.0, -(rand() % 12)); B[i] = r8(0.5, 1.0); }  // positive, 12 binades
            else if (variant == 3) { A[i] = -r8(0.5, 1.0) * (float)ldexp(1.0, -(rand() % 12)); B[i] = r8(0.5, 1.0); }  // NEGATIVE, 12 binades
            else { const size_t n = i / (16 * 32); A[i] = r8(0.5, 1.0) * ((n % 6) == 5 ? 1.f : ((n % 6) >= 2 ? 0.00390625f : 1.52587890625e-05f)); B[i] = r8(0.5, 1.0); }  // the six-plane pattern: 2 steps at 2^-16, 3 at 2^-8, 1 at 1, all positive
        }
        float *dA, *dB, *o1, *o2, *o3;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024); hipMalloc(&o3, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int N_use : {64, 512, 4096}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, N_use, o1, o2, o3, 0);
            float h1[256], h2[256], h3[256];
            hipMemcpy(h1, o1, 1024, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 1024, hipMemcpyDeviceToHost); hipMemcpy(h3, o3, 1024, hipMemcpyDeviceToHost);
            double mb = 0, mf = 0, mh = 0, rb = 0, rf = 0, rh = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ex = 0, mag = 0;
                for (int n = 0; n < N_use; ++n) for (int k = 0; k < 32; ++k) {
                    const double p = (double)A[((size_t)n * 16 + i) * 32 + k] * (double)B[((size_t)n * 16 + j) * 32 + k];
                    ex += p; mag += fabs(p);
                }
                const double ulp = ldexp(1.0, ilogb(fabs(ex) > 0 ? fabs(ex) : 1.0) - 23);
                const double eb = (h1[i * 16 + j] - ex) / ulp, ef = (h2[i * 16 + j] - ex) / ulp, eh = (h3[i * 16 + j] - ex) / ulp;
                mb += eb; mf += ef; mh += eh; rb += eb * eb; rf += ef * ef; rh += eh * eh;
            }
            printf("variant %d, %4d chained steps: error in ulps of the result, mean / rms over 256 outputs:  bf16 16x16x32 %+9.3f / %8.3f   f16 16x16x32 %+9.3f / %8.3f   f32 16x16x4 %+9.3f / %8.3f\n",
                   variant, N_use, mb / 256, sqrt(rb / 256), mh / 256, sqrt(rh / 256), mf / 256, sqrt(rf / 256));
        }
        hipFree(dA); hipFree(dB); hipFree(o1); hipFree(o2); hipFree(o3);
    }
    return 0;
}
