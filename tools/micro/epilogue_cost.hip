// What does the gradient epilogue of the loss sweeps cost per instruction on gfx950 when it runs alone (one wave per SIMD)?
// The same arithmetic as sweeph.hip: joint coefficient (3 fma + 2 mul + 2 exp + mul + fma per pair), per-table coefficient (2 mul + 2 exp + mul + 2 fma),
// fp16 hi/lo split (cvt_pk + 2 fma_mix per two values), 16 pairs per lane and 3 tables per iteration.  Prints cycles per iteration; count the
// instructions with `hipcc -S`.   MODE 0: everything; 1: without the exps (v * k instead); 2: only the exps + their argument multiplies.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int N_IT = 4000;
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, float k0s, float k1s, float c0, float c1, float b0, long long* cyc) {
    float s[3][16];
    for (int m = 0; m < 3; ++m) for (int e = 0; e < 16; ++e) s[m][e] = in[(m * 16 + e) * 256 + threadIdx.x];
    unsigned acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_IT; ++it) {
        float cj[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float sj = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) sj = fmaf(b0, s[m][e], sj);
            if (MODE == 1) cj[e] = c0 * (sj * k0s) + c1 * (sj * k1s);
            else if (MODE == 2) cj[e] = fexp2(sj * k0s) + fexp2(sj * k1s);
            else cj[e] = c0 * fexp2(sj * k0s) + c1 * fexp2(sj * k1s);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float sv = s[m][2 * p + u];
                    if (MODE == 1) v[u] = fmaf(b0, cj[2 * p + u], fmaf(c0, sv * k0s, c1 * (sv * k1s)));
                    else if (MODE == 2) v[u] = fexp2(sv * k0s) + fexp2(sv * k1s) + cj[2 * p + u];
                    else v[u] = fmaf(b0, cj[2 * p + u], fmaf(c0, fexp2(sv * k0s), c1 * fexp2(sv * k1s)));
                }
                if (MODE == 2) { acc ^= __builtin_bit_cast(unsigned, v[0]) ^ __builtin_bit_cast(unsigned, v[1]); continue; }
                const f16x2 h = __builtin_convertvector(f32x2{v[0], v[1]}, f16x2);
                const unsigned hi = __builtin_bit_cast(unsigned, h);
                unsigned l;
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v[0]));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v[1]));
                acc ^= hi ^ l;
            }
        }
        // feed the result back so that iterations are not independent of each other's values (keeps the compiler honest), cheaply
        s[0][it & 15] += __builtin_bit_cast(float, (acc & 0x007fffffu) | 0x30000000u);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * 256] = __builtin_bit_cast(float, acc) + s[0][0];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, const float* in, float* out, long long* cyc) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, in, out, 3.4e-6f, 3.4e-7f, 0.37f, 0.21f, 0.33f, cyc);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %8.1f cycles per iteration (16 pairs x 3 tables per lane)\n", name, (double)h / N_IT);
}
int main() {
    float *in, *out; (void)hipMalloc(&in, 48 * 256 * 4); (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMemset(in, 0x3f, 48 * 256 * 4);
    long long* cyc; (void)hipMalloc(&cyc, 64);
    run<0>("full epilogue", in, out, cyc);
    run<1>("without the 128 v_exp", in, out, cyc);
    run<2>("only argument multiplies + v_exp + adds", in, out, cyc);
    return 0;
}
