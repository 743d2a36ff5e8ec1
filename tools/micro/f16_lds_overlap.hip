// Issue cost of LDS reads beside f16 MFMA on one gfx950 SIMD (one wave per SIMD, 256-thread workgroups): per iteration 4 x (1 MFMA 16x16x32_f16
// + NR ds_read_b128 | ds_read_b64_tr_b16), the read results consumed by the NEXT iteration's MFMAs (software-pipelined, as a real kernel would).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int N_IT = 20000;

template <int NR, bool TR>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;     // fp16 1.0
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = lds + wave * 8192 + lane * (TR ? 8 : 16);
    f32x4 c[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    u32x4 a[4], nx[4];
    for (int j = 0; j < 4; ++j) a[j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    const f16x8 b8 = __builtin_bit_cast(f16x8, a[0]);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[j]), b8, c[j], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int off = ((i + j * NR + r) & 7) * 1024;
                if (TR) {
                    const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off)));
                    nx[j][2 * (r & 1)] = v[0]; nx[j][2 * (r & 1) + 1] = v[1];
                } else {
                    nx[(j + r) & 3] = *reinterpret_cast<const u32x4*>(base + off);
                }
            }
        }
        if (NR > 0) for (int j = 0; j < 4; ++j) a[j] = nx[j];
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int NR, bool TR>
void run(const char* name, int grid, float* out, long long* cyc) {
    hipLaunchKernelGGL((k<NR, TR>), dim3(grid), dim3(256), 0, 0, out, cyc);
    (void)hipDeviceSynchronize();
    long long hc[4]; (void)hipMemcpy(hc, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-44s  %7.1f cycles per iteration (4 MFMA + %d reads)  -> %.2f cycles per read\n", name, (double)hc[0] / N_IT, 4 * NR, NR ? ((double)hc[0] / N_IT - 64.9) / (4 * NR) : 0.0);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    float* out; (void)hipMalloc(&out, 4096);
    long long* cyc; (void)hipMalloc(&cyc, 64);
    run<0, false>("MFMA only", grid, out, cyc);
    run<1, false>("MFMA + 1 ds_read_b128", grid, out, cyc);
    run<2, false>("MFMA + 2 ds_read_b128", grid, out, cyc);
    run<4, false>("MFMA + 4 ds_read_b128", grid, out, cyc);
    run<1, true>("MFMA + 1 ds_read_b64_tr_b16", grid, out, cyc);
    run<2, true>("MFMA + 2 ds_read_b64_tr_b16", grid, out, cyc);
    run<4, true>("MFMA + 4 ds_read_b64_tr_b16", grid, out, cyc);
    return 0;
}
