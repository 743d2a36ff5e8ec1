// Microbenchmark: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and plain fp32 VALU (v_fma_f32) overlap on one gfx950 SIMD?
// Workgroups of 512 threads (2 waves per SIMD).  mode 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA, 4-7 VALU
// (one of each per SIMD); 3: waves 0-3 MFMA, 4-7 idle; 4: waves 0-3 idle, 4-7 VALU; 5: each wave interleaves 1 MFMA : 4 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N_IT = 20000;

__global__ __launch_bounds__(512) void k(int mode, float* out) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8] = {a, a + 1, a + 2, a + 3, a + 4, a + 5, a + 6, a + 7};
    const bool do_mfma = mode == 0 || mode == 5 || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_valu = mode == 1 || mode == 5 || ((mode == 2 || mode == 4) && wave >= 4);
    if (mode == 5) {
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[(j * 4 + q) & 7] = fmaf(v[(j * 4 + q) & 7], b, a);
            }
        }
    } else if (do_mfma) {
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
    } else if (do_valu) {
        for (int i = 0; i < N_IT; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j & 7] = fmaf(v[j & 7], b, a);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"all MFMA (8 waves: 4 MFMA x N per wave)", "all VALU (16 FMA x N per wave)", "4 MFMA waves + 4 VALU waves", "4 MFMA waves only", "4 VALU waves only", "every wave 4 MFMA + 16 FMA interleaved"};
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d  %-50s %8.3f ms   (%.1f cycles per loop iteration @2.4GHz)\n", mode, names[mode], ms, ms * 1e-3 * 2.4e9 / N_IT);
    }
    return 0;
}
