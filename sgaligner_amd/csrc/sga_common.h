// Shared device/host helpers for the sgaligner_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// The public C ABI: every translation unit sees the prototypes it implements, so a definition that drifts from
// include/sgaligner_hip.h is a compile error (conflicting types for an extern "C" function).
#include "../../include/sgaligner_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SGA_OK 0
#define SGA_ERR_ARG 1
#define SGA_ERR_HIP 2
#define SGA_ERR_WORKSPACE 3

// thread-local last-error text, read through sga_last_error()
void sga_set_error(const char* fmt, ...);

#define SGA_CHECK_ARG(cond, ...)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            sga_set_error(__VA_ARGS__);                  \
            return SGA_ERR_ARG;                          \
        }                                                \
    } while (0)

#define SGA_CHECK_LAUNCH(name)                                                      \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            sga_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
            return SGA_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

static inline int sga_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// row of C/D register r for lane-half h in the 32x32 MFMA accumulator layout
// (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
