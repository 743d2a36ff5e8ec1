// Scalar math of the contrastive / alignment loss shared by contrastive.hip (batch-global loss) and grouploss.hip
// (loss_group = b): reference src/aligner/losses.py:5-15 written per element, with its derivatives.
#pragma once
#include "sga_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float QEPS = 1e-9f;

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
#ifdef SGA_DBG_NOEXP
__device__ __forceinline__ float fexp2(float x) { return x; }
#else
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }


struct GV { float q, dd, dsa, dsb; };

// g = 1 / (1 + 1/u + 1/v + eps), u = d a + eps, v = d b + eps, and its derivatives wrt d, sa, sb; a = 1/(sa+eps),
// b = 1/(sb+eps) (losses.py:17-27 written per element).  Evaluated with ONE reciprocal: with
// w = 1 / ((1+eps) u v + u + v):  g = u v w,  g/u = v w,  g/v = u w  (transcendentals are quarter rate and this
// epilogue is VALU bound).
__device__ __forceinline__ GV g_full(float d, float a, float b) {
    const float u = fmaf(d, a, QEPS), v = fmaf(d, b, QEPS);
    const float uv = u * v;
    const float w = frcp(fmaf(1.f + QEPS, uv, u + v));
    const float qu = v * w, qv = u * w;               // g/u, g/v
    const float au = a * qu * qu, bv = b * qv * qv;
    GV o;
    o.q = uv * w;
    o.dd = au + bv;
    o.dsa = -d * a * au;
    o.dsb = -d * b * bv;
    return o;
}
// g with d g/dd, and the squared ratios the sum derivatives are made of:  dg/dsa = -d a^2 p,  dg/dsb = -d b^2 r
// (p = (g/u)^2, r = (g/v)^2).  The uniform factors -a^2 / -b^2 are applied once per wave when the partial sums are
// flushed, so an accumulation costs one fma on (weight * d).
struct GP { float q, dd, p, r; };
__device__ __forceinline__ GP g_parts(float d, float a, float b) {
    const float u = fmaf(d, a, QEPS), v = fmaf(d, b, QEPS);
    const float uv = u * v;
    const float w = frcp(fmaf(1.f + QEPS, uv, u + v));
    const float qu = v * w, qv = u * w;
    GP o;
    o.q = uv * w;
    o.p = qu * qu;
    o.r = qv * qv;
    o.dd = fmaf(a, o.p, b * o.r);
    return o;
}
__device__ __forceinline__ float g_val(float d, float a, float b) {
    const float u = fmaf(d, a, QEPS), v = fmaf(d, b, QEPS);
    const float uv = u * v;
    return uv * frcp(fmaf(1.f + QEPS, uv, u + v));
}


// Global scalar accumulators (loss sums, dL/d(sums), Gamma) are hit by every wave of every workgroup; a single
// set of addresses serialises in L2 (measured: ~10 of the 15 ms of the anchors backward).  Each accumulator therefore
// has SGA_SLOTS copies, a wave adds to copy (wave id mod SGA_SLOTS), and reduce_slots_kernel folds them into copy 0's
// final location.  Buffers passed to the C ABI hold (1 + SGA_SLOTS) * n doubles: [result n | slots].
constexpr int SGA_SLOTS = 128;
__device__ __forceinline__ int my_slot() {
    return (int)((blockIdx.x * gridDim.y + blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6)) % SGA_SLOTS;
}
__global__ void reduce_slots_kernel(double* __restrict__ buf, int n) {      // buf[0..n) = sum_s buf[n + s*n + i]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = 0.0;
    for (int sl = 0; sl < SGA_SLOTS; ++sl) v += buf[n + (size_t)sl * n + i];
    buf[i] = v;
}
static int zero_slots(double* buf, int n, hipStream_t s, const char* who) {
    if (hipMemsetAsync(buf, 0, (size_t)(1 + SGA_SLOTS) * n * sizeof(double), s) != hipSuccess) { sga_set_error("%s: memset failed", who); return SGA_ERR_HIP; }
    return SGA_OK;
}
static void fold_slots(double* buf, int n, hipStream_t s) {
    hipLaunchKernelGGL(reduce_slots_kernel, dim3((n + 63) / 64), dim3(64), 0, s, buf, n);
}


}  // namespace
