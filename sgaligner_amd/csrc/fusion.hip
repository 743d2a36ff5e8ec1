// Modality fusion: joint = cat_m( softmax(weight)[m] * x_m / max(||x_m||_2, 1e-12) ).
//
// Replaces reference MultiModalFusion.forward, src/aligner/sg_aligner.py:30-35 (F.softmax over the
// [M,1] weight :32, F.normalize row-L2 with eps 1e-12 and the concat :33-34) and its autograd.
// HBM-bound (reads M*T*D, writes M*T*D floats): one wave per object row, coalesced row reads,
// wave-shuffle reduction for the row norm; everything fused into a single pass per direction.
#include "sga_common.h"

namespace {

constexpr int FU_MAXM = 8;
struct FusionPtrs { const float* p[FU_MAXM]; };
struct FusionOutPtrs { float* p[FU_MAXM]; };

__device__ __forceinline__ void softmax_weights(const float* __restrict__ weight, int M, float (&w)[FU_MAXM]) {
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < FU_MAXM; ++m) if (m < M) mx = fmaxf(mx, weight[m]);
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < FU_MAXM; ++m) { w[m] = (m < M) ? expf(weight[m] - mx) : 0.f; s += w[m]; }
#pragma unroll
    for (int m = 0; m < FU_MAXM; ++m) w[m] /= s;
}

__global__ void fusion_fwd_kernel(FusionPtrs embs, const float* __restrict__ weight, float* __restrict__ joint,
                                  int T, int D, int M) {
    float w[FU_MAXM];
    softmax_weights(weight, M, w);
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < T; t += gridDim.x * wpb) {
#pragma unroll
        for (int m = 0; m < FU_MAXM; ++m) {
            if (m >= M) break;
            const float* x = embs.p[m] + (size_t)t * D;
            float ss = 0.f;
            for (int d = lane; d < D; d += 64) { const float v = x[d]; ss += v * v; }
            ss = wave_sum(ss);
            const float scale = w[m] / fmaxf(sqrtf(ss), 1e-12f);
            float* o = joint + (size_t)t * M * D + (size_t)m * D;
            for (int d = lane; d < D; d += 64) o[d] = x[d] * scale;
        }
    }
}

__global__ void fusion_bwd_kernel(FusionPtrs embs, const float* __restrict__ weight, const float* __restrict__ gjoint,
                                  FusionOutPtrs gembs, double* __restrict__ a_accum, int T, int D, int M) {
    float w[FU_MAXM];
    softmax_weights(weight, M, w);
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    double a_loc[FU_MAXM];
#pragma unroll
    for (int m = 0; m < FU_MAXM; ++m) a_loc[m] = 0.0;
    for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < T; t += gridDim.x * wpb) {
#pragma unroll
        for (int m = 0; m < FU_MAXM; ++m) {
            if (m >= M) break;
            const float* x = embs.p[m] + (size_t)t * D;
            const float* g = gjoint + (size_t)t * M * D + (size_t)m * D;
            float ss = 0.f, xg = 0.f;
            for (int d = lane; d < D; d += 64) { const float v = x[d]; ss += v * v; xg += v * g[d]; }
            ss = wave_sum(ss);
            xg = wave_sum(xg);
            const float nrm = sqrtf(ss);
            const bool clamped = nrm < 1e-12f;
            const float inv = 1.f / fmaxf(nrm, 1e-12f);
            const float dot = xg * inv;                      // g . xhat
            a_loc[m] += (double)dot;
            float* gx = gembs.p[m] + (size_t)t * D;
            const float c1 = w[m] * inv;
            const float c2 = clamped ? 0.f : w[m] * dot * inv * inv;   // xhat * dot / n = x * dot / n^2
            for (int d = lane; d < D; d += 64) gx[d] = c1 * g[d] - c2 * x[d];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < FU_MAXM; ++m)
            if (m < M && a_loc[m] != 0.0) atomicAdd(a_accum + m, a_loc[m]);
    }
}

// softmax Jacobian: g_raw[m] = w_m * (a_m - sum_k w_k a_k)
__global__ void fusion_weight_grad_kernel(const float* __restrict__ weight, const double* __restrict__ a_accum,
                                          float* __restrict__ gweight, int M, int accumulate) {
    if (threadIdx.x != 0) return;
    float w[FU_MAXM];
    softmax_weights(weight, M, w);
    double dotwa = 0.0;
    for (int m = 0; m < M; ++m) dotwa += (double)w[m] * a_accum[m];
    for (int m = 0; m < M; ++m) {
        const float g = (float)((double)w[m] * (a_accum[m] - dotwa));
        gweight[m] = accumulate ? gweight[m] + g : g;
    }
}

int grid_for_rows(int T, int wpb) {
    int g = (T + wpb - 1) / wpb;
    const int cap = sga_num_cus() * 8;
    return g > cap ? cap : (g < 1 ? 1 : g);
}

}  // namespace

extern "C" int sga_fusion_fwd(const float* const* embs, int M, const float* weight, float* joint, int T, int D,
                              void* stream) {
    SGA_CHECK_ARG(M >= 1 && M <= FU_MAXM, "sga_fusion_fwd: modal_num %d outside [1,%d]", M, FU_MAXM);
    SGA_CHECK_ARG(embs && weight && (joint || T == 0) && T >= 0 && D >= 1, "sga_fusion_fwd: bad argument");
    if (T == 0) return SGA_OK;
    FusionPtrs p{};
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(embs[m], "sga_fusion_fwd: null table %d", m); p.p[m] = embs[m]; }
    hipLaunchKernelGGL(fusion_fwd_kernel, dim3(grid_for_rows(T, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), p, weight,
                       joint, T, D, M);
    SGA_CHECK_LAUNCH("sga_fusion_fwd");
    return SGA_OK;
}

extern "C" size_t sga_fusion_bwd_workspace_bytes(int M) { return sizeof(double) * (size_t)(M > 0 ? M : 1); }

extern "C" int sga_fusion_bwd(const float* const* embs, int M, const float* weight, const float* gjoint,
                              float* const* gembs, float* gweight, int T, int D, void* workspace,
                              size_t workspace_bytes, void* stream) {
    SGA_CHECK_ARG(M >= 1 && M <= FU_MAXM, "sga_fusion_bwd: modal_num %d outside [1,%d]", M, FU_MAXM);
    SGA_CHECK_ARG(embs && weight && (gjoint || T == 0) && gembs && gweight && T >= 0 && D >= 1, "sga_fusion_bwd: bad argument");
    if (workspace_bytes < sga_fusion_bwd_workspace_bytes(M) || !workspace) {
        sga_set_error("sga_fusion_bwd: workspace too small");
        return SGA_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    FusionPtrs p{};
    FusionOutPtrs q{};
    for (int m = 0; m < M; ++m) { p.p[m] = embs[m]; q.p[m] = gembs[m]; }
    double* acc = static_cast<double*>(workspace);
    if (hipMemsetAsync(acc, 0, sizeof(double) * M, s) != hipSuccess) { sga_set_error("sga_fusion_bwd: memset failed"); return SGA_ERR_HIP; }
    if (T > 0) hipLaunchKernelGGL(fusion_bwd_kernel, dim3(grid_for_rows(T, 4)), dim3(256), 0, s, p, weight, gjoint, q, acc, T, D, M);
    hipLaunchKernelGGL(fusion_weight_grad_kernel, dim3(1), dim3(64), 0, s, weight, acc, gweight, M, 0);
    SGA_CHECK_LAUNCH("sga_fusion_bwd");
    return SGA_OK;
}
