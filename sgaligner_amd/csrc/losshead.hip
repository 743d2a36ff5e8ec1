// Scalar head of OverallLoss (reference src/aligner/losses.py:114-152 with CustomMultiLossLayer :28-34) on the raw double-summed
// terms the tiled kernels return: two one-thread kernels (forward, backward) instead of ~25 + ~35 one-element torch launches.
// At the reference's own batch sizes the step is bound by the NUMBER of launches, and this arithmetic was a fifth of them.
//
//   sums = [ S_icl[0..M] (M modality tables, then the joint) | S_ial_a[0..M-1] | S_ial_b[0..M-1] ]
//   icl_k = S_icl[k] / A^2                                   (.mean() over the A x A matrix, losses.py:57)
//   ial_i = z_ial (a S_ial_a[i] + (1 - a) S_ial_b[i])        (IALLoss: zoom 0.1, alpha 0.5, losses.py:93-97)
//   align = zoom * sum_i (exp(-lvA_i) ial_i + lvA_i)         (losses.py:126 with the multi-loss layer)
//   uni   = sum_i (exp(-lvC_i) icl_i + lvC_i)                multi = icl_M
//   out   = [ align + uni + multi, uni, multi, align ]
#include "sga_common.h"

namespace {

template <typename TS>
__global__ void loss_head_fwd_kernel(const TS* __restrict__ sums, const float* __restrict__ lv_ial, const float* __restrict__ lv_icl,
                                     int M, double inv_a2, double z_ial, double alpha, double zoom, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nt = M + 1;
    double uni = 0.0, align = 0.0;
    for (int i = 0; i < M; ++i) {
        const double icl = (double)sums[i] * inv_a2;
        const double ial = z_ial * (alpha * (double)sums[nt + i] + (1.0 - alpha) * (double)sums[nt + M + i]);
        uni += exp(-(double)lv_icl[i]) * icl + (double)lv_icl[i];
        align += exp(-(double)lv_ial[i]) * ial + (double)lv_ial[i];
    }
    align *= zoom;
    const double multi = (double)sums[M] * inv_a2;
    out[0] = align + uni + multi; out[1] = uni; out[2] = multi; out[3] = align;
}

template <typename TS>
__global__ void loss_head_bwd_kernel(const double* __restrict__ gout, const TS* __restrict__ sums, const float* __restrict__ lv_ial,
                                     const float* __restrict__ lv_icl, int M, double inv_a2, double z_ial, double alpha, double zoom,
                                     TS* __restrict__ dsums, float* __restrict__ dlv_ial, float* __restrict__ dlv_icl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nt = M + 1;
    const double g_u = gout[0] + gout[1], g_m = gout[0] + gout[2], g_a = (gout[0] + gout[3]) * zoom;
    for (int i = 0; i < M; ++i) {
        const double ec = exp(-(double)lv_icl[i]), ea = exp(-(double)lv_ial[i]);
        const double icl = (double)sums[i] * inv_a2;
        const double ial = z_ial * (alpha * (double)sums[nt + i] + (1.0 - alpha) * (double)sums[nt + M + i]);
        dsums[i] = (TS)(g_u * ec * inv_a2);
        dsums[nt + i] = (TS)(g_a * ea * z_ial * alpha);
        dsums[nt + M + i] = (TS)(g_a * ea * z_ial * (1.0 - alpha));
        dlv_icl[i] = (float)(g_u * (1.0 - ec * icl));
        dlv_ial[i] = (float)(g_a * (1.0 - ea * ial));
    }
    dsums[M] = (TS)(g_m * inv_a2);
}

}  // namespace

extern "C" int sga_loss_head_fwd(const void* sums, int sums_f64, const float* lv_ial, const float* lv_icl, int M, double inv_a2,
                                 double z_ial, double alpha, double zoom, double* out, void* stream) {
    SGA_CHECK_ARG(sums && lv_ial && lv_icl && out && M >= 1 && M <= 16, "sga_loss_head_fwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sums_f64) hipLaunchKernelGGL(loss_head_fwd_kernel<double>, dim3(1), dim3(64), 0, s, static_cast<const double*>(sums), lv_ial, lv_icl, M, inv_a2, z_ial, alpha, zoom, out);
    else hipLaunchKernelGGL(loss_head_fwd_kernel<float>, dim3(1), dim3(64), 0, s, static_cast<const float*>(sums), lv_ial, lv_icl, M, inv_a2, z_ial, alpha, zoom, out);
    SGA_CHECK_LAUNCH("sga_loss_head_fwd");
    return SGA_OK;
}

extern "C" int sga_loss_head_bwd(const double* gout, const void* sums, int sums_f64, const float* lv_ial, const float* lv_icl, int M,
                                 double inv_a2, double z_ial, double alpha, double zoom, void* dsums, float* dlv_ial,
                                 float* dlv_icl, void* stream) {
    SGA_CHECK_ARG(gout && sums && lv_ial && lv_icl && dsums && dlv_ial && dlv_icl && M >= 1 && M <= 16, "sga_loss_head_bwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sums_f64) hipLaunchKernelGGL(loss_head_bwd_kernel<double>, dim3(1), dim3(64), 0, s, gout, static_cast<const double*>(sums), lv_ial, lv_icl, M, inv_a2, z_ial, alpha, zoom, static_cast<double*>(dsums), dlv_ial, dlv_icl);
    else hipLaunchKernelGGL(loss_head_bwd_kernel<float>, dim3(1), dim3(64), 0, s, gout, static_cast<const float*>(sums), lv_ial, lv_icl, M, inv_a2, z_ial, alpha, zoom, static_cast<float*>(dsums), dlv_ial, dlv_icl);
    SGA_CHECK_LAUNCH("sga_loss_head_bwd");
    return SGA_OK;
}
