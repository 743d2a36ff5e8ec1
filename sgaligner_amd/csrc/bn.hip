// BatchNorm1d over point-major activations [R, C] (R = objects x points rows, C channels) fused with the activation
// and the residual that follow it in the PCT object encoder -- the train-mode layers of src/aligner/networks/pct.py:
// Embedding :122-123 (BN + ReLU), SA :226 (BN + ReLU, then x + x_s :229), NaivePCT.linear :289-293 (BN +
// LeakyReLU(0.2)), the head :311-315 (BN + ReLU on [T, C]).
//
// Four HBM-bound passes (a thread owns one channel of a row strip, lanes = consecutive channels -> coalesced):
//   bn_stats      sum x, sum x^2 per channel               (fp32 partials -> fp64 atomics)
//   bn_apply      y = act(x * scale + shift) (+ resid)     scale = gamma * rstd, shift = beta - mean * scale (host, C values)
//   bn_bwd_stats  g = dy * act'(x * scale + shift);  sum g, sum g * xhat per channel        (= dbeta, dgamma)
//   bn_bwd_apply  dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))
// The statistics of a batch couple every row, so each layer is (stats, apply): two reads of x, one write of y.
#include "sga_common.h"

namespace {

constexpr int BN_THREADS = 256;

__device__ __forceinline__ float act_fwd(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : 0.2f * v) : v); }
__device__ __forceinline__ float act_grad(float pre, int act) { return act == 1 ? (pre > 0.f ? 1.f : 0.f) : (act == 2 ? (pre > 0.f ? 1.f : 0.2f) : 1.f); }

__global__ void bn_stats_kernel(const float* __restrict__ X, long ld, int R, int C, double* __restrict__ sums) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    float s = 0.f, q = 0.f;
    if (c < C)
        for (int r = blockIdx.y * 4 + rw; r < R; r += gridDim.y * 4) {
            const float v = X[(size_t)r * ld + c];
            s += v;
            q = fmaf(v, v, q);
        }
    __shared__ float red[2][4][64];
    red[0][rw][threadIdx.x & 63] = s;
    red[1][rw][threadIdx.x & 63] = q;
    __syncthreads();
    if (rw == 0 && c < C) {
        const int k = threadIdx.x;
        atomicAdd(sums + c, (double)red[0][0][k] + (double)red[0][1][k] + (double)red[0][2][k] + (double)red[0][3][k]);
        atomicAdd(sums + C + c, (double)red[1][0][k] + (double)red[1][1][k] + (double)red[1][2][k] + (double)red[1][3][k]);
    }
}

__global__ void bn_apply_kernel(const float* __restrict__ X, long ldx, int R, int C, const float* __restrict__ scale,
                                const float* __restrict__ shift, int act, const float* __restrict__ resid, long ldr,
                                float* __restrict__ Y, long ldy) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c];
    for (int r = blockIdx.y * 4 + rw; r < R; r += gridDim.y * 4) {
        float v = act_fwd(fmaf(X[(size_t)r * ldx + c], sc, sh), act);
        if (resid) v += resid[(size_t)r * ldr + c];
        Y[(size_t)r * ldy + c] = v;
    }
}

__global__ void bn_bwd_stats_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ dY, long ldd, int R,
                                    int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                    double* __restrict__ sums) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    float s = 0.f, q = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
        for (int r = blockIdx.y * 4 + rw; r < R; r += gridDim.y * 4) {
            const float x = X[(size_t)r * ldx + c];
            const float g = dY[(size_t)r * ldd + c] * act_grad(fmaf(x, sc, sh), act);
            s += g;
            q = fmaf(g, (x - mu) * rs, q);
        }
    }
    __shared__ float red[2][4][64];
    red[0][rw][threadIdx.x & 63] = s;
    red[1][rw][threadIdx.x & 63] = q;
    __syncthreads();
    if (rw == 0 && c < C) {
        const int k = threadIdx.x;
        atomicAdd(sums + c, (double)red[0][0][k] + (double)red[0][1][k] + (double)red[0][2][k] + (double)red[0][3][k]);
        atomicAdd(sums + C + c, (double)red[1][0][k] + (double)red[1][1][k] + (double)red[1][2][k] + (double)red[1][3][k]);
    }
}

// train: dx = gr * (g - mg - xhat * mgx) with gr = gamma * rstd, mg = mean(g), mgx = mean(g xhat); eval: dx = scale * g
__global__ void bn_bwd_apply_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ dY, long ldd, int R,
                                    int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ mg, const float* __restrict__ mgx, int act,
                                    float* __restrict__ dX, long ldo) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    const float a = mg ? mg[c] : 0.f, b = mgx ? mgx[c] : 0.f;
    for (int r = blockIdx.y * 4 + rw; r < R; r += gridDim.y * 4) {
        const float x = X[(size_t)r * ldx + c];
        const float g = dY[(size_t)r * ldd + c] * act_grad(fmaf(x, sc, sh), act);
        dX[(size_t)r * ldo + c] = sc * (g - a - (x - mu) * rs * b);
    }
}

// per-channel arithmetic between the statistics pass and the apply pass, as ONE launch (it was ~20 one-element-per-channel torch ops
// per BatchNorm call: the PCT encoder has 9 of them per step and its small-batch step is bound by launch count).
//   training: mean = s1 / R, var = max(s2 / R - mean^2, 0) (biased, what BN normalises with); running_mean / running_var updated
//   with `momentum` like nn.BatchNorm1d (unbiased running variance), num_batches_tracked += 1;  eval: the running statistics.
//   out[0..C) = scale = gamma * rstd, out[C..2C) = shift = beta - mean * scale, out[2C..3C) = mean, out[3C..4C) = rstd
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int R, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ num_batches, float momentum, float eps, int training, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && num_batches) num_batches[0] += 1;
    if (c >= C) return;
    float mean, rstd;
    if (training) {
        const double m = sums[c] / (double)R;
        double v = sums[C + c] / (double)R - m * m;
        if (v < 0.0) v = 0.0;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * ((double)R / (double)(R > 1 ? R - 1 : 1)));
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(v + (double)eps));
    } else {
        mean = running_mean[c];
        rstd = rsqrtf(running_var[c] + eps);
    }
    const float sc = gamma[c] * rstd;
    out[c] = sc; out[C + c] = beta[c] - mean * sc; out[2 * C + c] = mean; out[3 * C + c] = rstd;
}

// backward counterpart: sums = [sum g | sum g xhat] (doubles) -> out = [dbeta | dgamma | mean_g | mean_gx] floats
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, int R, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double a = sums[c], b = sums[C + c];
    out[c] = (float)a; out[C + c] = (float)b; out[2 * C + c] = (float)(a / (double)R); out[3 * C + c] = (float)(b / (double)R);
}

inline dim3 bn_grid(int R, int C) {
    int gy = (R + 255) / 256;
    if (gy > 1024) gy = 1024;
    if (gy < 1) gy = 1;
    return dim3((C + 63) / 64, gy);
}

}  // namespace

extern "C" int sga_bn_stats(const float* X, long ldx, int R, int C, double* sums, void* stream) {
    SGA_CHECK_ARG(X && sums && R >= 0 && C >= 1 && ldx >= C, "sga_bn_stats: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(sums, 0, (size_t)2 * C * sizeof(double), s) != hipSuccess) { sga_set_error("sga_bn_stats: memset failed"); return SGA_ERR_HIP; }
    if (R == 0) return SGA_OK;
    hipLaunchKernelGGL(bn_stats_kernel, bn_grid(R, C), dim3(BN_THREADS), 0, s, X, ldx, R, C, sums);
    SGA_CHECK_LAUNCH("sga_bn_stats");
    return SGA_OK;
}

extern "C" int sga_bn_finalize(const double* sums, int R, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps, int training, float* out,
                               void* stream) {
    SGA_CHECK_ARG((sums || !training) && gamma && beta && running_mean && running_var && out && C >= 1 && R >= 0, "sga_bn_finalize: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), sums, R, C, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, training, out);
    SGA_CHECK_LAUNCH("sga_bn_finalize");
    return SGA_OK;
}

extern "C" int sga_bn_bwd_finalize(const double* sums, int R, int C, float* out, void* stream) {
    SGA_CHECK_ARG(sums && out && C >= 1 && R >= 1, "sga_bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), sums, R, C, out);
    SGA_CHECK_LAUNCH("sga_bn_bwd_finalize");
    return SGA_OK;
}

extern "C" int sga_bn_apply(const float* X, long ldx, int R, int C, const float* scale, const float* shift, int act,
                            const float* resid, long ldr, float* Y, long ldy, void* stream) {
    SGA_CHECK_ARG(X && scale && shift && Y && R >= 0 && C >= 1 && ldx >= C && ldy >= C && act >= 0 && act <= 2, "sga_bn_apply: bad argument");
    if (R == 0) return SGA_OK;
    hipLaunchKernelGGL(bn_apply_kernel, bn_grid(R, C), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), X, ldx, R, C, scale, shift, act, resid, ldr, Y, ldy);
    SGA_CHECK_LAUNCH("sga_bn_apply");
    return SGA_OK;
}

extern "C" int sga_bn_bwd_stats(const float* X, long ldx, const float* dY, long ldd, int R, int C, const float* scale,
                                const float* shift, const float* mean, const float* rstd, int act, double* sums,
                                void* stream) {
    SGA_CHECK_ARG(X && dY && scale && shift && mean && rstd && sums && R >= 0 && C >= 1, "sga_bn_bwd_stats: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(sums, 0, (size_t)2 * C * sizeof(double), s) != hipSuccess) { sga_set_error("sga_bn_bwd_stats: memset failed"); return SGA_ERR_HIP; }
    if (R == 0) return SGA_OK;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, bn_grid(R, C), dim3(BN_THREADS), 0, s, X, ldx, dY, ldd, R, C, scale, shift, mean, rstd, act, sums);
    SGA_CHECK_LAUNCH("sga_bn_bwd_stats");
    return SGA_OK;
}

extern "C" int sga_bn_bwd_apply(const float* X, long ldx, const float* dY, long ldd, int R, int C, const float* scale,
                                const float* shift, const float* mean, const float* rstd, const float* mean_g,
                                const float* mean_gx, int act, float* dX, long ldo, void* stream) {
    SGA_CHECK_ARG(X && dY && scale && shift && mean && rstd && dX && R >= 0 && C >= 1, "sga_bn_bwd_apply: bad argument");
    if (R == 0) return SGA_OK;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, bn_grid(R, C), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), X, ldx, dY, ldd, R, C, scale,
                       shift, mean, rstd, mean_g, mean_gx, act, dX, ldo);
    SGA_CHECK_LAUNCH("sga_bn_bwd_apply");
    return SGA_OK;
}
