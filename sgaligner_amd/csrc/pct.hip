// Point-Cloud-Transformer object encoder ('pct' module) -- inference (eval-mode) kernels.
//
// Replaces, together with sga_gemm_ex (gemm.hip), the forward of NaivePCT (src/aligner/networks/pct.py:275-317) in
// eval mode: the per-point convolutions are GEMMs over the point-major activation matrix [T*N, C] with the eval-mode
// BatchNorms folded into weight/bias on the host; this file holds what is not a GEMM:
//
//   * the self-attention of SA.forward (pct.py:211-222).  q_conv and k_conv SHARE their weight (pct.py:199), so the
//     energy matrix E = Q Q^T / sqrt(da) is symmetric; attention = softmax over the LAST dim (rows), and the output
//     x_s = x_v @ attention sums over the FIRST index:  Xs[j,:] = sum_i softmax_i(E)[i,j] V[i,:].
//     Flash style, never materialising the N x N matrix: pass 1 (`attn_stats`) computes the row maximum m_i and
//     row sum l_i of every row (symmetry lets a wave hold "its" rows as MFMA columns, so both reductions are in-lane);
//     pass 2 (`attn_apply`) recomputes 32x32 energy tiles with lane = output row j, turns them into
//     p[i,j] = exp(E[i,j] - m_i) / l_i in the accumulator registers -- which are exactly the A-operand layout of the
//     next MFMA -- and chains  Xs[j, :] += p^T V  without leaving registers.  Exact fp32 MFMA (32x32x2).
//   * the max over the N points of every object (pct.py:308).
//
// Layouts: Q [T*N, 32], V / Xs [T*N, 128] row-major fp32 (leading dimensions passed), every object has N points.
#include "mfma_tiles.h"

namespace {

constexpr int PA_THREADS = 256;
constexpr int DA = 32;                 // channels // 4 of SA(128)
constexpr int DV = 128;
constexpr float LOG2E_F = 1.4426950408889634f;

// MFMA k mapping used for BOTH operands of the energy product: step s, half h  <->  channel 16*h + s, so a lane's
// 16 operand values are 16 contiguous floats of its row.
__device__ __forceinline__ void load_q16(const float* __restrict__ row, int h, float (&q)[16]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + 16 * h + 4 * v);
        q[4 * v + 0] = t[0]; q[4 * v + 1] = t[1]; q[4 * v + 2] = t[2]; q[4 * v + 3] = t[3];
    }
}

// ---- pass 1: m_i = max_j E[i,j],  l_i = sum_j exp(scale * (E[i,j] - m_i))      (E raw, scale = 1/sqrt(da))
__global__ __launch_bounds__(PA_THREADS) void attn_stats_kernel(const float* __restrict__ Q, long ldq, int T, int N,
                                                                float scale, float* __restrict__ mstat,
                                                                float* __restrict__ lstat) {
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int i0 = (blockIdx.y * 4 + wave) * 32;
    if (i0 >= N) return;
    const float* Qt = Q + (size_t)t * N * ldq;
    const int my_i = min(i0 + l31, N - 1);
    float bq[16];
    load_q16(Qt + (size_t)my_i * ldq, h, bq);
    const float c = scale * LOG2E_F;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < N; j0 += 32) {
        float aq[16];
        load_q16(Qt + (size_t)min(j0 + l31, N - 1) * ldq, h, aq);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s], bq[s], acc, 0, 0, 0);
        // acc[r] = E[j0 + row(r,h)][my_i]  (= E[my_i][j0 + row(r,h)], the matrix is symmetric)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (j0 + mfma32_row(r, h) >= N) acc[r] = -INFINITY;
            tmax = fmaxf(tmax, acc[r]);
        }
        const float mn = fmaxf(m, tmax);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f((acc[r] - mn) * c);      // exp2(-inf) = 0
        l = l * __builtin_amdgcn_exp2f((m - mn) * c) + sum;
        m = mn;
    }
    // the two halves of the wave hold disjoint j subsets of the same row
    const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    const float mn = fmaxf(m, mo);
    l = l * __builtin_amdgcn_exp2f((m - mn) * c) + lo * __builtin_amdgcn_exp2f((mo - mn) * c);
    if (h == 0 && i0 + l31 < N) {
        mstat[(size_t)t * N + i0 + l31] = mn;
        lstat[(size_t)t * N + i0 + l31] = l;
    }
}

// ---- pass 2: Xs[j, :] = sum_i exp(scale (E[i,j] - m_i)) / l_i * V[i, :]
constexpr int QS = 36;                 // LDS row strides (floats): 16-byte aligned, spread over the banks
constexpr int VS = DV + 4;
constexpr int BUF_F = 32 * QS + 32 * VS + 64;       // Q tile, V tile, m[32], 1/l[32]

__global__ __launch_bounds__(PA_THREADS) void attn_apply_kernel(const float* __restrict__ Q, long ldq,
                                                                const float* __restrict__ V, long ldv, int T, int N,
                                                                float scale, const float* __restrict__ mstat,
                                                                const float* __restrict__ lstat,
                                                                float* __restrict__ Xs, long ldx) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF_F];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int j0 = (blockIdx.y * 4 + wave) * 32;                // this wave's 32 output rows (may be past N: idle wave)
    const float* Qt = Q + (size_t)t * N * ldq;
    const float* Vt = V + (size_t)t * N * ldv;
    const float* mt = mstat + (size_t)t * N;
    const float* lt = lstat + (size_t)t * N;
    const float c = scale * LOG2E_F;

    float bq[16];
    load_q16(Qt + (size_t)min(j0 + l31, N - 1) * ldq, h, bq);
    f32x16 out[4];
    zero_acc<4>(out);

    auto stage = [&](int i0, float* buf) {                      // rows [i0, i0+32) of Q, V and the row statistics
        {
            const int r = tid >> 3, c4 = (tid & 7) * 4;          // 32 rows x 8 quads
            const int row = min(i0 + r, N - 1);
            *reinterpret_cast<f32x4*>(buf + r * QS + c4) = *reinterpret_cast<const f32x4*>(Qt + (size_t)row * ldq + c4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = k * PA_THREADS + tid, r = e >> 5, c4 = (e & 31) * 4;      // 32 rows x 32 quads
            const int row = min(i0 + r, N - 1);
            *reinterpret_cast<f32x4*>(buf + 32 * QS + r * VS + c4) = *reinterpret_cast<const f32x4*>(Vt + (size_t)row * ldv + c4);
        }
        if (tid < 32) {
            const bool ok = i0 + tid < N;
            buf[32 * QS + 32 * VS + tid] = ok ? mt[i0 + tid] : 0.f;
            buf[32 * QS + 32 * VS + 32 + tid] = ok ? 1.f / lt[i0 + tid] : 0.f;      // rows past N contribute nothing
        }
    };

    const int ntile = (N + 31) / 32;
    stage(0, lds);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();                                         // tile `it` staged / other buffer free
        if (it + 1 < ntile) stage((it + 1) * 32, lds + ((it + 1) & 1) * BUF_F);
        const float* buf = lds + (it & 1) * BUF_F;
        const float* qs = buf + l31 * QS + 16 * h;               // A operand: lane = row i of the tile
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(qs + 4 * v);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], bq[4 * v + u], acc, 0, 0, 0);
        }
        // acc[r] = E[i = tile row(r,h)][j = this lane]  ->  p = exp(scale (E - m_i)) / l_i, in place
        const float* ms = buf + 32 * QS + 32 * VS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(ms + 8 * g + 4 * h);
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(ms + 32 + 8 * g + 4 * h);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[4 * g + u] = __builtin_amdgcn_exp2f((acc[4 * g + u] - m4[u]) * c) * r4[u];
        }
        // Xs[j, :] += sum_i p[i,j] V[i, :]   (A = p from the accumulators, B = V rows from LDS)
        const float* vs = buf + 32 * QS + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* vrow = vs + mfma32_row(s, h) * VS;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) out[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[s], vrow[ct * 32], out[ct], 0, 0, 0);
        }
    }
    if (j0 >= N) return;
    float* xo = Xs + (size_t)t * N * ldx;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + mfma32_row(r, h);
            if (j < N) xo[(size_t)j * ldx + ct * 32 + l31] = out[ct][r];
        }
}

// ---- G[t, c] = max_n Y[t*N + n, c]
__global__ void segment_max_kernel(const float* __restrict__ Y, long ldy, int T, int N, int C, float* __restrict__ G) {
    const int t = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    __shared__ float red[4][64];
    float m = -INFINITY;
    if (c < C)
        for (int n = rw; n < N; n += 4) m = fmaxf(m, Y[((size_t)t * N + n) * ldy + c]);
    red[rw][threadIdx.x & 63] = m;
    __syncthreads();
    if (rw == 0 && c < C) {
        const int k = threadIdx.x;
        G[(size_t)t * C + c] = fmaxf(fmaxf(red[0][k], red[1][k]), fmaxf(red[2][k], red[3][k]));
    }
}

}  // namespace

extern "C" int sga_pct_attention(const float* Q, long ldq, const float* V, long ldv, int T, int N, float* stats,
                                 float* Xs, long ldx, void* stream) {
    SGA_CHECK_ARG(Q && V && stats && Xs, "sga_pct_attention: null pointer");
    SGA_CHECK_ARG(T >= 0 && N >= 1 && ldq >= DA && ldv >= DV && ldx >= DV, "sga_pct_attention: bad sizes");
    SGA_CHECK_ARG(ldq % 4 == 0 && ldv % 4 == 0 && reinterpret_cast<uintptr_t>(Q) % 16 == 0 && reinterpret_cast<uintptr_t>(V) % 16 == 0,
                  "sga_pct_attention: Q / V rows must be 16-byte aligned");
    if (T == 0) return SGA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)DA);
    float* mstat = stats;
    float* lstat = stats + (size_t)T * N;
    dim3 grid(T, (N + 127) / 128);
    hipLaunchKernelGGL(attn_stats_kernel, grid, dim3(PA_THREADS), 0, s, Q, ldq, T, N, scale, mstat, lstat);
    hipLaunchKernelGGL(attn_apply_kernel, grid, dim3(PA_THREADS), 0, s, Q, ldq, V, ldv, T, N, scale, mstat, lstat, Xs, ldx);
    SGA_CHECK_LAUNCH("sga_pct_attention");
    return SGA_OK;
}

extern "C" int sga_segment_max(const float* Y, long ldy, int T, int N, int C, float* G, void* stream) {
    SGA_CHECK_ARG(Y && G && T >= 0 && N >= 1 && C >= 1 && ldy >= C, "sga_segment_max: bad argument");
    if (T == 0) return SGA_OK;
    hipLaunchKernelGGL(segment_max_kernel, dim3(T, (C + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), Y, ldy, T, N, C, G);
    SGA_CHECK_LAUNCH("sga_segment_max");
    return SGA_OK;
}
