// Point-Cloud-Transformer object encoder ('pct' module): attention forward / backward, point max.
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

// OWN == false (forward):  out[own j, :] = sum_{tile i} exp(s (E[i,j] - m_i)) / l_i * V[i, :]          (tile-row statistics)
// OWN == true  (backward):  out[own i, :] = sum_{tile j} exp(s (E[i,j] - m_i)) / l_i * dXs[j, :] = dV[i]  (own-row statistics)
template <bool OWN>
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
    const float m_own = OWN ? mt[min(j0 + l31, N - 1)] : 0.f;
    const float rl_own = OWN ? 1.f / lt[min(j0 + l31, N - 1)] : 0.f;
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
            buf[32 * QS + 32 * VS + 32 + tid] = ok ? (OWN ? 1.f : 1.f / lt[i0 + tid]) : 0.f;   // rows past N contribute nothing
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
            for (int u = 0; u < 4; ++u)
                acc[4 * g + u] = OWN ? __builtin_amdgcn_exp2f((acc[4 * g + u] - m_own) * c) * (rl_own * r4[u])
                                     : __builtin_amdgcn_exp2f((acc[4 * g + u] - m4[u]) * c) * r4[u];
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

// ---- delta_i = <V[i, :], dV[i, :]>   (= sum_j P[i,j] dP[i,j], the softmax-backward row term)
__global__ void rowdot_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb, size_t R,
                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t r = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    float s = A[r * lda + lane] * B[r * ldb + lane] + A[r * lda + 64 + lane] * B[r * ldb + 64 + lane];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[r] = s;
}

// ---- backward, dQ:  dQ[i, :] = s * sum_j ( P[i,j] (dP[i,j] - delta_i) + P[j,i] (dP[j,i] - delta_j) ) Q[j, :]
// with dP[i,j] = V[i] . dXs[j] (the energy uses ONE shared Q, so both the "query" and the "key" role of row i collect
// here).  A wave owns 32 rows i; per 32-row tile j: E (16 MFMAs), dP[i,j] and dP[j,i] (64 + 64, K = 128), then the
// 32x32 coefficient tile -- lane = i, registers = j, the A-operand layout -- feeds dQ += G Q[j] (16 MFMAs).
constexpr int DQ_BUF_F = 32 * QS + 2 * 32 * VS + 4 * 32;      // Q, V, dXs tiles + m, 1/l, delta, valid of the tile rows

__global__ __launch_bounds__(PA_THREADS) void attn_bwd_dq_kernel(const float* __restrict__ Q, long ldq,
                                                                 const float* __restrict__ V, long ldv,
                                                                 const float* __restrict__ dXs, long ldd, int T, int N,
                                                                 float scale, const float* __restrict__ mstat,
                                                                 const float* __restrict__ lstat,
                                                                 const float* __restrict__ delta,
                                                                 float* __restrict__ dQ, long ldo) {
    extern __shared__ __attribute__((aligned(16))) float dlds[];       // [2][DQ_BUF_F]
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int i0 = (blockIdx.y * 4 + wave) * 32;
    const size_t base = (size_t)t * N;
    const float* Qt = Q + base * ldq;
    const float* Vt = V + base * ldv;
    const float* Dt = dXs + base * ldd;
    const float c = scale * LOG2E_F;
    const int my_i = min(i0 + l31, N - 1);

    float bq[16], bv[64], bd[64];
    load_q16(Qt + (size_t)my_i * ldq, h, bq);
#pragma unroll
    for (int v = 0; v < 16; ++v) {                             // k mapping of the K = 128 products: step s, half h <-> channel 64 h + s
        const f32x4 a = *reinterpret_cast<const f32x4*>(Vt + (size_t)my_i * ldv + 64 * h + 4 * v);
        const f32x4 b = *reinterpret_cast<const f32x4*>(Dt + (size_t)my_i * ldd + 64 * h + 4 * v);
        bv[4 * v] = a[0]; bv[4 * v + 1] = a[1]; bv[4 * v + 2] = a[2]; bv[4 * v + 3] = a[3];
        bd[4 * v] = b[0]; bd[4 * v + 1] = b[1]; bd[4 * v + 2] = b[2]; bd[4 * v + 3] = b[3];
    }
    const float m_i = mstat[base + my_i], rl_i = 1.f / lstat[base + my_i], dl_i = delta[base + my_i];
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;

    auto stage = [&](int j0, float* buf) {
        {
            const int r = tid >> 3, c4 = (tid & 7) * 4;
            const int row = min(j0 + r, N - 1);
            *reinterpret_cast<f32x4*>(buf + r * QS + c4) = *reinterpret_cast<const f32x4*>(Qt + (size_t)row * ldq + c4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = k * PA_THREADS + tid, r = e >> 5, c4 = (e & 31) * 4;
            const int row = min(j0 + r, N - 1);
            *reinterpret_cast<f32x4*>(buf + 32 * QS + r * VS + c4) = *reinterpret_cast<const f32x4*>(Vt + (size_t)row * ldv + c4);
            *reinterpret_cast<f32x4*>(buf + 32 * QS + 32 * VS + r * VS + c4) = *reinterpret_cast<const f32x4*>(Dt + (size_t)row * ldd + c4);
        }
        if (tid < 32) {
            const bool ok = j0 + tid < N;
            float* st = buf + 32 * QS + 2 * 32 * VS;
            st[tid] = ok ? mstat[base + j0 + tid] : 0.f;
            st[32 + tid] = ok ? 1.f / lstat[base + j0 + tid] : 0.f;
            st[64 + tid] = ok ? delta[base + j0 + tid] : 0.f;
            st[96 + tid] = ok ? 1.f : 0.f;
        }
    };

    const int ntile = (N + 31) / 32;
    stage(0, dlds);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        if (it + 1 < ntile) stage((it + 1) * 32, dlds + ((it + 1) & 1) * DQ_BUF_F);
        const float* buf = dlds + (it & 1) * DQ_BUF_F;
        // all three products as D[m = tile row j][n = own row i]: lane = i, registers = j
        f32x16 e, p1, p2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { e[r] = 0.f; p1[r] = 0.f; p2[r] = 0.f; }
        {
            const float* qs = buf + l31 * QS + 16 * h;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(qs + 4 * v);
#pragma unroll
                for (int u = 0; u < 4; ++u) e = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], bq[4 * v + u], e, 0, 0, 0);
            }
            const float* vs = buf + 32 * QS + l31 * VS + 64 * h;            // V[j]   . dXs[i]  -> dP[j,i]
            const float* ds = vs + 32 * VS;                                   // dXs[j] . V[i]    -> dP[i,j]
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(vs + 4 * v);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(ds + 4 * v);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], bd[4 * v + u], p2, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d4[u], bv[4 * v + u], p1, 0, 0, 0);
                }
            }
        }
        // G[i,j] = s ( P[i,j] (dP[i,j] - delta_i) + P[j,i] (dP[j,i] - delta_j) ), tile rows past N masked
        const float* st = buf + 32 * QS + 2 * 32 * VS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(st + 8 * g + 4 * h);
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(st + 32 + 8 * g + 4 * h);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(st + 64 + 8 * g + 4 * h);
            const f32x4 ok4 = *reinterpret_cast<const f32x4*>(st + 96 + 8 * g + 4 * h);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float pij = __builtin_amdgcn_exp2f((e[r] - m_i) * c) * rl_i;
                const float pji = __builtin_amdgcn_exp2f((e[r] - m4[u]) * c) * r4[u];
                e[r] = ok4[u] * scale * (pij * (p1[r] - dl_i) + pji * (p2[r] - d4[u]));
            }
        }
        // dQ[i, :] += sum_j G[i,j] Q[j, :]
        const float* qrow = buf + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(e[s], qrow[mfma32_row(s, h) * QS], dq, 0, 0, 0);
    }
    if (i0 >= N) return;
    float* qo = dQ + base * ldo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + mfma32_row(r, h);
        if (i < N) qo[(size_t)i * ldo + l31] = dq[r];
    }
}

// ---- G[t, c] = max_n Y[t*N + n, c]   (first maximum on ties, as torch.max)
__global__ void segment_max_kernel(const float* __restrict__ Y, long ldy, int T, int N, int C, float* __restrict__ G,
                                   int* __restrict__ amax) {
    const int t = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    __shared__ float red[4][64];
    __shared__ int redi[4][64];
    float m = -INFINITY;
    int mi = 0;
    if (c < C)
        for (int n = rw; n < N; n += 4) {
            const float v = Y[((size_t)t * N + n) * ldy + c];
            if (v > m) { m = v; mi = n; }
        }
    red[rw][threadIdx.x & 63] = m;
    redi[rw][threadIdx.x & 63] = mi;
    __syncthreads();
    if (rw == 0 && c < C) {
        const int k = threadIdx.x;
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red[w][k] > m || (red[w][k] == m && redi[w][k] < mi)) { m = red[w][k]; mi = redi[w][k]; }
        G[(size_t)t * C + c] = m;
        if (amax) amax[(size_t)t * C + c] = mi;
    }
}

// G[t, c] = max_n act(scale_c Y[t*N + n, c] + shift_c): BatchNorm-apply + LeakyReLU(slope) folded into the point max of the widest stage
// (same per-element arithmetic as sga_bn_apply followed by segment_max_kernel, so the same numbers and the same first-maximum rule --
// but Y is read once and never rewritten)
__global__ void segment_max_affine_kernel(const float* __restrict__ Y, long ldy, int T, int N, int C, const float* __restrict__ scale,
                                          const float* __restrict__ shift, float slope, float* __restrict__ G, int* __restrict__ amax) {
    const int t = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    __shared__ float red[4][64];
    __shared__ int redi[4][64];
    float m = -INFINITY;
    int mi = 0;
    if (c < C) {
        const float sc = scale[c], sh = shift[c];
        for (int n = rw; n < N; n += 4) {
            float v = fmaf(Y[((size_t)t * N + n) * ldy + c], sc, sh);
            v = v > 0.f ? v : v * slope;
            if (v > m) { m = v; mi = n; }
        }
    }
    red[rw][threadIdx.x & 63] = m;
    redi[rw][threadIdx.x & 63] = mi;
    __syncthreads();
    if (rw == 0 && c < C) {
        const int k = threadIdx.x;
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red[w][k] > m || (red[w][k] == m && redi[w][k] < mi)) { m = red[w][k]; mi = redi[w][k]; }
        G[(size_t)t * C + c] = m;
        amax[(size_t)t * C + c] = mi;
    }
}

// backward of the point max: dY[t*N + amax[t,c], c] = dG[t,c] (dY zeroed by the caller of the kernel)
__global__ void segment_max_bwd_kernel(const float* __restrict__ dG, const int* __restrict__ amax, int T, int N, int C,
                                       float* __restrict__ dY, long ldd) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)T * C) return;
    const int t = (int)(e / C), c = (int)(e % C);
    dY[((size_t)t * N + amax[e]) * ldd + c] = dG[e];
}


// ---- backward of the encoder's widest stage, 512 -> 1024 conv + BatchNorm + LeakyReLU + max over points (pct.py:282-286,306-308),
// WITHOUT the [T*N, 1024] gradient tensors.  Only the arg-max rows carry dL/dz, and the batch-statistic terms of the BatchNorm backward
// are affine in y = cat W^T:   dL/dy[r,c] = a_c + b_c y[r,c] + (scale_c dz[t,c] at r = argmax(t,c)),  so
//     dW   = a (x) colsum(cat) + diag(b) W (cat^T cat) + sum_t coef[t,c] cat[argrow(t,c), :]
//     dcat = 1 (x) (a^T W) + cat (W^T diag(b) W) + scatter(coef[t,c] W[c, :] -> argrow(t,c))
// : a 512 x 512 Gram matrix and one [T*N,512] x [512,512] product (half the FLOPs of dY W and dY^T cat) + two sparse passes.
// head_prep: per channel (one workgroup per 64 channels, no atomics) dz at the arg-max rows from dL/dg and g (LeakyReLU' and z from the
// sign of g), x_hat there from z, S1 = sum dz, S2 = sum dz x_hat  ->  coef = scale dz,  a, b, dgamma = S2, dbeta = S1.
__global__ __launch_bounds__(256) void head_prep_kernel(const float* __restrict__ dG, const float* __restrict__ G, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ fin, int T, int C, double R,
                                                        int training, float slope, float* __restrict__ coef, float* __restrict__ ab) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rw = threadIdx.x >> 6;
    __shared__ double r1[4][64], r2[4][64];
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float ga = gamma[c], be = beta[c], sc = fin[c];
        const float ig = ga != 0.f ? 1.f / ga : __int_as_float(0x7fc00000);       // gamma == 0: x_hat is not recoverable from z -> NaN (loud)
        for (int t = rw; t < T; t += 4) {
            const float g = G[(size_t)t * C + c], dg = dG[(size_t)t * C + c];
            const float dz = g > 0.f ? dg : dg * slope;
            const float z = g > 0.f ? g : g / slope;
            coef[(size_t)t * C + c] = sc * dz;
            s1 += (double)dz;
            s2 += (double)dz * (double)((z - be) * ig);
        }
    }
    r1[rw][threadIdx.x & 63] = s1; r2[rw][threadIdx.x & 63] = s2;
    __syncthreads();
    if (rw == 0 && c < C) {
        const int k = threadIdx.x;
        const double S1 = r1[0][k] + r1[1][k] + r1[2][k] + r1[3][k], S2 = r2[0][k] + r2[1][k] + r2[2][k] + r2[3][k];
        const float sc = fin[c], mean = fin[2 * C + c], rstd = fin[3 * C + c];
        float a = 0.f, b = 0.f;
        if (training) {                       // dy = scale (dz - S1/R - x_hat S2/R),  x_hat = (y - mean) rstd
            b = (float)(-(double)sc * (double)rstd * S2 / R);
            a = (float)(-(double)sc * S1 / R) - b * mean;
        }
        ab[c] = a; ab[C + c] = b; ab[2 * C + c] = (float)S2; ab[3 * C + c] = (float)S1;
    }
}

// dW[c,:] = b_c WG[c,:] + a_c cs + sum_t coef[t,c] cat[t N + amax[t,c], :];  Wb[c,:] = b_c W[c,:].   One workgroup per channel.
__global__ __launch_bounds__(256) void head_dw_kernel(const float* __restrict__ WG, const float* __restrict__ W, const float* __restrict__ ab,
                                                      const float* __restrict__ cs, const float* __restrict__ coef, const int* __restrict__ amax,
                                                      const float* __restrict__ cat, long ldc, int T, int N, int C, int K,
                                                      float* __restrict__ dW, float* __restrict__ Wb, float* __restrict__ a0) {
    const int c = blockIdx.x;
    const float a = ab[c], b = ab[C + c];
    for (int k0 = threadIdx.x * 4; k0 < K; k0 += 256 * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < T; ++t) {
            const float cf = coef[(size_t)t * C + c];
            const f32x4 row = *reinterpret_cast<const f32x4*>(cat + ((size_t)t * N + amax[(size_t)t * C + c]) * ldc + k0);
            acc += row * cf;
        }
        const f32x4 wg = *reinterpret_cast<const f32x4*>(WG + (size_t)c * K + k0);
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)c * K + k0);
        const f32x4 csv = *reinterpret_cast<const f32x4*>(cs + k0);
        *reinterpret_cast<f32x4*>(dW + (size_t)c * K + k0) = wg * b + csv * a + acc;
        *reinterpret_cast<f32x4*>(Wb + (size_t)c * K + k0) = w * b;
        if (a != 0.f) {                                     // a0[k] = sum_c a_c W[c,k]  (caller-zeroed; C x K atomics in all)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(a0 + k0 + e, a * w[e]);
        }
    }
}

// dcat[t N + r, :] += sum over the channels c whose maximum sits in row r of coef[t,c] W[c,:].  One workgroup per object: its C channels
// are bucketed by arg-max row in LDS (counting sort), then every occupied row is summed in registers by one wave and added to dcat
// with a plain read-modify-write -- the rows of an object belong to its workgroup alone, so no atomics (the first version issued
// T*C*K = 168 M fp32 atomics per step: 0.74 ms of the 15.5).
constexpr int HS_MAXN = 1024, HS_MAXC = 1024;
__global__ __launch_bounds__(256) void head_scatter_kernel(const float* __restrict__ coef, const int* __restrict__ amax, const float* __restrict__ W,
                                                           int T, int N, int C, int K, float* __restrict__ dcat, long ldd) {
    __shared__ int cnt[HS_MAXN], start[HS_MAXN + 1], order[HS_MAXC];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = tid; r < N; r += 256) cnt[r] = 0;
    __syncthreads();
    for (int c = tid; c < C; c += 256) atomicAdd(&cnt[amax[(size_t)t * C + c]], 1);
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int r = 0; r < N; ++r) { start[r] = acc; acc += cnt[r]; } start[N] = acc; }
    __syncthreads();
    for (int r = tid; r < N; r += 256) cnt[r] = 0;                      // reused as the fill cursor
    __syncthreads();
    for (int c = tid; c < C; c += 256) { const int r = amax[(size_t)t * C + c]; order[start[r] + atomicAdd(&cnt[r], 1)] = c; }
    __syncthreads();
    for (int r = wave; r < N; r += 4) {
        const int b = start[r], e = start[r + 1];
        if (b == e) continue;                                           // wave-uniform
        float* row = dcat + ((size_t)t * N + r) * ldd;
        for (int k0 = lane * 4; k0 < K; k0 += 256) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int q = b; q < e; ++q) {
                const int c = order[q];
                acc += *reinterpret_cast<const f32x4*>(W + (size_t)c * K + k0) * coef[(size_t)t * C + c];
            }
            f32x4* dst = reinterpret_cast<f32x4*>(row + k0);
            *dst = *dst + acc;
        }
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
    hipLaunchKernelGGL(attn_apply_kernel<false>, grid, dim3(PA_THREADS), 0, s, Q, ldq, V, ldv, T, N, scale, mstat, lstat, Xs, ldx);
    SGA_CHECK_LAUNCH("sga_pct_attention");
    return SGA_OK;
}

// backward of sga_pct_attention: stats as left by the forward; work: T*N floats (delta); outputs dQ [T*N,32], dV [T*N,128]
extern "C" int sga_pct_attention_bwd(const float* Q, long ldq, const float* V, long ldv, const float* dXs, long ldd, int T,
                                     int N, const float* stats, float* work, float* dQ, long ldo, float* dV, long ldw,
                                     void* stream) {
    SGA_CHECK_ARG(Q && V && dXs && stats && work && dQ && dV, "sga_pct_attention_bwd: null pointer");
    SGA_CHECK_ARG(T >= 0 && N >= 1 && ldq >= DA && ldv >= DV && ldd >= DV && ldo >= DA && ldw >= DV, "sga_pct_attention_bwd: bad sizes");
    SGA_CHECK_ARG(ldq % 4 == 0 && ldv % 4 == 0 && ldd % 4 == 0 && reinterpret_cast<uintptr_t>(Q) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(V) % 16 == 0 && reinterpret_cast<uintptr_t>(dXs) % 16 == 0,
                  "sga_pct_attention_bwd: Q / V / dXs rows must be 16-byte aligned");
    if (T == 0) return SGA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)DA);
    const float* mstat = stats;
    const float* lstat = stats + (size_t)T * N;
    dim3 grid(T, (N + 127) / 128);
    // dV[i] = sum_j P[i,j] dXs[j]: the forward kernel with own-row statistics and dXs as the second tile
    hipLaunchKernelGGL(attn_apply_kernel<true>, grid, dim3(PA_THREADS), 0, s, Q, ldq, dXs, ldd, T, N, scale, mstat, lstat, dV, ldw);
    const size_t R = (size_t)T * N;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, V, ldv, dV, ldw, R, work);
    const size_t lds = (size_t)2 * DQ_BUF_F * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(PA_THREADS), lds, s, Q, ldq, V, ldv, dXs, ldd, T, N, scale, mstat, lstat, work, dQ, ldo);
    SGA_CHECK_LAUNCH("sga_pct_attention_bwd");
    return SGA_OK;
}

extern "C" int sga_segment_max(const float* Y, long ldy, int T, int N, int C, float* G, int32_t* argmax, void* stream) {
    SGA_CHECK_ARG(Y && G && T >= 0 && N >= 1 && C >= 1 && ldy >= C, "sga_segment_max: bad argument");
    if (T == 0) return SGA_OK;
    hipLaunchKernelGGL(segment_max_kernel, dim3(T, (C + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), Y, ldy, T, N, C, G, argmax);
    SGA_CHECK_LAUNCH("sga_segment_max");
    return SGA_OK;
}

extern "C" int sga_segment_max_bwd(const float* dG, const int32_t* argmax, int T, int N, int C, float* dY, long ldd, void* stream) {
    SGA_CHECK_ARG(dG && argmax && dY && T >= 0 && N >= 1 && C >= 1 && ldd >= C, "sga_segment_max_bwd: bad argument");
    if (T == 0) return SGA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemset2DAsync(dY, ldd * sizeof(float), 0, (size_t)C * sizeof(float), (size_t)T * N, s) != hipSuccess) { sga_set_error("sga_segment_max_bwd: memset failed"); return SGA_ERR_HIP; }
    const size_t n = (size_t)T * C;
    hipLaunchKernelGGL(segment_max_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dG, argmax, T, N, C, dY, ldd);
    SGA_CHECK_LAUNCH("sga_segment_max_bwd");
    return SGA_OK;
}

// ---- algebraic backward of conv(512->1024) + BatchNorm + LeakyReLU + point max (head_*_kernel above; ops glue in pct_ops.py) -------
extern "C" int sga_pct_head_prep(const float* dG, const float* G, const float* gamma, const float* beta, const float* fin, int T, int C,
                                 long R, int training, float slope, float* coef, float* ab, void* stream) {
    SGA_CHECK_ARG(dG && G && gamma && beta && fin && coef && ab && T >= 0 && C >= 1 && R >= 1 && slope > 0.f, "sga_pct_head_prep: bad argument");
    hipLaunchKernelGGL(head_prep_kernel, dim3((C + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), dG, G, gamma, beta, fin, T, C,
                       (double)R, training, slope, coef, ab);
    SGA_CHECK_LAUNCH("sga_pct_head_prep");
    return SGA_OK;
}

extern "C" int sga_pct_head_dw(const float* WG, const float* W, const float* ab, const float* cs, const float* coef, const int32_t* amax,
                               const float* cat, long ldc, int T, int N, int C, int K, float* dW, float* Wb, float* a0, void* stream) {
    SGA_CHECK_ARG(WG && W && ab && cs && coef && amax && cat && dW && Wb && a0 && T >= 0 && N >= 1 && C >= 1 && K >= 4 && K % 4 == 0 && ldc % 4 == 0,
                  "sga_pct_head_dw: bad argument (K and the row stride of cat must be multiples of 4)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(head_dw_kernel, dim3(C), dim3(256), 0, s, WG, W, ab, cs, coef, amax, cat, ldc, T, N, C, K, dW, Wb, a0);
    SGA_CHECK_LAUNCH("sga_pct_head_dw");
    return SGA_OK;
}

extern "C" int sga_pct_head_scatter(const float* coef, const int32_t* amax, const float* W, int T, int N, int C, int K, float* dcat, long ldd,
                                    void* stream) {
    SGA_CHECK_ARG(coef && amax && W && dcat && T >= 0 && N >= 1 && C >= 1 && K >= 4 && K % 4 == 0 && ldd >= K && ldd % 4 == 0,
                  "sga_pct_head_scatter: bad argument (K and the row stride of dcat must be multiples of 4)");
    SGA_CHECK_ARG(N <= HS_MAXN && C <= HS_MAXC, "sga_pct_head_scatter: at most %d points per object and %d channels", HS_MAXN, HS_MAXC);
    if (T == 0) return SGA_OK;
    hipLaunchKernelGGL(head_scatter_kernel, dim3(T), dim3(256), 0, static_cast<hipStream_t>(stream), coef, amax, W, T, N, C, K, dcat, ldd);
    SGA_CHECK_LAUNCH("sga_pct_head_scatter");
    return SGA_OK;
}

extern "C" int sga_segment_max_affine(const float* Y, long ldy, int T, int N, int C, const float* scale, const float* shift, float slope,
                                      float* G, int32_t* argmax, void* stream) {
    SGA_CHECK_ARG(Y && G && argmax && scale && shift && T >= 0 && N >= 1 && C >= 1 && ldy >= C, "sga_segment_max_affine: bad argument");
    if (T == 0) return SGA_OK;
    hipLaunchKernelGGL(segment_max_affine_kernel, dim3(T, (C + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), Y, ldy, T, N, C,
                       scale, shift, slope, G, argmax);
    SGA_CHECK_LAUNCH("sga_segment_max_affine");
    return SGA_OK;
}
