// loss_group = b: the contrastive (ICL) + alignment (IAL) loss evaluated on G independent groups of b consecutive subscan
// pairs -- exactly what the reference computes when its trainer feeds b pairs per iteration
// (configs/scan3r/scan3r_ground_truth.yaml:27 batch_size 2; src/aligner/losses.py:5-15,43-58,68-97,114-152) -- for ALL
// groups of a large batch in one set of launches, so a 512- or 4096-pair device batch can be compared number for number
// with the CPU reference run b pairs at a time (SURVEY.md 8d "loss grouping switch").
//
// A group is reference-sized (b in {2,4} pairs: <= ~150 anchors, <= ~400 negatives per side), so -- unlike the
// batch-global loss in contrastive.hip -- its similarity blocks ARE materialised, per group and per modality table:
//     Sg [2 na, W],  W = na + nj1 + nj2,     rows 0..na-1 = X1_g (top), rows na..2na-1 = X2_g (bottom)
//     top    row i : [ X1_i.X2_j (j < na) | X1_i.N1 | X1_i.N2 ]
//     bottom row i : [ X2_i.X1_j (= S[j,i]) | X2_i.N1 | X2_i.N2 ]
// (c2 with b = 2: 46 MB for all 256 groups x 3 tables).  The joint table is the fusion of the M tables, so its
// similarities are S_J = sum_m beta_m S_m (beta_m = w_m^2 / sum w^2, sg_aligner.py:32-34) and are derived on the fly.
//   group_sim_kernel   Z            -> Sg                                  (one workgroup per group x table)
//   group_fwd_kernel   Sg           -> sums[g][k][8], out[g][ICL_k | IALa_m | IALb_m]   (one workgroup per group)
//   group_bwd_kernel   Sg, coef     -> Sg := dL/dSg (in place), gamma[g][m] = dL/dbeta_m
//   group_grad_kernel  dL/dSg, Z    -> dZ rows of the group (every packed row belongs to exactly one group: plain stores)
// All arithmetic fp32 (fp64 only for the per-group scalar sums), VALU: the total work is O(B/b * b^2), three orders of
// magnitude below the batch-global loss, so these kernels are written for clarity, not for the MFMA roofline.
#include <stdlib.h>

#include "loss_math.h"
#include "mfma_tiles.h"

namespace {

constexpr int GL_THREADS = 256;
constexpr int GL_DP = 104;                       // packed row width (same operand layout as the fused global path)

struct GroupArgs {
    int M, A, J1, G;
    const float* Z[4];
    const float* beta;                           // [M] (M >= 2)
    const int32_t* grp;                          // [G][8]: a0, na, j1, nj1, j2, nj2, -, -
    const int64_t* soff;                         // [G+1] float offsets of the groups' Sg blocks inside one table's buffer
    int64_t stot;                                // soff[G]: per-table stride of S
    float* S;                                    // [M][stot]
    double* sums;                                // [G][NT][8]
    double* out;                                 // [G][NT + 2M]
    const float* coef;                           // [G][NT + 2M]
    double* gamma;                               // [G][M]
    float* dZ[4];
    float alpha, kc, ki, itc, iti;
};

struct Grp { int a0, na, j1, nj1, j2, nj2, W; };
__device__ __forceinline__ Grp load_grp(const int32_t* g) {
    Grp r{g[0], g[1], g[2], g[3], g[4], g[5], 0};
    r.W = r.na + r.nj1 + r.nj2;
    return r;
}
// packed Z row of similarity-block row r (0..2na-1) / column c (0..W-1) of a group
__device__ __forceinline__ int row_z(const Grp& g, int A, int r) { return r < g.na ? g.a0 + r : A + g.a0 + (r - g.na); }
__device__ __forceinline__ int col_z(const Grp& g, int A, int J1, int half, int c) {
    if (c < g.na) return half ? g.a0 + c : A + g.a0 + c;            // top rows meet X2, bottom rows meet X1
    if (c < g.na + g.nj1) return 2 * A + g.j1 + (c - g.na);
    return 2 * A + J1 + g.j2 + (c - g.na - g.nj1);
}
// sum family of a negatives column for a row half: s11, s12 (top) / s22, s21 (bottom) -- contrastive.hip fill_groups
__device__ __forceinline__ int fam_of(const Grp& g, int half, int c) {
    const bool n1 = c < g.na + g.nj1;
    return half ? (n1 ? 3 : 2) : (n1 ? 0 : 1);
}

__global__ __launch_bounds__(GL_THREADS) void group_sim_kernel(GroupArgs a) {
    const int gi = blockIdx.x, m = blockIdx.y;
    const Grp g = load_grp(a.grp + gi * 8);
    const float* __restrict__ Z = a.Z[m];
    float* __restrict__ S = a.S + (size_t)m * a.stot + a.soff[gi];
    const int n = 2 * g.na * g.W;
    for (int e = threadIdx.x; e < n; e += GL_THREADS) {
        const int r = e / g.W, c = e - r * g.W;
        const f32x4* x = reinterpret_cast<const f32x4*>(Z + (size_t)row_z(g, a.A, r) * GL_DP);
        const f32x4* y = reinterpret_cast<const f32x4*>(Z + (size_t)col_z(g, a.A, a.J1, r >= g.na, c) * GL_DP);
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < GL_DP / 4; ++q) {
            const f32x4 u = x[q], v = y[q];
            acc = fmaf(u[0], v[0], acc); acc = fmaf(u[1], v[1], acc); acc = fmaf(u[2], v[2], acc); acc = fmaf(u[3], v[3], acc);
        }
        S[e] = acc;
    }
}

// block-wide sum of per-thread partials into an LDS double array (slot e), NV values per thread
template <int NV>
__device__ __forceinline__ void block_accumulate(const float (&v)[NV], double* lds_acc) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const double s = wave_sum_d((double)v[e]);
        if (lane == 0) atomicAdd(lds_acc + e, s);
    }
}

template <int M>
__global__ __launch_bounds__(GL_THREADS) void group_fwd_kernel(GroupArgs a) {
    constexpr int NT = M > 1 ? M + 1 : 1, NO = NT + (M > 1 ? 2 * M : 0);
    __shared__ double s_sum[NT * 8];
    __shared__ double s_out[NO];
    __shared__ float s_inv[NT * 8];
    const int gi = blockIdx.x, tid = threadIdx.x;
    const Grp g = load_grp(a.grp + gi * 8);
    const float* __restrict__ S = a.S + a.soff[gi];
    for (int e = tid; e < NT * 8; e += GL_THREADS) s_sum[e] = 0.0;
    for (int e = tid; e < NO; e += GL_THREADS) s_out[e] = 0.0;
    __syncthreads();
    float beta[M];
#pragma unroll
    for (int m = 0; m < M; ++m) beta[m] = M > 1 ? a.beta[m] : 1.f;

    // ---- pass 1: the 4 x 2 sums of every table over this group's anchors x negatives blocks (losses.py:10-11)
    {
        float p[NT * 8];
#pragma unroll
        for (int e = 0; e < NT * 8; ++e) p[e] = 0.f;
        const int wn = g.nj1 + g.nj2, n = 2 * g.na * wn;
        for (int e = tid; e < n; e += GL_THREADS) {
            const int r = e / wn, c = g.na + (e - r * wn);
            const int fam = fam_of(g, r >= g.na, c);
            float sj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float s = S[(size_t)m * a.stot + (size_t)r * g.W + c];
                sj = fmaf(beta[m], s, sj);
                const float e0 = fexp2(s * a.kc), e1 = fexp2(s * a.ki);
#pragma unroll
                for (int f = 0; f < 4; ++f) { p[m * 8 + f * 2] += f == fam ? e0 : 0.f; p[m * 8 + f * 2 + 1] += f == fam ? e1 : 0.f; }
            }
            if (M > 1) {
                const float e0 = fexp2(sj * a.kc), e1 = fexp2(sj * a.ki);
#pragma unroll
                for (int f = 0; f < 4; ++f) { p[(NT - 1) * 8 + f * 2] += f == fam ? e0 : 0.f; p[(NT - 1) * 8 + f * 2 + 1] += f == fam ? e1 : 0.f; }
            }
        }
        block_accumulate<NT * 8>(p, s_sum);
    }
    __syncthreads();
    for (int e = tid; e < NT * 8; e += GL_THREADS) {
        a.sums[(size_t)gi * NT * 8 + e] = s_sum[e];
        s_inv[e] = (float)(1.0 / (s_sum[e] + 1e-9));
    }
    __syncthreads();

    // ---- pass 2: anchors x anchors terms (losses.py:12-15, 51-57, 84-95)
    {
        float o[NO];
#pragma unroll
        for (int e = 0; e < NO; ++e) o[e] = 0.f;
        const int n = g.na * g.na;
        const float* js = s_inv + (NT - 1) * 8;
        for (int e = tid; e < n; e += GL_THREADS) {
            const int i = e / g.na, j = e - i * g.na;
            float xs[M], ys[M], xj = 0.f, yj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                xs[m] = S[(size_t)m * a.stot + (size_t)i * g.W + j];                 // S_m[i,j]
                ys[m] = S[(size_t)m * a.stot + (size_t)(g.na + i) * g.W + j];        // S_m[j,i]
                xj = fmaf(beta[m], xs[m], xj); yj = fmaf(beta[m], ys[m], yj);
            }
            float lqma = 0.f, lqmb = 0.f;
            if (M > 1) {
                const float dji = fexp2(xj * a.ki);
                lqma = flog(g_val(dji, js[1], js[3]));
                lqmb = flog(g_val(dji, js[5], js[7]));
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const float x = k < M ? xs[k < M ? k : 0] : xj, y = k < M ? ys[k < M ? k : 0] : yj;
                const float* is = s_inv + k * 8;
                const float qa = g_val(fexp2(x * a.kc), is[0], is[2]);
                const float qb = g_val(fexp2(y * a.kc), is[4], is[6]);
                o[k] -= flog(a.alpha * qa + (1.f - a.alpha) * qb);
                if (M > 1 && k < M) {
                    const float dm = fexp2(x * a.ki);
                    const float qoa = g_val(dm, is[1], is[3]), qob = g_val(dm, is[5], is[7]);
                    o[NT + (k < M ? k : 0)] += __expf(qoa) * (qoa - lqma);
                    o[NT + M + (k < M ? k : 0)] += __expf(qob) * (qob - lqmb);
                }
            }
        }
        block_accumulate<NO>(o, s_out);
    }
    __syncthreads();
    for (int e = tid; e < NO; e += GL_THREADS) a.out[(size_t)gi * NO + e] = s_out[e];
}

template <int M>
__global__ __launch_bounds__(GL_THREADS) void group_bwd_kernel(GroupArgs a) {
    constexpr int NT = M > 1 ? M + 1 : 1, NO = NT + (M > 1 ? 2 * M : 0);
    __shared__ double s_gs[NT * 8];
    __shared__ double s_gam[M];
    __shared__ float s_inv[NT * 8];
    __shared__ float s_c[NT * 8];                  // dL/d(sums) * 1/tau, per (table, family, temperature)
    const int gi = blockIdx.x, tid = threadIdx.x;
    const Grp g = load_grp(a.grp + gi * 8);
    float* __restrict__ S = a.S + a.soff[gi];
    for (int e = tid; e < NT * 8; e += GL_THREADS) {
        s_gs[e] = 0.0;
        s_inv[e] = (float)(1.0 / (a.sums[(size_t)gi * NT * 8 + e] + 1e-9));
    }
    for (int e = tid; e < M; e += GL_THREADS) s_gam[e] = 0.0;
    __syncthreads();
    float beta[M];
#pragma unroll
    for (int m = 0; m < M; ++m) beta[m] = M > 1 ? a.beta[m] : 1.f;
    const float* coef = a.coef + (size_t)gi * NO;
    float gam[M];
#pragma unroll
    for (int m = 0; m < M; ++m) gam[m] = 0.f;

    // ---- anchors x anchors: G_m[i,j] = dL/dS_m[i,j] (both roles of S[i,j]: the x of term (i,j), the y of term (j,i)) + beta_m dL/dS_J[i,j];
    //      written over the top-left block, the bottom-left (transposed duplicate) block becomes 0.  Same algebra as
    //      contrastive.hip anchor_multi_bwd16_kernel.
    {
        float ags[NT * 8];
#pragma unroll
        for (int e = 0; e < NT * 8; ++e) ags[e] = 0.f;
        const int n = g.na * g.na;
        const float* js = s_inv + (NT - 1) * 8;
        const float al = a.alpha, be = 1.f - a.alpha;
        for (int e = tid; e < n; e += GL_THREADS) {
            const int i = e / g.na, j = e - i * g.na;
            float xs[M], ys[M], xj = 0.f, yj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                xs[m] = S[(size_t)m * a.stot + (size_t)i * g.W + j];
                ys[m] = S[(size_t)m * a.stot + (size_t)(g.na + i) * g.W + j];
                xj = fmaf(beta[m], xs[m], xj); yj = fmaf(beta[m], ys[m], yj);
            }
            float gJ = 0.f, EA = 0.f, EB = 0.f, lqma = 0.f, lqmb = 0.f, dji = 0.f;
            GP MA{}, MB{};
            if (M > 1) {
                const float cJ = coef[NT - 1];
                const float dx = fexp2(xj * a.kc), dy = fexp2(yj * a.kc);
                const GP Ax = g_parts(dx, js[0], js[2]), Bx = g_parts(dx, js[4], js[6]);
                const float qAy = g_val(dy, js[0], js[2]), qBy = g_val(dy, js[4], js[6]);
                const float wA = (-cJ * al) * frcp(al * Ax.q + be * qBy) * dx;
                const float wB = (-cJ * be) * frcp(al * qAy + be * Bx.q) * dx;
                gJ = fmaf(wA, Ax.dd, wB * Bx.dd) * a.itc;
                float* q = ags + (NT - 1) * 8;
                q[0] = fmaf(wA, Ax.p, q[0]); q[2] = fmaf(wA, Ax.r, q[2]); q[4] = fmaf(wB, Bx.p, q[4]); q[6] = fmaf(wB, Bx.r, q[6]);
                dji = fexp2(xj * a.ki);
                MA = g_parts(dji, js[1], js[3]); MB = g_parts(dji, js[5], js[7]);
                lqma = flog(MA.q); lqmb = flog(MB.q);
            }
            float gx[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float* is = s_inv + m * 8;
                const float c = coef[m];
                const float x = xs[m], y = ys[m];
                const float dx = fexp2(x * a.kc), dy = fexp2(y * a.kc);
                const GP Ax = g_parts(dx, is[0], is[2]), Bx = g_parts(dx, is[4], is[6]);
                const float qAy = g_val(dy, is[0], is[2]), qBy = g_val(dy, is[4], is[6]);
                const float wA = (-c * al) * frcp(al * Ax.q + be * qBy) * dx;
                const float wB = (-c * be) * frcp(al * qAy + be * Bx.q) * dx;
                float gxm = fmaf(wA, Ax.dd, wB * Bx.dd) * a.itc;
                float* q = ags + m * 8;
                q[0] = fmaf(wA, Ax.p, q[0]); q[2] = fmaf(wA, Ax.r, q[2]); q[4] = fmaf(wB, Bx.p, q[4]); q[6] = fmaf(wB, Bx.r, q[6]);
                if (M > 1) {
                    const float ca = coef[NT + m], cb = coef[NT + M + m];
                    const float dm = fexp2(x * a.ki);
                    const GP OA = g_parts(dm, is[1], is[3]), OB = g_parts(dm, is[5], is[7]);
                    const float eA = ca * __expf(OA.q), eB = cb * __expf(OB.q);
                    const float tA = eA * (OA.q - lqma + 1.f) * dm, tB = eB * (OB.q - lqmb + 1.f) * dm;
                    gxm = fmaf(fmaf(tA, OA.dd, tB * OB.dd), a.iti, gxm);
                    q[1] = fmaf(tA, OA.p, q[1]); q[3] = fmaf(tA, OA.r, q[3]); q[5] = fmaf(tB, OB.p, q[5]); q[7] = fmaf(tB, OB.r, q[7]);
                    EA += eA; EB += eB;
                }
                gx[m] = gxm;
            }
            if (M > 1) {
                const float uA = -EA * frcp(MA.q) * dji, uB = -EB * frcp(MB.q) * dji;
                gJ = fmaf(fmaf(uA, MA.dd, uB * MB.dd), a.iti, gJ);
                float* q = ags + (NT - 1) * 8;
                q[1] = fmaf(uA, MA.p, q[1]); q[3] = fmaf(uA, MA.r, q[3]); q[5] = fmaf(uB, MB.p, q[5]); q[7] = fmaf(uB, MB.r, q[7]);
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                gam[m] = fmaf(gJ, xs[m], gam[m]);
                S[(size_t)m * a.stot + (size_t)i * g.W + j] = fmaf(beta[m], gJ, gx[m]);
                S[(size_t)m * a.stot + (size_t)(g.na + i) * g.W + j] = 0.f;
            }
        }
        // dg/dsum = -d inv^2 (g/u)^2: the uniform factor -inv^2 is applied once here
#pragma unroll
        for (int e = 0; e < NT * 8; ++e) ags[e] *= -s_inv[e] * s_inv[e];
        block_accumulate<NT * 8>(ags, s_gs);
    }
    __syncthreads();
    for (int e = tid; e < NT * 8; e += GL_THREADS) s_c[e] = (float)(s_gs[e] * (double)((e & 1) ? a.iti : a.itc));
    __syncthreads();

    // ---- anchors x negatives: dL/dU = sum_temp dL/dsum * exp(U/tau)/tau (the sums' backward), joint share folded in
    {
        const int wn = g.nj1 + g.nj2, n = 2 * g.na * wn;
        for (int e = tid; e < n; e += GL_THREADS) {
            const int r = e / wn, c = g.na + (e - r * wn);
            const int fam = fam_of(g, r >= g.na, c);
            float s[M], sj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) { s[m] = S[(size_t)m * a.stot + (size_t)r * g.W + c]; sj = fmaf(beta[m], s[m], sj); }
            float cj = 0.f;
            if (M > 1) cj = s_c[(NT - 1) * 8 + fam * 2] * fexp2(sj * a.kc) + s_c[(NT - 1) * 8 + fam * 2 + 1] * fexp2(sj * a.ki);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float cm = s_c[m * 8 + fam * 2] * fexp2(s[m] * a.kc) + s_c[m * 8 + fam * 2 + 1] * fexp2(s[m] * a.ki);
                gam[m] = fmaf(cj, s[m], gam[m]);
                S[(size_t)m * a.stot + (size_t)r * g.W + c] = fmaf(beta[m], cj, cm);
            }
        }
    }
    if (M > 1) {
        block_accumulate<M>(gam, s_gam);
        __syncthreads();
        for (int e = tid; e < M; e += GL_THREADS) a.gamma[(size_t)gi * M + e] = s_gam[e];
    }
}

// dZ rows of a group from its coefficient blocks C = dL/dSg:  every packed row (X1_i, X2_i, N1_j, N2_j) belongs to exactly
// one group, so each output element has one writer.  Thread = (output-row slot, column d); coefficients are wave-uniform loads.
__global__ __launch_bounds__(GL_THREADS) void group_grad_kernel(GroupArgs a) {
    const int gi = blockIdx.x, m = blockIdx.y;
    const Grp g = load_grp(a.grp + gi * 8);
    const float* __restrict__ Z = a.Z[m];
    const float* __restrict__ C = a.S + (size_t)m * a.stot + a.soff[gi];
    float* __restrict__ dZ = a.dZ[m];
    const int d = threadIdx.x & 127, slot = threadIdx.x >> 7;
    if (d >= GL_DP) return;
    const int rows = 2 * g.na + g.nj1 + g.nj2;
    for (int o = slot; o < rows; o += GL_THREADS / 128) {
        float acc = 0.f;
        int zrow;
        if (o < 2 * g.na) {                       // anchor rows: "row role" over all their columns ...
            const int half = o >= g.na;
            zrow = row_z(g, a.A, o);
            const float* crow = C + (size_t)o * g.W;
            for (int c = half ? g.na : 0; c < g.W; ++c)          // (bottom-left block is zero by construction)
                acc = fmaf(crow[c], Z[(size_t)col_z(g, a.A, a.J1, half, c) * GL_DP + d], acc);
            if (half) {                           // ... X2_i is also the COLUMN i of the top-left block: dX2_i += sum_r G[r,i] X1_r
                const int i = o - g.na;
                for (int r = 0; r < g.na; ++r) acc = fmaf(C[(size_t)r * g.W + i], Z[(size_t)(g.a0 + r) * GL_DP + d], acc);
            }
        } else {                                  // negatives: "column role" against both anchor halves
            const int c = g.na + (o - 2 * g.na);
            zrow = col_z(g, a.A, a.J1, 0, c);
            for (int r = 0; r < 2 * g.na; ++r) acc = fmaf(C[(size_t)r * g.W + c], Z[(size_t)row_z(g, a.A, r) * GL_DP + d], acc);
        }
        dZ[(size_t)zrow * GL_DP + d] += acc;
    }
}

int fill_group_args(GroupArgs& a, const float* const* Z, int M, const float* beta, int A, int J1, const int32_t* groups, int G,
                    const int64_t* soff, int64_t stot, float alpha, float tau_icl, float tau_ial, float* S, const char* who) {
    if (M < 1 || M > 4) { sga_set_error("%s: M=%d outside [1,4]", who, M); return SGA_ERR_ARG; }
    if (!Z || !groups || !soff || !S || (M > 1 && !beta) || A < 0 || J1 < 0 || G < 0 || stot < 0 || !(tau_icl > 0) || !(tau_ial > 0)) {
        sga_set_error("%s: bad argument", who);
        return SGA_ERR_ARG;
    }
    a.M = M; a.A = A; a.J1 = J1; a.G = G; a.beta = beta; a.grp = groups; a.soff = soff; a.stot = stot; a.S = S; a.alpha = alpha;
    a.kc = LOG2E / tau_icl; a.ki = LOG2E / tau_ial; a.itc = 1.f / tau_icl; a.iti = 1.f / tau_ial;
    for (int m = 0; m < M; ++m) { if (!Z[m]) { sga_set_error("%s: null table %d", who, m); return SGA_ERR_ARG; } a.Z[m] = Z[m]; }
    return SGA_OK;
}

// group_sim on the matrix cores (15.6 ms per step for the VALU kernel at b = 16 on a 512-pair batch): a wave owns a 32 x 32 tile of a
// group's Sg (top and bottom halves tiled separately: their first na columns meet different anchor sets), both operands are Z rows
// read as float4 straight from L2 (k = 8q + 4h + r: one b128 per operand feeds 4 MFMAs), rows of Sg are stored 128 bytes at a time.
__global__ __launch_bounds__(GL_THREADS) void group_sim_mfma_kernel(GroupArgs a) {
    const int gi = blockIdx.x, m = blockIdx.y;
    const Grp g = load_grp(a.grp + gi * 8);
    const float* __restrict__ Z = a.Z[m];
    float* __restrict__ S = a.S + (size_t)m * a.stot + a.soff[gi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int na = g.na, W = g.W;
    const int tR = (na + 31) >> 5, tC = (W + 31) >> 5, ntile = 2 * tR * tC;
    for (int tile = blockIdx.z * (GL_THREADS / 64) + wave; tile < ntile; tile += gridDim.z * (GL_THREADS / 64)) {
        const int half = tile >= tR * tC, t2 = tile - half * tR * tC;
        const int r0 = (t2 / tC) * 32, c0 = (t2 % tC) * 32;
        const int ri = min(r0 + l31, na - 1), cj = min(c0 + l31, W - 1);               // clamped: rows / columns past the end are not stored
        const float* __restrict__ zr = Z + (size_t)row_z(g, a.A, half * na + ri) * GL_DP + 4 * h;
        const float* __restrict__ zc = Z + (size_t)col_z(g, a.A, a.J1, half, cj) * GL_DP + 4 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < GL_DP / 8; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(zr + 8 * q);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(zc + 8 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bv[r], acc, 0, 0, 0);
        }
        if (c0 + l31 < W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = r0 + mfma32_row(r, h);
                if (i < na) S[(size_t)(half * na + i) * W + c0 + l31] = acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// group_grad on the matrix cores.  The VALU kernel above walks every (row, column) of a group's dL/dSg with one thread per embedding
// column: fine for b = 2 (it was written for that), 13 ms per step at b = 4 and 62 ms at b = 8 on a 512-pair batch -- more than the
// whole batch-global loss.  Here a wave owns a 32-row tile of the group's output rows and accumulates its 32 x 104 block of dZ in
// four 32x32 MFMA accumulators; the coefficient is the A operand straight from the dL/dSg block (row-major for the "row role" of the
// anchors, transposed -- lane = output row, coalesced -- for the "column role" of X2 and of the negatives), the gathered Z rows are
// the B operand (lane = embedding column: 128-byte reads).  No LDS, everything comes from L2: a group's blocks are <= ~1 MB.
//   X1 tile (rows i):      dX1_i  = sum_c G[i, c] Zcol_top(c)                                   c in [0, W)
//   X2 tile (rows i):      dX2_i  = sum_{c >= na} G[na + i, c] Zcol_bot(c)  +  sum_r G[r, i] X1_r   r in [0, na)
//   negatives tile (c):    dN_c   = sum_r G[r, na + c] Zrow(r)                                   r in [0, 2 na)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GL_THREADS) void group_grad_mfma_kernel(GroupArgs a) {
    const int gi = blockIdx.x, m = blockIdx.y;
    const Grp g = load_grp(a.grp + gi * 8);
    const float* __restrict__ Z = a.Z[m];
    const float* __restrict__ C = a.S + (size_t)m * a.stot + a.soff[gi];
    float* __restrict__ dZ = a.dZ[m];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int na = g.na, W = g.W, nneg = g.nj1 + g.nj2;
    const int tA = (na + 31) >> 5, tN = (nneg + 31) >> 5, ntile = 2 * tA + tN;
    for (int tile = blockIdx.z * (GL_THREADS / 64) + wave; tile < ntile; tile += gridDim.z * (GL_THREADS / 64)) {
        f32x16 acc[4];
        zero_acc<4>(acc);
        const int cls = tile < tA ? 0 : (tile < 2 * tA ? 1 : 2);
        const int o0 = cls == 0 ? tile * 32 : (cls == 1 ? (tile - tA) * 32 : (tile - 2 * tA) * 32);   // first row of the tile within its class
        const int nvalid = (cls == 2 ? nneg : na) - o0;                                                  // rows of the tile that exist (>= 1)
        const bool rv = l31 < nvalid;
        // ---- row-role segment (A row-major: lane's row of G, k along the row)
        if (cls <= 1) {
            const int grow = (cls == 0 ? 0 : na) + o0 + (rv ? l31 : 0);
            const float* __restrict__ crow = C + (size_t)grow * W;
            for (int k0 = cls == 0 ? 0 : na; k0 < W; k0 += 2) {
                const int k = k0 + h;
                const bool kv = k < W;
                const float av = (rv && kv) ? crow[k] : 0.f;
                const float* __restrict__ zr = Z + (size_t)col_z(g, a.A, a.J1, cls, kv ? k : W - 1) * GL_DP;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int d = ct * 32 + l31;
                    const float bv = (kv && d < GL_DP) ? zr[d] : 0.f;
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ct], 0, 0, 0);
                }
            }
        }
        // ---- column-role segment (A transposed: k = a row r of G, lane = this tile's column)
        if (cls >= 1) {
            const int gcol = (cls == 1 ? 0 : na) + o0 + (rv ? l31 : 0);
            const int kend = cls == 1 ? na : 2 * na;
            for (int k0 = 0; k0 < kend; k0 += 2) {
                const int r = k0 + h;
                const bool kv = r < kend;
                const float av = (rv && kv) ? C[(size_t)r * W + gcol] : 0.f;
                const float* __restrict__ zr = Z + (size_t)row_z(g, a.A, kv ? r : 0) * GL_DP;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int d = ct * 32 + l31;
                    const float bv = (kv && d < GL_DP) ? zr[d] : 0.f;
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ct], 0, 0, 0);
                }
            }
        }
        // ---- store: acc[ct][r] = dZ[row(r,h) of the tile][ct*32 + lane&31]; every packed row belongs to exactly one tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = mfma32_row(r, h);
            if (i >= nvalid) continue;
            const int zrow = cls == 0 ? g.a0 + o0 + i : (cls == 1 ? a.A + g.a0 + o0 + i : col_z(g, a.A, a.J1, 0, na + o0 + i));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int d = ct * 32 + l31;
                if (d < GL_DP) dZ[(size_t)zrow * GL_DP + d] += acc[ct][r];
            }
        }
    }
}

}  // namespace

extern "C" int sga_group_loss_fwd(const float* const* Z, int M, const float* beta, int A, int J1, const int32_t* groups, int G,
                                  const int64_t* s_off, int64_t s_total, float alpha, float tau_icl, float tau_ial, float* S,
                                  double* sums, double* out, int use_valu, void* stream) {
    if (G == 0) return SGA_OK;
    GroupArgs a{};
    if (int rc = fill_group_args(a, Z, M, beta, A, J1, groups, G, s_off, s_total, alpha, tau_icl, tau_ial, S, "sga_group_loss_fwd")) return rc;
    SGA_CHECK_ARG(sums && out, "sga_group_loss_fwd: null output");
    a.sums = sums; a.out = out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_valu) {
        hipLaunchKernelGGL(group_sim_kernel, dim3(G, M), dim3(GL_THREADS), 0, s, a);
    } else {
        int gz = (8 * sga_num_cus()) / (G * M > 0 ? G * M : 1);
        if (gz < 1) gz = 1;
        if (gz > 16) gz = 16;
        hipLaunchKernelGGL(group_sim_mfma_kernel, dim3(G, M, gz), dim3(GL_THREADS), 0, s, a);
    }
    if (M == 1) hipLaunchKernelGGL(group_fwd_kernel<1>, dim3(G), dim3(GL_THREADS), 0, s, a);
    else if (M == 2) hipLaunchKernelGGL(group_fwd_kernel<2>, dim3(G), dim3(GL_THREADS), 0, s, a);
    else if (M == 3) hipLaunchKernelGGL(group_fwd_kernel<3>, dim3(G), dim3(GL_THREADS), 0, s, a);
    else hipLaunchKernelGGL(group_fwd_kernel<4>, dim3(G), dim3(GL_THREADS), 0, s, a);
    SGA_CHECK_LAUNCH("sga_group_loss_fwd");
    return SGA_OK;
}

extern "C" int sga_group_loss_bwd(const float* const* Z, int M, const float* beta, int A, int J1, const int32_t* groups, int G,
                                  const int64_t* s_off, int64_t s_total, float alpha, float tau_icl, float tau_ial, float* S,
                                  const double* sums, const float* coef, float* const* dZ, double* gamma, int use_valu, void* stream) {
    if (G == 0) return SGA_OK;
    GroupArgs a{};
    if (int rc = fill_group_args(a, Z, M, beta, A, J1, groups, G, s_off, s_total, alpha, tau_icl, tau_ial, S, "sga_group_loss_bwd")) return rc;
    SGA_CHECK_ARG(sums && coef && dZ && (M == 1 || gamma), "sga_group_loss_bwd: null argument");
    a.sums = const_cast<double*>(sums); a.coef = coef; a.gamma = gamma;
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(dZ[m], "sga_group_loss_bwd: null dZ"); a.dZ[m] = dZ[m]; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (M == 1) hipLaunchKernelGGL(group_bwd_kernel<1>, dim3(G), dim3(GL_THREADS), 0, s, a);
    else if (M == 2) hipLaunchKernelGGL(group_bwd_kernel<2>, dim3(G), dim3(GL_THREADS), 0, s, a);
    else if (M == 3) hipLaunchKernelGGL(group_bwd_kernel<3>, dim3(G), dim3(GL_THREADS), 0, s, a);
    else hipLaunchKernelGGL(group_bwd_kernel<4>, dim3(G), dim3(GL_THREADS), 0, s, a);
    if (use_valu) {
        hipLaunchKernelGGL(group_grad_kernel, dim3(G, M), dim3(GL_THREADS), 0, s, a);
    } else {
        int gz = (8 * sga_num_cus()) / (G * M > 0 ? G * M : 1);          // tiles of a group are spread over gz workgroups of 4 waves
        if (gz < 1) gz = 1;
        if (gz > 16) gz = 16;
        hipLaunchKernelGGL(group_grad_mfma_kernel, dim3(G, M, gz), dim3(GL_THREADS), 0, s, a);
    }
    SGA_CHECK_LAUNCH("sga_group_loss_bwd");
    return SGA_OK;
}
