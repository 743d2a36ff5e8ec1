// Dense node-similarity + contrastive (ICL) / alignment (IAL) loss, forward and backward, without
// ever materialising an anchors x negatives matrix.
//
// Replaces reference src/aligner/losses.py: calculate_prob_dist :5-15, ICLLoss.forward :43-58,
// IALLoss.forward :68-97 (and their autograd).  For each embedding table k (modalities + 'joint')
// with L2-normalised rows and index sets e1i/e2i (A anchors each), e1j (J1), e2j (J2):
//     X1 = E[e1i], X2 = E[e2i], N1 = E[e1j], N2 = E[e2j]            (packed row blocks of Z_k)
//     s11 = sum exp(X1 N1^T/t)  s12 = sum exp(X1 N2^T/t)  s22 = sum exp(X2 N2^T/t)  s21 = sum exp(X2 N1^T/t)
//     S = X1 X2^T ;  qA[i,j] = g(exp(S[i,j]/t); s11, s12) ;  qB[i,j] = g(exp(S[j,i]/t); s22, s21)
//     g(d; sa, sb) = 1 / (1 + 1/(d/(sa+1e-9)+1e-9) + 1/(d/(sb+1e-9)+1e-9) + 1e-9)
//     ICL_k  = sum_ij -log(a qA + (1-a) qB)                         (t = 0.1; the mean's 1/A^2 is applied by the host)
//     IALa_m = sum_ij exp(qoA)(qoA - log qmA), IALb_m likewise with qB   (t = 1; qo from table m, qm from 'joint')
//
// Kernel set (all exact fp32 on v_mfma_f32_32x32x2_f32, fp64 only for the global scalar sums):
//   gather      E, idx            -> Z (normalised, K padded to a multiple of 8), row norms
//   sweep<sum>  Z                 -> the 4x2 global sums per table           (anchors x negatives, pass 1)
//   anchor<fwd> Z (all tables)    -> ICL / IALa / IALb sums                  (anchors x anchors,  pass 2)
//   anchor<bwd> Z, upstream coefs -> dL/dS stash (transposed) + dL/d(sums)   (anchors x anchors)
//   (sga_gemm)  stash, Z          -> dZ anchor rows
//   sweep<grad> Z, dL/d(sums)     -> dZ += coefficient-weighted negatives    (owner-stationary, two sweeps)
//   scatter     dZ, Z, norms, idx -> dE (normalisation Jacobian + index_add)
// S tiles are produced in the orientation "lane = owner row, registers = other rows", which is also
// the MFMA A-operand layout, so the gradient GEMM (coefficients x other rows) chains straight from the
// accumulators with no LDS transpose and each owner row is accumulated by exactly one wave.
#include "mfma_tiles.h"
#include <type_traits>

#include "loss_math.h"
#include "wide16_api.h"

// gemm.hip (include/sgaligner_hip.h): the stash gradient of the anchors x anchors backward runs on the GEMM kernels
extern "C" int sga_gemm(int transA, int transB, int M, int N, int K, const void* A, long lda, int a_is_f64, const float* B,
                        long ldb, float* C, long ldc, const float* bias, int accumulate, void* stream);

namespace {

constexpr int CT_THREADS = 256;
constexpr int CT_MAXT = 9;            // modalities (<= 8) + joint
// ------------------------------------------------------------------------------------------------
// gather + normalise:  Z[r, :] = E[idx[r], :] / max(||.||, 1e-12), zero padded to Dp; nrm[r] = ||.||
// (F.normalize(emb, dim=1) then emb[data_dict[...]]: losses.py:44-48, :73-79, :84-87)
// ------------------------------------------------------------------------------------------------
__global__ void gather_normalize_kernel(const float* __restrict__ E, int D, const int* __restrict__ idx, int R,
                                        float* __restrict__ Z, int Dp, float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
        const float* x = E + (size_t)idx[r] * D;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = x[d]; ss += v * v; }
        ss = wave_sum(ss);
        const float n = sqrtf(ss);
        const float inv = 1.f / fmaxf(n, 1e-12f);
        float* z = Z + (size_t)r * Dp;
        for (int d = lane; d < Dp; d += 64) z[d] = d < D ? x[d] * inv : 0.f;
        if (lane == 0) nrm[r] = n;
    }
}

// dE[idx[r], :] += J_normalize^T dZ[r, :]
__global__ void scatter_normalize_bwd_kernel(const float* __restrict__ dZ, const float* __restrict__ Z,
                                             const float* __restrict__ nrm, const int* __restrict__ idx, int R, int D,
                                             int Dp, float* __restrict__ dE) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
        const float* g = dZ + (size_t)r * Dp;
        const float* z = Z + (size_t)r * Dp;
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += g[d] * z[d];
        dot = wave_sum(dot);
        const float n = nrm[r];
        const bool clamped = n < 1e-12f;
        const float inv = 1.f / fmaxf(n, 1e-12f);
        if (clamped) dot = 0.f;
        float* o = dE + (size_t)idx[r] * D;
        for (int d = lane; d < D; d += 64) atomicAdd(o + d, (g[d] - z[d] * dot) * inv);
    }
}

// ------------------------------------------------------------------------------------------------
// owner-stationary sweeps over (owner rows) x (other rows): pass-1 sums and the negatives' gradient
// ------------------------------------------------------------------------------------------------
struct SweepSeg { int row0, n, fam; };                 // other rows [row0, row0+n), sum family 0..3
struct SweepGroup { int own0, nown, blk0, nseg; SweepSeg seg[2]; int nsplit; };   // nsplit: multi kernel only
struct SweepArgs {
    const float* Z; int Dp; int ngroups; SweepGroup grp[4];
    float k0, k1;                   // log2(e)/tau for the two temperatures
    float it0, it1;                 // 1/tau
    double* sums;                   // [8]  (fam*2 + temp)            (SUM mode: output)
    const double* gs;               // [8]  dL/d(sums)                (GRAD mode: input)
    float* dZ;                      // [R][Dp]                        (GRAD mode: atomic accumulate)
    int col0;                       // first gradient column of this pass (GRAD, Dp > NCT*32)
};

template <int NJT, int NCT, bool GRAD>
__global__ __launch_bounds__(CT_THREADS) void sweep_kernel(SweepArgs a) {
    constexpr int OT = NJT * 32;                      // other rows per step
    constexpr int GW = NCT * 32;                      // gradient columns per pass
    extern __shared__ __attribute__((aligned(16))) float lds[];    // max(S chunks, gradient tile): sweep_lds_bytes()
    float* own_s = lds;
    float* oth_s = lds + 128 * SGA_LDS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const SweepGroup& grp = a.grp[g];
    const int own0 = grp.own0 + ((int)blockIdx.x - grp.blk0) * 128;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + wave * 32 + (lane & 31);

    f32x16 gacc[GRAD ? NCT : 1];
    if (GRAD) zero_acc<GRAD ? NCT : 1>(gacc);
    double dsum[2][2] = {{0.0, 0.0}, {0.0, 0.0}};

#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const SweepSeg seg = grp.seg[sg];
        float c0 = 0.f, c1 = 0.f;
        if (GRAD) { c0 = (float)(a.gs[seg.fam * 2 + 0] * (double)a.it0); c1 = (float)(a.gs[seg.fam * 2 + 1] * (double)a.it1); }
        const int ntile = (seg.n + OT - 1) / OT;
        for (int jt = blockIdx.y; jt < ntile; jt += gridDim.y) {
            const int j0 = seg.row0 + jt * OT, j_end = seg.row0 + seg.n;
            f32x16 sacc[NJT];
            zero_acc<NJT>(sacc);
            for (int k0 = 0; k0 < a.Dp; k0 += SGA_KC) {
                __syncthreads();
                lds_load_rows<128, CT_THREADS>(own_s, a.Z, a.Dp, own0, own_end, k0, a.Dp, tid);
                lds_load_rows<OT, CT_THREADS>(oth_s, a.Z, a.Dp, j0, j_end, k0, a.Dp, tid);
                __syncthreads();
                mfma_chunk<NJT>(sacc, oth_s, own_s + (wave * 32 + (lane & 31)) * SGA_LDS_STRIDE, lane);
            }
            if (!GRAD) {
                float p0 = 0.f, p1 = 0.f;
                const bool iv = my_i < own_end;
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float okf = (iv && (j0 + t * 32 + mfma32_row(r, h) < j_end)) ? 1.f : 0.f;
                        p0 = fmaf(okf, fexp2(sacc[t][r] * a.k0), p0);
                        p1 = fmaf(okf, fexp2(sacc[t][r] * a.k1), p1);
                    }
                dsum[sg][0] += (double)p0;
                dsum[sg][1] += (double)p1;
            } else {
                // coefficient dL/d(dot) = sum_temp dL/ds * exp(dot/tau)/tau, in place (A-operand layout)
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sacc[t][r] = c0 * fexp2(sacc[t][r] * a.k0) + c1 * fexp2(sacc[t][r] * a.k1);
                __syncthreads();
                // stage the other rows' gradient columns [OT][GW] (zero beyond valid rows / Dp)
                for (int e = tid; e < OT * (GW / 4); e += CT_THREADS) {
                    const int r = e / (GW / 4), c = (e % (GW / 4)) * 4;
                    const int gr = j0 + r, gc = a.col0 + c;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (gr < j_end && gc < a.Dp) v = *reinterpret_cast<const f32x4*>(a.Z + (size_t)gr * a.Dp + gc);
                    *reinterpret_cast<f32x4*>(lds + r * GW + c) = v;
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float av = sacc[t][s];
                        const float* brow = lds + (t * 32 + mfma32_row(s, h)) * GW + (lane & 31);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            gacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[ct * 32], gacc[ct], 0, 0, 0);
                    }
            }
        }
    }
    if (!GRAD) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const double v = wave_sum_d(dsum[sg][tt]);
                if (lane == 0 && sg < grp.nseg && v != 0.0) atomicAdd(a.sums + 8 + my_slot() * 8 + grp.seg[sg].fam * 2 + tt, v);
            }
    } else {
        // gacc[ct][r] = dOwner[wave*32 + row(r,h)][col0 + ct*32 + (lane&31)]
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int d = a.col0 + ct * 32 + (lane & 31);
            if (d < a.Dp) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = own0 + wave * 32 + mfma32_row(r, h);
                    if (i < own_end) atomicAdd(a.dZ + (size_t)i * a.Dp + d, gacc[ct][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wide tables (Dp > 128: 1024-d modality tables, 300-/3072-d joint tables on the general path).  sweep_kernel<.,10,true> covers 320
// gradient columns per pass and recomputes the K = Dp similarity tile in every pass (Dp = 3072: 10 passes, 5.5x the necessary FLOPs;
// 293 of the 340 ms of a BASELINE configs[4]-shaped step).  For wide rows S is the expensive part, so the trade of the 100-d path
// is reversed: ONE anchor-owner sweep computes S and the coefficient c_ij = dL/dS_ij and writes it, transposed, to a stash
// Ct[g][j - n1][i - own0] (lane = anchor: 128-byte stores); both gradients are then plain GEMMs on the stash,
//   dZ[anchors] += Ct^T Z[negatives]   (gemm_tn)        dZ[negatives] += Ct Z[anchors]   (gemm_nn),
// so S is computed once instead of 2 x passes times.  The stash is bounded by the caller's workspace: anchor-row blocks.
// ------------------------------------------------------------------------------------------------
struct CoefArgs {
    const float* Z; int Dp; SweepGroup grp[2];
    float k0, k1, it0, it1;
    const double* gs;               // [8] dL/d(sums)
    float* stash[2];                // per anchor group: [J1 + J2][ld] (negative-major)
    int ld, n1;                     // stash row length (anchors in this block), first negative row of the packed table
};

template <int NJT>
__global__ __launch_bounds__(CT_THREADS) void sweep_coef_kernel(CoefArgs a) {
    constexpr int OT = NJT * 32;
    __shared__ __attribute__((aligned(16))) float own_s[128 * SGA_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float oth_s[OT * SGA_LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int g = ((int)blockIdx.x >= a.grp[1].blk0 && a.grp[1].nown > 0) ? 1 : 0;
    const SweepGroup& grp = a.grp[g];
    const int own0 = grp.own0 + ((int)blockIdx.x - grp.blk0) * 128;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + wave * 32 + (lane & 31);
    float* __restrict__ st = a.stash[g];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const SweepSeg seg = grp.seg[sg];
        const float c0 = (float)(a.gs[seg.fam * 2 + 0] * (double)a.it0), c1 = (float)(a.gs[seg.fam * 2 + 1] * (double)a.it1);
        const int ntile = (seg.n + OT - 1) / OT;
        for (int jt = blockIdx.y; jt < ntile; jt += gridDim.y) {
            const int j0 = seg.row0 + jt * OT, j_end = seg.row0 + seg.n;
            f32x16 sacc[NJT];
            zero_acc<NJT>(sacc);
            for (int k0 = 0; k0 < a.Dp; k0 += SGA_KC) {
                __syncthreads();
                lds_load_rows<128, CT_THREADS>(own_s, a.Z, a.Dp, own0, own_end, k0, a.Dp, tid);
                lds_load_rows<OT, CT_THREADS>(oth_s, a.Z, a.Dp, j0, j_end, k0, a.Dp, tid);
                __syncthreads();
                mfma_chunk<NJT>(sacc, oth_s, own_s + (wave * 32 + (lane & 31)) * SGA_LDS_STRIDE, lane);
            }
            if (my_i < own_end) {
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + t * 32 + mfma32_row(r, h);
                        if (j < j_end) st[(size_t)(j - a.n1) * a.ld + (my_i - grp.own0)] = c0 * fexp2(sacc[t][r] * a.k0) + c1 * fexp2(sacc[t][r] * a.k1);
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path of the sweeps for Dp <= 128 (emb_dim = 100 -> Dp = 104): the wave's 32 owner rows live in
// registers for the whole sweep (NQ float4 per lane = the MFMA B operand of every S tile), and each
// 128-row "other" tile is staged ONCE into LDS as full rows and serves both the S tiles (ds_read_b128
// along k) and the gradient GEMM (ds_read_b32 along the columns): one global->LDS pass and two barriers
// per tile instead of one per 32-wide K chunk and a second staging for the gradient.
// ------------------------------------------------------------------------------------------------
template <int NQ, bool GRAD>
__global__ __launch_bounds__(CT_THREADS) void sweep_fast_kernel(SweepArgs a) {
    constexpr int DP = NQ * 8;
    constexpr int STR = DP + 4;                         // (DP+4)/4 odd -> conflict-free ds_read_b128 over 16 rows
    constexpr int NJT = 4, OT = 128, NCT = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [OT][STR] + 32 floats of slack

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const SweepGroup& grp = a.grp[g];
    const int own0 = grp.own0 + ((int)blockIdx.x - grp.blk0) * 128;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + wave * 32 + (lane & 31);

    // owner rows -> registers (zero for rows past the group's end)
    f32x4 own[NQ];
    {
        const float* src = a.Z + (size_t)(my_i < own_end ? my_i : own0) * DP + 4 * h;
        const float msk = my_i < own_end ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + 8 * q);
            own[q] = v * msk;
        }
    }
    f32x16 gacc[GRAD ? NCT : 1];
    if (GRAD) zero_acc<GRAD ? NCT : 1>(gacc);
    double dsum[2][2] = {{0.0, 0.0}, {0.0, 0.0}};

#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const SweepSeg seg = grp.seg[sg];
        float c0 = 0.f, c1 = 0.f;
        if (GRAD) { c0 = (float)(a.gs[seg.fam * 2 + 0] * (double)a.it0); c1 = (float)(a.gs[seg.fam * 2 + 1] * (double)a.it1); }
        const int ntile = (seg.n + OT - 1) / OT;
        for (int jt = blockIdx.y; jt < ntile; jt += gridDim.y) {
            const int j0 = seg.row0 + jt * OT, j_end = seg.row0 + seg.n;
            __syncthreads();                              // previous tile fully consumed
            for (int e = tid; e < OT * (DP / 4); e += CT_THREADS) {
                const int r = e / (DP / 4), c = (e % (DP / 4)) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (j0 + r < j_end) v = *reinterpret_cast<const f32x4*>(a.Z + (size_t)(j0 + r) * DP + c);
                *reinterpret_cast<f32x4*>(lds + r * STR + c) = v;
            }
            __syncthreads();
            // ---- S tiles: lane = owner row, registers = other rows
            f32x16 sacc[NJT];
            zero_acc<NJT>(sacc);
            const float* ap = lds + (lane & 31) * STR + 4 * h;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int t = 0; t < NJT; ++t) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(ap + t * 32 * STR + 8 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], own[q][r], sacc[t], 0, 0, 0);
                }
            }
            if (!GRAD) {
                float p0 = 0.f, p1 = 0.f;
                const bool iv = my_i < own_end;
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float okf = (iv && (j0 + t * 32 + mfma32_row(r, h) < j_end)) ? 1.f : 0.f;
                        p0 = fmaf(okf, fexp2(sacc[t][r] * a.k0), p0);
                        p1 = fmaf(okf, fexp2(sacc[t][r] * a.k1), p1);
                    }
                dsum[sg][0] += (double)p0;
                dsum[sg][1] += (double)p1;
            } else {
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sacc[t][r] = c0 * fexp2(sacc[t][r] * a.k0) + c1 * fexp2(sacc[t][r] * a.k1);
                // ---- gradient GEMM straight from the accumulators (rows past j_end are zero in LDS)
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float av = sacc[t][s];
                        const float* brow = lds + (t * 32 + mfma32_row(s, h)) * STR + (lane & 31);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            gacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[ct * 32], gacc[ct], 0, 0, 0);
                    }
            }
        }
    }
    if (!GRAD) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const double v = wave_sum_d(dsum[sg][tt]);
                if (lane == 0 && sg < grp.nseg && v != 0.0) atomicAdd(a.sums + 8 + my_slot() * 8 + grp.seg[sg].fam * 2 + tt, v);
            }
    } else {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int d = ct * 32 + (lane & 31);
            if (d < DP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = own0 + wave * 32 + mfma32_row(r, h);
                    if (i < own_end) atomicAdd(a.dZ + (size_t)i * DP + d, gacc[ct][r]);
                }
            }
        }
    }
}

template <int NJT, int NCT, bool GRAD>
static void launch_sweep(const SweepArgs& a, int nblk, int gy, hipStream_t s) {
    const size_t sf = (size_t)(128 + NJT * 32) * SGA_LDS_STRIDE, gf = GRAD ? (size_t)NJT * 32 * NCT * 32 : 0;
    const size_t lds = (sf > gf ? sf : gf) * sizeof(float);
    auto k = sweep_kernel<NJT, NCT, GRAD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nblk, gy), dim3(CT_THREADS), lds, s, a);
}

template <int NQ, bool GRAD>
static void launch_sweep_fast(const SweepArgs& a, int nblk, int gy, hipStream_t s) {
    const size_t lds = (size_t)(128 * (NQ * 8 + 4) + 32) * sizeof(float);
    auto k = sweep_fast_kernel<NQ, GRAD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nblk, gy), dim3(CT_THREADS), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// Fused multi-table sweeps ("joint = fusion of these tables" case, the normal pipeline).
//
// The joint row is cat_m(w_m zhat_m)/||.|| (sg_aligner.py:32-34 followed by F.normalize in losses.py:44,
// 73), so with beta_m = w_m^2 / sum_k w_k^2 every joint similarity is  S_J = sum_m beta_m S_m : the 300-d
// table never has to be multiplied.  One launch computes the M modality S tiles of an
// (owner block, other tile) pair, derives S_J, and
//   SUM  mode: accumulates the 4x2 global sums of all M+1 tables;
//   GRAD mode: forms c_m = dL/dS_m + beta_m dL/dS_J, chains the M gradient GEMMs from the accumulators,
//              and accumulates Gamma_m = sum dL/dS_J * S_m  (= dL/dbeta_m through the negatives).
// vs. the per-table path this removes the joint table's S and gradient GEMMs (~half of all loss FLOPs).
// 32-row other tiles for all M tables are streamed into a double-buffered LDS ring by global_load_lds DMA (no VGPR
// round trip) one step ahead of the MFMAs.  Requires Dp == 104 (emb_dim 100) and 32 readable rows past the end of
// every Z buffer.  (A 32x32x2 form of this kernel -- 32 owner rows per wave, the whole 512-entry register file, one
// wave per SIMD -- ran the gradient sweep in 22.5 ms; sweep16_kernel below replaced it at 19.4 ms.)
// ------------------------------------------------------------------------------------------------
struct MultiArgs {
    int M; const float* Z[4]; int ngroups; SweepGroup grp[4];
    float k0, k1, it0, it1;
    const float* beta;              // [M]
    double* sums;                   // [(M+1)][8]            SUM out
    const double* gs;               // [(M+1)][8]            GRAD in (joint = row M)
    float* dZ[4];                   // GRAD out, atomic accumulate
    double* gamma;                  // [M]                   GRAD out
    int ktail;                      // K steps of 4 past k = 96 that hold data: ceil((D - 96) / 4), D = 100 -> 1 (columns 100..103 are zero padding)
    int swap_tail;                  // centred tables (sga_loss_centre_tables: columns 100 = b, 101 = 1): the OWNER reads columns 100 and 101 swapped, so that the K tail adds b_i + b_j
};

// ------------------------------------------------------------------------------------------------
// 16x16x4 form of the fused multi-table sweep (M = 2, 3; the path the benchmark runs).
//
// The 32x32 predecessor kept M*52 owner-operand registers and M*64 gradient accumulators per lane: the whole
// 512-entry file, one wave per SIMD, so nothing overlaps its exp2/coefficient VALU work with MFMA (measured 70 % of
// the fp32 MFMA peak).  Here a wave owns 16 rows instead of 32: M*26 operand + M*28 accumulator registers, 8 waves
// per workgroup = two per SIMD, and the second wave's MFMAs run under the first one's epilogue.  Same workgroup
// geometry otherwise (128 owner rows, 32-row other tiles of all M tables through the double-buffered DMA ring), so
// plan_multi's uniform work units are shared.  The gradient GEMM is 7 column tiles of 16 (112 >= 104) instead of
// 4 of 32 (128): 12 % less padded MFMA work.
//
// MFMA bookkeeping (v_mfma_f32_16x16x4_f32: A lane&15 = row, B lane&15 = column, lane>>4 = k; D[4*(lane>>4)+r][lane&15]):
//   S^T tile:  A = other rows from LDS, B = owner rows (registers)  ->  lane&15 = owner row, (lane>>4, r) = other row
//   which is exactly the A-operand layout of  dZ[own, :] += C[own, other] * Z[other, :]  with k = lane>>4.
// D row rho = 4*(lane>>4) + r holds other row pi(rho) = rho with bits 1 and 2 swapped: the gradient GEMM's B reads
// (ds_read_b32, lanes 0-31 = two k groups) then hit rows 2 apart = 16 banks apart instead of the same 16 banks.
// ------------------------------------------------------------------------------------------------
#ifndef SGA_S16_WAVES
#define SGA_S16_WAVES 4
#endif
constexpr int S16_WAVES = SGA_S16_WAVES;   // waves per workgroup; with 4, two workgroups share a CU (2 x 78 KiB of LDS)
// The owner rows (the S product's register operand, used for nothing else) carry the factor log2(e)/tau1, so the MFMA result is
// already the exp2 argument of the tau1 terms and the tau0 argument is one multiply away (k0/k1).  (Round 3; also tried there and
// dropped: gradient columns 96..99 on VALU -- a broadcast ds_read_b128 + 4 FMAs per step instead of the 7th, 3/4-padded MFMA
// tile: 24 MFMAs per tile less, same time; tools/experiments/r03_sweep16_vtail_prescale_nomask.diff.txt.)
constexpr int S16_THREADS = S16_WAVES * 64;
constexpr int S16_OWN = S16_WAVES * 16;     // owner rows per workgroup
// SGA_DBG_NOEXP / NOBAR / NODMA / NOS / NOG: timing-only ablation switches (wrong results) for tools/build_variant.sh;
// never defined in the product build (DESIGN.md 3b lists what they measured).
__device__ __forceinline__ int s16_pi(int rho) { return (rho & 9) | ((rho & 2) << 1) | ((rho & 4) >> 1); }
// A wave-uniform float that was produced by VALU arithmetic (so it sits in a VGPR) moved to an SGPR.
__device__ __forceinline__ float to_sgpr(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

// KT2: the table is wider than 100 columns (a second K step past k = 96 holds data).  emb_dim = 100 runs the KT2 = false build, whose
// register file has no room for the 2 M operands of a step it would never execute.
template <int M, bool GRAD, bool KT2 = false>
__global__ __launch_bounds__(S16_THREADS, (S16_WAVES == 4 && M <= 3) ? 2 : 1) void sweep16_kernel(MultiArgs a) {
    constexpr int DP = 104, OT = 32, NCT = 7, NTL = KT2 ? 2 : 1;
    constexpr int TILE_F = OT * DP, BUF_F = M * TILE_F;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][M][OT][DP] + slack for the 7th column tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const SweepGroup& grp = a.grp[g];
    // XCD-aware work order.  Workgroup b is dispatched to XCD b % 8 (its own 4 MiB L2); a group's work list, ordered (split major,
    // owner block minor), is cut into 8 contiguous chunks, one per XCD: the ~64 workgroups resident on an XCD at any time are
    // consecutive owner blocks of the SAME split, i.e. they stream the same "other" tiles in the same order and share them in that
    // XCD's L2 (owner-block-major order put every split on every XCD: 7.1 GB of L2 misses per launch at configs[1]).
    // plan_multi pads every group to a multiple of 8 workgroups (blk0 % 8 == 0); the padding workgroups exit here.
    const int wg_in_grp = (int)blockIdx.x - grp.blk0;
    const int nsplit = grp.nsplit, n_ob = (grp.nown + S16_OWN - 1) / S16_OWN, n_units = n_ob * nsplit;
    const int unit = (wg_in_grp & 7) * ((n_units + 7) >> 3) + (wg_in_grp >> 3);
    if ((wg_in_grp >> 3) >= ((n_units + 7) >> 3) || unit >= n_units) return;
    const int split = unit / n_ob;
    const int own0 = grp.own0 + (unit - split * n_ob) * S16_OWN;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + wave * 16 + l15;
    const bool iv = my_i < own_end;

    f32x4 own[M][6];
    float ownt[M][NTL], beta[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const float* src = a.Z[m] + (size_t)(iv ? my_i : own0) * DP;
        const float msk = iv ? a.k1 : 0.f;                      // pre-scaled: S arrives as the tau1 exp2 argument
#pragma unroll
        for (int q = 0; q < 6; ++q) own[m][q] = *reinterpret_cast<const f32x4*>(src + 16 * q + 4 * g4) * msk;   // k = 16q + 4g4 + r
#pragma unroll
        for (int t = 0; t < NTL; ++t) ownt[m][t] = src[96 + 4 * t + ((t == 1 && a.swap_tail) ? (g4 ^ 1) : g4)] * msk;      // k = 96 + 4t + g4
        beta[m] = a.beta[m];
    }
    f32x4 gacc[GRAD ? M : 1][NCT];
#pragma unroll
    for (int m = 0; m < (GRAD ? M : 1); ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) gacc[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gam[M];
#pragma unroll
    for (int m = 0; m < M; ++m) gam[m] = 0.f;
    // exp2 arguments from an S value `sv` as the MFMA delivers it (already times log2(e)/tau1):  tau1 term exp2(sv),  tau0 term exp2(sv * ka)
    const float ka = to_sgpr(a.k0 / a.k1);
    auto e0 = [&](float sv) { return fexp2(sv * ka); };
    auto e1 = [&](float sv) { return fexp2(sv); };

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // M0 (the DMA's LDS address) must be provably uniform
    // A tile is M x 13 chunks of 1 KiB (64 lanes x 16 B).  Wave w fetches chunks (w + m) % WAVES + WAVES * k of table m: the table index is
    // a compile-time constant of every DMA (its base pointer stays in two SGPRs) and the chunk offset is one uniform value plus a
    // literal.  (Chunk c = c0 + wave over the flattened M x 13 list made the table a run-time index: an s_load of a.Z[m] from the
    // kernel arguments + s_waitcnt before each of the 10 DMAs of every tile, and 80 loop-invariant SGPRs, half of them spilled.)
    auto issue = [&](int j0, float* buf) {
        int l4 = threadIdx.x;
        asm volatile("" : "+v"(l4));      // lane offset recomputed here (2 VALU): as a loop invariant it is folded into per-lane 64-bit bases that
        l4 = (l4 & 63) * 4;               // get spilled, and a scratch reload's vmcnt(0) in front of the DMAs serialises them
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int rot = (wave_u + m) & (S16_WAVES - 1);
            const float* src = a.Z[m] + ((size_t)j0 * DP + rot * 256) + l4;    // uniform base (SGPR pair) + lane offset
            float* dst = buf + m * TILE_F + rot * 256;
#pragma unroll
            for (int k = 0; k * S16_WAVES < 13; ++k) {
                if ((k + 1) * S16_WAVES > 13 && rot + k * S16_WAVES >= 13) break;      // uniform; only the last k can fall off the table
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * S16_WAVES * 256),
                                                 (__attribute__((address_space(3))) void*)(dst + k * S16_WAVES * 256), 16, 0, 0);
            }
        }
    };
    // this lane's other rows inside a 32-row tile: element (jh, r) <-> row jh*16 + pi(4*g4 + r)
    int jrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) jrow[r] = s16_pi(4 * g4 + r);
    const int arow = s16_pi(l15);                                  // the other row this lane feeds as MFMA A operand

#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const SweepSeg seg = grp.seg[sg];
        const int ntile = (seg.n + OT - 1) / OT;
        const int j_end = seg.row0 + seg.n;
        float c0[M + 1], c1[M + 1];
#pragma unroll
        for (int m = 0; m <= M; ++m) {                      // uniform, but fp64 arithmetic leaves them in VGPRs: 2 (M + 1) registers of a full file
            c0[m] = GRAD ? to_sgpr((float)(a.gs[m * 8 + seg.fam * 2 + 0] * (double)a.it0)) : 0.f;
            c1[m] = GRAD ? to_sgpr((float)(a.gs[m * 8 + seg.fam * 2 + 1] * (double)a.it1)) : 0.f;
        }
        double dsum[M + 1][2];
#pragma unroll
        for (int m = 0; m <= M; ++m) { dsum[m][0] = 0.0; dsum[m][1] = 0.0; }

        __syncthreads();
        if (split < ntile) issue(seg.row0 + split * OT, lds);
        int it = 0;
        for (int jt = split; jt < ntile; jt += nsplit, ++it) {
            float* buf = lds + (it & 1) * BUF_F;
            const int j0 = seg.row0 + jt * OT;
#ifndef SGA_DBG_NOBAR
            __syncthreads();                               // tile `it` landed / other buffer free
#endif
#ifndef SGA_DBG_NODMA
            if (jt + nsplit < ntile) issue(seg.row0 + (jt + nsplit) * OT, lds + ((it + 1) & 1) * BUF_F);
#endif
            if (GRAD && j0 + OT > j_end) {
                // Last, partial tile of a segment (uniform, once per segment): the rows past the segment's end hold the next segment's
                // data.  Zeroing them in LDS makes every one of their contributions vanish by itself -- S = 0, c * 0 added to the owner
                // gradient, 0 added to Gamma -- so the gradient epilogue needs no per-element validity mask at all.
                const int nval = j_end - j0;
                for (int x = tid; x < M * (OT - nval) * (DP / 4); x += S16_THREADS) {
                    const int m = x / ((OT - nval) * (DP / 4)), rem = x - m * ((OT - nval) * (DP / 4));
                    *reinterpret_cast<f32x4*>(buf + m * TILE_F + nval * DP + rem * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();
            }

            // ---- S^T tiles of the M tables: lane&15 = owner row, (g4, r) = other row.
            // Per 16-row half the M accumulation chains are interleaved (a dependent MFMA is M issues away) and the
            // A operands are read one K group ahead into the other of two register sets (assigning alternately, never
            // copying, keeps hipcc from folding the two sets back into one load->wait->use chain).
            f32x4 sacc[M][2];
#ifdef SGA_DBG_NOS
#pragma unroll
            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                for (int m = 0; m < M; ++m) { sacc[m][jh] = f32x4{0.1f, 0.2f, 0.3f, 0.4f} * buf[lane]; }
#else
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const float* ap = buf + (jh * 16 + arow) * DP + 4 * g4;
                const float* at = buf + (jh * 16 + arow) * DP + 96 + g4;
                f32x4 avA[M], avB[M];
                float tl[M][NTL];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    sacc[m][jh] = f32x4{0.f, 0.f, 0.f, 0.f};
                    avA[m] = *reinterpret_cast<const f32x4*>(ap + m * TILE_F);
                    avB[m] = avA[m];
                }
                __builtin_amdgcn_sched_group_barrier(0x100, M, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (q < 5) {
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            if (q & 1) avA[m] = *reinterpret_cast<const f32x4*>(ap + m * TILE_F + 16 * (q + 1));
                            else avB[m] = *reinterpret_cast<const f32x4*>(ap + m * TILE_F + 16 * (q + 1));
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < M; ++m)
#pragma unroll
                            for (int t = 0; t < NTL; ++t) tl[m][t] = at[m * TILE_F + 4 * t];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            sacc[m][jh] = __builtin_amdgcn_mfma_f32_16x16x4f32((q & 1) ? avB[m][r] : avA[m][r], own[m][q][r], sacc[m][jh], 0, 0, 0);
                    if (q < 5) __builtin_amdgcn_sched_group_barrier(0x100, M, 0);          // next group's reads first ...
                    else __builtin_amdgcn_sched_group_barrier(0x100, NTL * M, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * M, 0);                 // ... then this group's MFMAs
                }
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    if (t >= a.ktail) break;                       // uniform: an all-zero padding step is skipped (exact)
#pragma unroll
                    for (int m = 0; m < M; ++m)
                        sacc[m][jh] = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[m][t], ownt[m][t], sacc[m][jh], 0, 0, 0);
                }
            }
#endif

            if (!GRAD) {
                float p0[M + 1], p1[M + 1];
#pragma unroll
                for (int m = 0; m <= M; ++m) { p0[m] = 0.f; p1[m] = 0.f; }
                // interior tiles (every owner row of the block and every other row of the tile valid) add unmasked
                auto sums_tile = [&](auto masked_c) {
                    constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float okf = (!MASKED || (iv && (j0 + jh * 16 + jrow[r] < j_end))) ? 1.f : 0.f;
                            float sj = 0.f;
#pragma unroll
                            for (int m = 0; m < M; ++m) {
                                const float sv = sacc[m][jh][r];
                                sj = fmaf(beta[m], sv, sj);
                                p0[m] = MASKED ? fmaf(okf, e0(sv), p0[m]) : p0[m] + e0(sv);
                                p1[m] = MASKED ? fmaf(okf, e1(sv), p1[m]) : p1[m] + e1(sv);
                            }
                            p0[M] = MASKED ? fmaf(okf, e0(sj), p0[M]) : p0[M] + e0(sj);
                            p1[M] = MASKED ? fmaf(okf, e1(sj), p1[M]) : p1[M] + e1(sj);
                        }
                };
                if (j0 + OT <= j_end && own0 + S16_OWN <= own_end) sums_tile(std::false_type{}); else sums_tile(std::true_type{});   // uniform
#pragma unroll
                for (int m = 0; m <= M; ++m) { dsum[m][0] += (double)p0[m]; dsum[m][1] += (double)p1[m]; }
            } else {
                // No validity masks here: rows past a segment's end were zeroed in LDS above, and owner rows past the group's end carry
                // zero operands and are never written back.
                float cj[2][4];
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sj = 0.f;
#pragma unroll
                        for (int m = 0; m < M; ++m) sj = fmaf(beta[m], sacc[m][jh][r], sj);
                        cj[jh][r] = c0[M] * e0(sj) + c1[M] * e1(sj);
                    }
                if (g < 2) {                                    // Gamma_m = sum dL/dS_J * S_m, each pair once (anchor-owner sweep)
#pragma unroll
                    for (int m = 0; m < M; ++m)
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) gam[m] = fmaf(cj[jh][r], sacc[m][jh][r], gam[m]);
                }
                // c_m = dL/dS_m + beta_m dL/dS_J, then dZ_m[own, :] += c_m * Z_m[other, :].  The M*8 (table, other row)
                // steps are software pipelined: while step e's 7 MFMAs issue, step e+1's coefficient (2 exp2 + ~8 VALU)
                // and its 7 B operands (ds_read_b32) are produced into the other register set.
                float cmv[2], bv[2][NCT];
                auto coef = [&](int e) {
                    const int m = e >> 3, jh = (e >> 2) & 1, r = e & 3;
                    const float sv = sacc[m][jh][r];
                    return fmaf(beta[m], cj[jh][r], c0[m] * e0(sv) + c1[m] * e1(sv));
                };
                auto bload = [&](int e, float* dst) {
                    const int m = e >> 3, jh = (e >> 2) & 1, r = e & 3;
                    const float* bb = buf + m * TILE_F + (jh * 16 + jrow[r]) * DP + l15;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) dst[ct] = bb[ct * 16];
                };
                cmv[0] = coef(0);
                bload(0, bv[0]);
#pragma unroll
                for (int e = 0; e < M * 8; ++e) {
                    if (e + 1 < M * 8) { cmv[(e + 1) & 1] = coef(e + 1); bload(e + 1, bv[(e + 1) & 1]); }
#ifdef SGA_DBG_NOG
                    gacc[GRAD ? (e >> 3) : 0][0][0] += cmv[e & 1] * bv[e & 1][0];
#else
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        gacc[GRAD ? (e >> 3) : 0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cmv[e & 1], bv[e & 1][ct], gacc[GRAD ? (e >> 3) : 0][ct], 0, 0, 0);
#endif
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < M; ++m) asm volatile("" : "+v"(gam[m]));     // keep the updates out of the loop latch
            }
        }
        if (!GRAD) {
#pragma unroll
            for (int m = 0; m <= M; ++m)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const double v = wave_sum_d(dsum[m][tt]);
                    if (lane == 0 && v != 0.0) atomicAdd(a.sums + (M + 1) * 8 * (1 + my_slot()) + m * 8 + seg.fam * 2 + tt, v);
                }
        }
    }
    if (GRAD) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float* dz = a.dZ[m];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int d = ct * 16 + l15;
                if (d < DP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = own0 + wave * 16 + 4 * g4 + r;
                        if (i < own_end) atomicAdd(dz + (size_t)i * DP + d, gacc[GRAD ? m : 0][ct][r]);
                    }
                }
            }
        }
        if (g < 2) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(gam[m]) / a.k1;       // Gamma was accumulated on pre-scaled S values
                if (lane == 0 && v != 0.f) atomicAdd(a.gamma + M * (1 + my_slot()) + m, (double)v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// M = 4 (point + gat + rel + attr: the module list of every config the reference ships) on PAIRED waves.
//
// sweep16_kernel<4> needs 104 owner-operand + 112 gradient-accumulator + 32 S-tile registers per lane: one wave per SIMD, so
// nothing runs under a wave's epilogue or barrier wait (0.61 of the MFMA peak against 0.72 for M = 3 with two waves per SIMD).
// Here a workgroup still owns 64 rows but runs 8 waves: wave (rg, th) owns row group rg (16 rows) and TABLES {2 th, 2 th + 1} only --
// half of the operands, S tiles and accumulators (~200 registers: two waves per SIMD again).  The joint similarity needs all four
// S tiles of an element, so the two waves of a row group swap their halves through LDS once per tile (16 values per lane each way,
// one extra workgroup barrier -- a pairwise LDS-flag hand-over instead measured 2 % slower: the spinning wave takes issue slots);
// the joint coefficient is then computed by both (the only duplicated work, ~100 VALU per tile) and
// each wave forms the coefficients and gradient GEMMs of its own two tables.  Same work units, XCD order and DMA ring as sweep16_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int S4_THREADS = 512;
template <bool GRAD>
__global__ __launch_bounds__(S4_THREADS, 1) void sweep16x2_kernel(MultiArgs a) {
    constexpr int M = 4, MT = 2, DP = 104, OT = 32, NCT = 7;
    constexpr int TILE_F = OT * DP, BUF_F = M * TILE_F;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][M][OT][DP] + 32 slack + exchange [8 waves][16][64]
    float* xbuf = lds + 2 * BUF_F + 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int rg = wave & 3, th = wave >> 2;                        // row group, table half
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const SweepGroup& grp = a.grp[g];
    const int wg_in_grp = (int)blockIdx.x - grp.blk0;
    const int nsplit = grp.nsplit, n_ob = (grp.nown + S16_OWN - 1) / S16_OWN, n_units = n_ob * nsplit;
    const int unit = (wg_in_grp & 7) * ((n_units + 7) >> 3) + (wg_in_grp >> 3);     // XCD-aware order, see sweep16_kernel
    if ((wg_in_grp >> 3) >= ((n_units + 7) >> 3) || unit >= n_units) return;
    const int split = unit / n_ob;
    const int own0 = grp.own0 + (unit - split * n_ob) * S16_OWN;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + rg * 16 + l15;
    const bool iv = my_i < own_end;

    f32x4 own[MT][6];
    float ownt[MT][2], bm[MT], bp[MT];                             // beta of this wave's tables / of its partner's
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float* src = a.Z[2 * th + m] + (size_t)(iv ? my_i : own0) * DP;
        const float msk = iv ? a.k1 : 0.f;                      // pre-scaled owner rows, as in sweep16_kernel
#pragma unroll
        for (int q = 0; q < 6; ++q) own[m][q] = *reinterpret_cast<const f32x4*>(src + 16 * q + 4 * g4) * msk;
#pragma unroll
        for (int t = 0; t < 2; ++t) ownt[m][t] = src[96 + 4 * t + ((t == 1 && a.swap_tail) ? (g4 ^ 1) : g4)] * msk;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) { bm[m] = a.beta[2 * th + m]; bp[m] = a.beta[2 * (1 - th) + m]; }
    f32x4 gacc[GRAD ? MT : 1][NCT];
#pragma unroll
    for (int m = 0; m < (GRAD ? MT : 1); ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) gacc[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gam[MT] = {0.f, 0.f};
    const float ka = a.k0 / a.k1;                                  // S arrives times log2(e)/tau1: tau1 term exp2(sv), tau0 term exp2(sv * ka)
    auto e0 = [&](float sv) { return fexp2(sv * ka); };
    auto e1 = [&](float sv) { return fexp2(sv); };

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int j0, float* buf) {      // chunk (wave + 2m) % 8 (+ 8) of table m: compile-time table index, see sweep16_kernel
        int l4 = threadIdx.x;
        asm volatile("" : "+v"(l4));
        l4 = (l4 & 63) * 4;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int rot = (wave_u + 2 * m) & 7;
            const float* src = a.Z[m] + ((size_t)j0 * DP + rot * 256) + l4;
            float* dst = buf + m * TILE_F + rot * 256;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && rot + 8 >= 13) break;            // uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 8 * 256),
                                                 (__attribute__((address_space(3))) void*)(dst + k * 8 * 256), 16, 0, 0);
            }
        }
    };
    int jrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) jrow[r] = s16_pi(4 * g4 + r);
    const int arow = s16_pi(l15);
    float* xmine = xbuf + ((rg * 2 + th) * 16) * 64 + lane;
    const float* xpart = xbuf + ((rg * 2 + (1 - th)) * 16) * 64 + lane;

#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const SweepSeg seg = grp.seg[sg];
        const int ntile = (seg.n + OT - 1) / OT;
        const int j_end = seg.row0 + seg.n;
        float c0[MT + 1], c1[MT + 1];                              // this wave's two tables, then the joint table
#pragma unroll
        for (int m = 0; m <= MT; ++m) {
            const int tab = m == MT ? M : 2 * th + m;
            c0[m] = GRAD ? (float)(a.gs[tab * 8 + seg.fam * 2 + 0] * (double)a.it0) : 0.f;
            c1[m] = GRAD ? (float)(a.gs[tab * 8 + seg.fam * 2 + 1] * (double)a.it1) : 0.f;
        }
        double dsum[MT + 1][2];                                    // own two tables + this wave's half of the joint table
#pragma unroll
        for (int m = 0; m <= MT; ++m) { dsum[m][0] = 0.0; dsum[m][1] = 0.0; }

        __syncthreads();
        if (split < ntile) issue(seg.row0 + split * OT, lds);
        int it = 0;
        for (int jt = split; jt < ntile; jt += nsplit, ++it) {
            float* buf = lds + (it & 1) * BUF_F;
            const int j0 = seg.row0 + jt * OT;
            __syncthreads();                               // tile `it` landed / other buffer free / exchange buffer free
            if (jt + nsplit < ntile) issue(seg.row0 + (jt + nsplit) * OT, lds + ((it + 1) & 1) * BUF_F);
            if (GRAD && j0 + OT > j_end) {                 // partial last tile of a segment: zero the rows past its end (see sweep16_kernel)
                const int nval = j_end - j0;
                for (int x = tid; x < M * (OT - nval) * (DP / 4); x += S4_THREADS) {
                    const int m = x / ((OT - nval) * (DP / 4)), rem = x - m * ((OT - nval) * (DP / 4));
                    *reinterpret_cast<f32x4*>(buf + m * TILE_F + nval * DP + rem * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();
            }

            // ---- S^T tiles of this wave's two tables
            f32x4 mine[MT][2], part[MT][2];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const float* ap = buf + (2 * th) * TILE_F + (jh * 16 + arow) * DP + 4 * g4;
                const float* at = buf + (2 * th) * TILE_F + (jh * 16 + arow) * DP + 96 + g4;
                f32x4 sacc[MT], avA[MT], avB[MT];
                float tl[MT][2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    sacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                    avA[m] = *reinterpret_cast<const f32x4*>(ap + m * TILE_F);
                    avB[m] = avA[m];
                }
                __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (q < 5) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            if (q & 1) avA[m] = *reinterpret_cast<const f32x4*>(ap + m * TILE_F + 16 * (q + 1));
                            else avB[m] = *reinterpret_cast<const f32x4*>(ap + m * TILE_F + 16 * (q + 1));
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < MT; ++m) { tl[m][0] = at[m * TILE_F]; tl[m][1] = at[m * TILE_F + 4]; }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            sacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32((q & 1) ? avB[m][r] : avA[m][r], own[m][q][r], sacc[m], 0, 0, 0);
                    if (q < 5) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT, 0);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (t >= a.ktail) break;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        sacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[m][t], ownt[m][t], sacc[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xmine[((m * 2 + jh) * 4 + r) * 64] = sacc[m][r];
                    mine[m][jh] = sacc[m];
                }
            }
            __syncthreads();                               // both halves of every row group are in the exchange buffer
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = xpart[((m * 2 + jh) * 4 + r) * 64];
                    part[m][jh] = v;
                }

            if (!GRAD) {
                float p0[MT + 1], p1[MT + 1];
#pragma unroll
                for (int m = 0; m <= MT; ++m) { p0[m] = 0.f; p1[m] = 0.f; }
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float okf = (iv && (j0 + jh * 16 + jrow[r] < j_end)) ? 1.f : 0.f;
                        float sj = 0.f;
#pragma unroll
                        for (int m = 0; m < MT; ++m) sj = fmaf(bm[m], mine[m][jh][r], fmaf(bp[m], part[m][jh][r], sj));
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const float sv = mine[m][jh][r];
                            p0[m] = fmaf(okf, e0(sv), p0[m]);
                            p1[m] = fmaf(okf, e1(sv), p1[m]);
                        }
                        if (jh == th) {                             // wave-uniform: the joint table's sums, one 16-row half per partner
                            p0[MT] = fmaf(okf, e0(sj), p0[MT]);
                            p1[MT] = fmaf(okf, e1(sj), p1[MT]);
                        }
                    }
#pragma unroll
                for (int m = 0; m <= MT; ++m) { dsum[m][0] += (double)p0[m]; dsum[m][1] += (double)p1[m]; }
            } else {
                float cj[2][4];                             // no validity masks: see sweep16_kernel
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sj = 0.f;
#pragma unroll
                        for (int m = 0; m < MT; ++m) sj = fmaf(bm[m], mine[m][jh][r], fmaf(bp[m], part[m][jh][r], sj));
                        cj[jh][r] = c0[MT] * e0(sj) + c1[MT] * e1(sj);
                    }
                if (g < 2) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) gam[m] = fmaf(cj[jh][r], mine[m][jh][r], gam[m]);
                }
                float cmv[2], bv[2][NCT];
                auto coef = [&](int e) {
                    const int m = e >> 3, jh = (e >> 2) & 1, r = e & 3;
                    const float sv = mine[m][jh][r];
                    return fmaf(bm[m], cj[jh][r], c0[m] * e0(sv) + c1[m] * e1(sv));
                };
                auto bload = [&](int e, float* dst) {
                    const int m = e >> 3, jh = (e >> 2) & 1, r = e & 3;
                    const float* bb = buf + (2 * th + m) * TILE_F + (jh * 16 + jrow[r]) * DP + l15;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) dst[ct] = bb[ct * 16];
                };
                cmv[0] = coef(0);
                bload(0, bv[0]);
#pragma unroll
                for (int e = 0; e < MT * 8; ++e) {
                    if (e + 1 < MT * 8) { cmv[(e + 1) & 1] = coef(e + 1); bload(e + 1, bv[(e + 1) & 1]); }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        gacc[GRAD ? (e >> 3) : 0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cmv[e & 1], bv[e & 1][ct], gacc[GRAD ? (e >> 3) : 0][ct], 0, 0, 0);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(gam[m]));
            }
        }
        if (!GRAD) {
#pragma unroll
            for (int m = 0; m <= MT; ++m) {
                const int tab = m == MT ? M : 2 * th + m;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const double v = wave_sum_d(dsum[m][tt]);
                    if (lane == 0 && v != 0.0) atomicAdd(a.sums + (M + 1) * 8 * (1 + my_slot()) + tab * 8 + seg.fam * 2 + tt, v);
                }
            }
        }
    }
    if (GRAD) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float* dz = a.dZ[2 * th + m];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int d = ct * 16 + l15;
                if (d < DP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = own0 + rg * 16 + 4 * g4 + r;
                        if (i < own_end) atomicAdd(dz + (size_t)i * DP + d, gacc[GRAD ? m : 0][ct][r]);
                    }
                }
            }
        }
        if (g < 2) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float v = wave_sum(gam[m]) / a.k1;       // accumulated on pre-scaled S values
                if (lane == 0 && v != 0.f) atomicAdd(a.gamma + M * (1 + my_slot()) + 2 * th + m, (double)v);
            }
        }
    }
}

template <bool GRAD>
static void launch_sweep16x2(const MultiArgs& a, int nwg, hipStream_t s) {
    const size_t lds = ((size_t)2 * 4 * 32 * 104 + 32 + 8 * 16 * 64) * sizeof(float);
    auto k = sweep16x2_kernel<GRAD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(S4_THREADS), lds, s, a);
}

template <int M, bool GRAD>
static void launch_sweep16(const MultiArgs& a, int nwg, hipStream_t s) {
    const size_t lds = ((size_t)2 * M * 32 * 104 + 32) * sizeof(float);
    auto k = a.ktail > 1 ? sweep16_kernel<M, GRAD, true> : sweep16_kernel<M, GRAD, false>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(S16_THREADS), lds, s, a);
}

// joint operand rows for the anchors x anchors kernels: ZJ[r, m*104 + d] = sqrt(beta_m) Z_m[r, d]
__global__ void build_joint_kernel(MultiArgs a, float* __restrict__ ZJ, int rows) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += gridDim.x * wpb)
        for (int m = 0; m < a.M; ++m) {
            const float sb = sqrtf(a.beta[m]);
            for (int d = lane; d < 104; d += 64) ZJ[(size_t)r * (a.M * 104) + m * 104 + d] = sb * a.Z[m][(size_t)r * 104 + d];
        }
}

// fold dL/dZJ back: dZ_m[r,:] += sqrt(beta_m) dZJ[r, block m];  gamma2[m] += <dZJ[r, block m], Z_m[r,:]>
__global__ void fold_joint_kernel(MultiArgs a, const float* __restrict__ dZJ, int rows, double* __restrict__ gamma2) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += gridDim.x * wpb)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m >= a.M) break;
            const float sb = sqrtf(a.beta[m]);
            for (int d = lane; d < 104; d += 64) {
                const float gj = dZJ[(size_t)r * (a.M * 104) + m * 104 + d];
                acc[m] = fmaf(gj, a.Z[m][(size_t)r * 104 + d], acc[m]);
                a.dZ[m][(size_t)r * 104 + d] += sb * gj;
            }
        }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float v = wave_sum(acc[m]);
        if (lane == 0 && m < a.M && v != 0.f) atomicAdd(gamma2 + m, (double)v);
    }
}

// poison = NaN if any row of any table took F.normalize's eps branch (then S_J != sum beta_m S_m)
__global__ void check_norms_kernel(const float* __restrict__ nrm, int n, float* __restrict__ poison) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (!(nrm[i] >= 1e-12f)) *poison = __builtin_nanf("");
}

// fp16 staging of the general-width anchors x anchors kernel: the same LDS tiles hold [rows][64 halfs + 8 pad] (144 B = the fp32 tiles' 36
// floats per row: conflict-free ds_read_b128), a K chunk is 64 columns = 4 steps of v_mfma_f32_32x32x16_f16 (lane: row lane & 31, k slots
// 8 (lane >> 5) .. + 7 of each step)
typedef _Float16 ak_f16x8 __attribute__((ext_vector_type(8)));
template <int NROWS, int NTHREADS>
__device__ __forceinline__ void lds_load_rows_h(float* __restrict__ tile, const _Float16* __restrict__ g, int ld, int row0, int nrows, int k0,
                                                int ncols, int tid) {
    unsigned char* t8 = reinterpret_cast<unsigned char*>(tile);
#pragma unroll
    for (int e = tid; e < NROWS * 8; e += NTHREADS) {
        const int r = e >> 3, c = (e & 7) * 8;
        ak_f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int gr = row0 + r, gc = k0 + c;
        if (gr < nrows && gc < ncols) v = *reinterpret_cast<const ak_f16x8*>(g + (size_t)gr * ld + gc);     // ncols % 8 == 0
        *reinterpret_cast<ak_f16x8*>(t8 + r * 144 + c * 2) = v;
    }
}
template <int NT>
__device__ __forceinline__ void mfma_chunk_h(f32x16 (&acc)[NT], const float* __restrict__ a_tile, const float* __restrict__ b_row, int lane) {
    const unsigned char* ap = reinterpret_cast<const unsigned char*>(a_tile) + (lane & 31) * 144 + (lane >> 5) * 16;
    const unsigned char* bp = reinterpret_cast<const unsigned char*>(b_row) + (lane >> 5) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const ak_f16x8 b = *reinterpret_cast<const ak_f16x8*>(bp + 32 * q);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const ak_f16x8 av = *reinterpret_cast<const ak_f16x8*>(ap + t * 32 * 144 + 32 * q);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b, acc[t], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// anchors x anchors: loss terms (fwd) and dL/dS + dL/d(sums) (bwd), all tables in one pass
// ------------------------------------------------------------------------------------------------
struct AnchorArgs {
    int NT, A, i_lo, i_hi;           // anchor rows [i_lo, i_hi) are this process's shard of block I
    const float* Z[CT_MAXT]; int Dp[CT_MAXT];
    const _Float16* Zh[CT_MAXT];   // optional (MFMA mode 'f16', tables wider than 128 columns): fp16 copy of table k's rows -- its similarities then run
                                   // on v_mfma_f32_32x32x16_f16 (fp16 inputs, fp32 accumulate: the arithmetic of wide16.hip's sweeps)
    const double* sums;            // [NT][8]
    float alpha, kc, ki, itc, iti; // ICL alpha; log2e/tau and 1/tau for ICL (c) and IAL (i)
    double* out;                   // fwd: [NT] icl sums, [M] iala, [M] ialb
    const float* coef;             // bwd: upstream dL/d(out) in the same order
    float* M1[CT_MAXT];            // bwd: stash, M1[k][j*A + i] = dL/dS_k[i,j]
    double* gs;                    // bwd: [NT][8] dL/d(sums)
    const float* SP[CT_MAXT];      // PRE: the similarity blocks formed beforehand (wide16.hip's tile core), SP[k][j * ldp + (i - i_lo)] = X1[i] . X2[j]
    const float* SQ[CT_MAXT];      //      SQ[k][j * ldp + (i - i_lo)] = X2[i] . X1[j]
    long ldp;
};

// PRE: epilogue only -- every table's two similarity blocks are read from memory in the accumulator layout (lanes along i: coalesced), no
// K loop, no LDS tiles (mode 'f16' with all tables wide: the products run on the fp16 tile core at ~0.4 of the fp16 MFMA peak instead of
// this kernel's single-buffered 128 x 64 staging).
template <bool BWD, bool PRE = false>
__global__ __launch_bounds__(CT_THREADS) void anchor_kernel(AnchorArgs a) {
    constexpr int NJT = PRE ? 1 : 2, OT = 32 * NJT;         // (PRE: one 32-column tile per workgroup -- half the live registers of the epilogue)
    constexpr int LR = PRE ? 1 : 128, LO = PRE ? 1 : OT;
    __shared__ __attribute__((aligned(16))) float own1[LR * SGA_LDS_STRIDE];    // X1 rows of block I
    __shared__ __attribute__((aligned(16))) float own2[LR * SGA_LDS_STRIDE];    // X2 rows of block I
    __shared__ __attribute__((aligned(16))) float oth1[LO * SGA_LDS_STRIDE];    // X2 rows of block J  (for P)
    __shared__ __attribute__((aligned(16))) float oth2[LO * SGA_LDS_STRIDE];    // X1 rows of block J  (for Q)
    __shared__ float inv_s[CT_MAXT * 8];                                        // 1/(sum + 1e-9)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int A = a.A, NT = a.NT, M = NT > 1 ? NT - 1 : 0;
    double* const out_s = BWD ? nullptr : a.out + (NT + 2 * M) * (1 + my_slot());
    double* const gs_s = BWD ? a.gs + NT * 8 * (1 + my_slot()) : nullptr;
    for (int e = tid; e < NT * 8; e += CT_THREADS) inv_s[e] = (float)(1.0 / (a.sums[e] + 1e-9));
    const int i0 = a.i_lo + blockIdx.x * 128, j0 = blockIdx.y * OT;
    const int my_i = i0 + wave * 32 + (lane & 31);
    const bool iv = my_i < a.i_hi;
    const int ns = a.i_hi - a.i_lo;

    f32x16 xJ[NJT], gJ[NJT];
    zero_acc<NJT>(xJ);
    zero_acc<NJT>(gJ);

    for (int it = 0; it < NT; ++it) {
        const int k = (NT > 1) ? (it == 0 ? NT - 1 : it - 1) : 0;       // joint first, then the modalities
        const bool is_joint = NT > 1 && it == 0;
        const float* Z = a.Z[k];
        const int Dp = a.Dp[k];
        f32x16 P[NJT], Q[NJT];
        zero_acc<NJT>(P);
        zero_acc<NJT>(Q);
        const _Float16* Zh = a.Zh[k];
        if constexpr (PRE) {
            __syncthreads();                                    // (inv_s)
            const float* sp = a.SP[k] + (iv ? my_i - a.i_lo : 0);
            const float* sq = a.SQ[k] + (iv ? my_i - a.i_lo : 0);
#pragma unroll
            for (int t = 0; t < NJT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = min(j0 + t * 32 + mfma32_row(r, h), A - 1);
                    P[t][r] = sp[(size_t)j * a.ldp];
                    Q[t][r] = sq[(size_t)j * a.ldp];
                }
        } else
        if (Zh) {                                               // uniform: fp16 inputs, 64 columns per chunk
            for (int k0 = 0; k0 < Dp; k0 += 64) {
                __syncthreads();
                lds_load_rows_h<128, CT_THREADS>(own1, Zh, Dp, i0, A, k0, Dp, tid);
                lds_load_rows_h<128, CT_THREADS>(own2, Zh, Dp, A + i0, 2 * A, k0, Dp, tid);
                lds_load_rows_h<OT, CT_THREADS>(oth1, Zh, Dp, A + j0, 2 * A, k0, Dp, tid);
                lds_load_rows_h<OT, CT_THREADS>(oth2, Zh, Dp, j0, A, k0, Dp, tid);
                __syncthreads();
                const int ro = (wave * 32 + (lane & 31)) * SGA_LDS_STRIDE;
                mfma_chunk_h<NJT>(P, oth1, own1 + ro, lane);
                mfma_chunk_h<NJT>(Q, oth2, own2 + ro, lane);
            }
        } else
        for (int k0 = 0; k0 < Dp; k0 += SGA_KC) {
            __syncthreads();
            lds_load_rows<128, CT_THREADS>(own1, Z, Dp, i0, A, k0, Dp, tid);
            lds_load_rows<128, CT_THREADS>(own2, Z, Dp, A + i0, 2 * A, k0, Dp, tid);
            lds_load_rows<OT, CT_THREADS>(oth1, Z, Dp, A + j0, 2 * A, k0, Dp, tid);
            lds_load_rows<OT, CT_THREADS>(oth2, Z, Dp, j0, A, k0, Dp, tid);
            __syncthreads();
            const int ro = (wave * 32 + (lane & 31)) * SGA_LDS_STRIDE;
            mfma_chunk<NJT>(P, oth1, own1 + ro, lane);      // P[i,j] = X1[i].X2[j] = S[i,j]
            mfma_chunk<NJT>(Q, oth2, own2 + ro, lane);      // Q[i,j] = X2[i].X1[j] = S[j,i]
        }
        if (is_joint) {
#pragma unroll
            for (int t = 0; t < NJT; ++t) xJ[t] = P[t];
        }
        const float* is = inv_s + k * 8;                    // [fam*2 + temp]
        const float a11c = is[0], a12c = is[2], a22c = is[4], a21c = is[6];
        const float a11i = is[1], a12i = is[3], a22i = is[5], a21i = is[7];
        const float* js = inv_s + (NT - 1) * 8;
        const float j11 = js[1], j12 = js[3], j22 = js[5], j21 = js[7];

        if (!BWD) {
            float icl = 0.f, la = 0.f, lb = 0.f;
#pragma unroll
            for (int t = 0; t < NJT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // (selects, not mask multiplies, here: measured 2 ms faster per launch in this VALU-heavy epilogue)
                    const bool ok = iv && (j0 + t * 32 + mfma32_row(r, h) < A);
                    const float x = P[t][r], y = Q[t][r];
                    const float qa = g_val(fexp2(x * a.kc), a11c, a12c);
                    const float qb = g_val(fexp2(y * a.kc), a22c, a21c);
                    const float term = -flog(a.alpha * qa + (1.f - a.alpha) * qb);
                    icl += ok ? term : 0.f;
                    if (M > 0 && !is_joint) {
                        const float dm = fexp2(x * a.ki), dj = fexp2(xJ[t][r] * a.ki);
                        const float qoa = g_val(dm, a11i, a12i), qma = g_val(dj, j11, j12);
                        const float qob = g_val(dm, a22i, a21i), qmb = g_val(dj, j22, j21);
                        const float ta = __expf(qoa) * (qoa - flog(qma));
                        const float tb = __expf(qob) * (qob - flog(qmb));
                        la += ok ? ta : 0.f;
                        lb += ok ? tb : 0.f;
                    }
                    // one element at a time: fully interleaved, the 32 unrolled elements need >512 registers (spills, 1 wave/SIMD)
                    __builtin_amdgcn_sched_barrier(0);
                }
            icl = wave_sum(icl);
            if (lane == 0) atomicAdd(out_s + k, (double)icl);
            if (M > 0 && !is_joint) {
                la = wave_sum(la);
                lb = wave_sum(lb);
                if (lane == 0) { atomicAdd(out_s + NT + k, (double)la); atomicAdd(out_s + NT + M + k, (double)lb); }
            }
        } else {
            const float c = a.coef[k];
            const float ca = (M > 0 && !is_joint) ? a.coef[NT + k] : 0.f;
            const float cb = (M > 0 && !is_joint) ? a.coef[NT + M + k] : 0.f;
            float gs_c[4] = {0.f, 0.f, 0.f, 0.f};          // this table, ICL temperature
            float gs_i[4] = {0.f, 0.f, 0.f, 0.f};          // this table, IAL temperature
            float gs_j[4] = {0.f, 0.f, 0.f, 0.f};          // joint table, IAL temperature
#pragma unroll
            for (int t = 0; t < NJT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = iv && (j0 + t * 32 + mfma32_row(r, h) < A);
                    const float x = P[t][r], y = Q[t][r];
                    const float dx = fexp2(x * a.kc), dy = fexp2(y * a.kc);
                    const GV Ax = g_full(dx, a11c, a12c), Bx = g_full(dx, a22c, a21c);
                    const float qAy = g_val(dy, a11c, a12c), qBy = g_val(dy, a22c, a21c);
                    const float z_ij = a.alpha * Ax.q + (1.f - a.alpha) * qBy;
                    const float z_ji = a.alpha * qAy + (1.f - a.alpha) * Bx.q;
                    const float wA = ok ? -c * a.alpha * frcp(z_ij) : 0.f;
                    const float wB = ok ? -c * (1.f - a.alpha) * frcp(z_ji) : 0.f;
                    float gx = (wA * Ax.dd + wB * Bx.dd) * dx * a.itc;
                    gs_c[0] += wA * Ax.dsa; gs_c[1] += wA * Ax.dsb; gs_c[2] += wB * Bx.dsa; gs_c[3] += wB * Bx.dsb;
                    if (M > 0 && !is_joint) {
                        const float dm = fexp2(x * a.ki), dj = fexp2(xJ[t][r] * a.ki);
                        const GV OA = g_full(dm, a11i, a12i), OB = g_full(dm, a22i, a21i);
                        const GV MA = g_full(dj, j11, j12), MB = g_full(dj, j22, j21);
                        const float eA = ok ? __expf(OA.q) : 0.f, eB = ok ? __expf(OB.q) : 0.f;
                        const float tA = ca * eA * (OA.q - flog(MA.q) + 1.f), uA = -ca * eA * frcp(MA.q);
                        const float tB = cb * eB * (OB.q - flog(MB.q) + 1.f), uB = -cb * eB * frcp(MB.q);
                        gx += (tA * OA.dd + tB * OB.dd) * dm * a.iti;
                        gJ[t][r] += (uA * MA.dd + uB * MB.dd) * dj * a.iti;
                        gs_i[0] += tA * OA.dsa; gs_i[1] += tA * OA.dsb; gs_i[2] += tB * OB.dsa; gs_i[3] += tB * OB.dsb;
                        gs_j[0] += uA * MA.dsa; gs_j[1] += uA * MA.dsb; gs_j[2] += uB * MB.dsa; gs_j[3] += uB * MB.dsb;
                    }
                    P[t][r] = gx;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float vc = wave_sum(gs_c[f]);
                if (lane == 0 && vc != 0.f) atomicAdd(gs_s + k * 8 + f * 2 + 0, (double)vc);
                if (M > 0 && !is_joint) {
                    const float vi = wave_sum(gs_i[f]), vj = wave_sum(gs_j[f]);
                    if (lane == 0 && vi != 0.f) atomicAdd(gs_s + k * 8 + f * 2 + 1, (double)vi);
                    if (lane == 0 && vj != 0.f) atomicAdd(gs_s + (NT - 1) * 8 + f * 2 + 1, (double)vj);
                }
            }
            if (is_joint) {
#pragma unroll
                for (int t = 0; t < NJT; ++t) gJ[t] = P[t];
            } else if (iv) {
                float* m1 = a.M1[k];
#pragma unroll
                for (int t = 0; t < NJT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + t * 32 + mfma32_row(r, h);
                        if (j < A) m1[(size_t)j * ns + (my_i - a.i_lo)] = P[t][r];
                    }
            }
        }
    }
    if (BWD && NT > 1 && iv) {
        float* m1 = a.M1[NT - 1];
#pragma unroll
        for (int t = 0; t < NJT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + t * 32 + mfma32_row(r, h);
                if (j < A) m1[(size_t)j * ns + (my_i - a.i_lo)] = gJ[t][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// dZ anchor rows from the dL/dS stash (M1[j*A + i] = dL/dS[i,j]) without LDS:
//   TRANS = 1 :  dX1[i, :] += sum_j M1[j, i] X2[j, :]      (A operand column-read: coalesced along i)
//   TRANS = 0 :  dX2[j, :] += sum_i M1[j, i] X1[i, :]      (A operand row-read: float4 along i)
// One wave owns a 32-row output block and NCT 32-column tiles; both MFMA operands are loaded straight from
// global/L2 in fragment order (the B rows X[k, :] are shared by every wave and stay L2/L1 resident), K is split
// across blockIdx.y and the partial tiles are added atomically into the zero-initialised dZ rows.
// ------------------------------------------------------------------------------------------------
template <int NCT, bool TRANS>
__global__ __launch_bounds__(256) void stash_gemm_kernel(const float* __restrict__ M1, const float* __restrict__ X,
                                                         float* __restrict__ out, int MR, int KR, int ldm, int ld, int Dp, int k_per_split) {
    // out[MR rows] += op(M1)[MR, KR] X[KR rows];  op(M1)[m,k] = TRANS ? M1[k*ldm + m] : M1[m*ldm + k].
    // X / out point at the first column of this launch's column block; rows are `ld` floats apart, Dp columns are valid
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    if (m0 >= MR) return;
    const int kbeg = blockIdx.y * k_per_split, kend = min(KR, kbeg + k_per_split);
    const int m = min(m0 + l31, MR - 1);                // clamped rows are computed but never stored
    f32x16 acc[NCT];
    zero_acc<NCT>(acc);
    int ncol[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) ncol[ct] = min(ct * 32 + l31, Dp - 1);
    for (int k0 = kbeg; k0 < kend; k0 += 8) {           // A, k_per_split multiples of 8 are not required: tail clamps + masks
        float av[4];
        float kmask[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + 4 * h + r;
            kmask[r] = k < kend ? 1.f : 0.f;
            const int kc = min(k, KR - 1);
            av[r] = (TRANS ? M1[(size_t)kc * ldm + m] : M1[(size_t)m * ldm + kc]) * kmask[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kc = min(k0 + 4 * h + r, KR - 1);
            const float* xr = X + (size_t)kc * ld;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], xr[ncol[ct]], acc[ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int d = ct * 32 + l31;
        if (d < Dp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + mfma32_row(r, h);
                if (row < MR) atomicAdd(out + (size_t)row * ld + d, acc[ct][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused anchors x anchors kernel for the "joint = fusion of the M tables" case (Dp == 104).
//
// Same math as anchor_kernel, but S_J = sum_m beta_m S_m is derived in registers, so neither the 312-wide joint
// operand nor its dL/dS stash exist: G_m = dL/dS_m + beta_m dL/dS_J is written directly, and Gamma_m = sum dL/dS_J S_m
// (dL/dbeta) is accumulated on the side.  Geometry, chosen from the counters of anchor_kernel (49 % of wave time in
// waitcnt/barrier on single-buffered K-chunk staging, 512 registers -> 1 wave/SIMD):
//   * a workgroup owns 32 anchor rows I for ALL its J tiles: X1_I / X2_I of the M tables (2*M*13 KiB) are staged into
//     LDS once and are the MFMA B operands (ds_read_b128), so "lane = anchor row i";
//   * each of the 4 waves walks its own 32-row J tiles; the J-side operands go straight from global/L2 into MFMA
//     A-operand fragments (one float4 per lane per 4 MFMAs) -- no staging, no barriers in the loop;
//   * all 2*M S tiles of a (I,J) tile stay in registers (96 for M = 3), ~200 VGPRs total -> 2 waves per SIMD, so one
//     wave's transcendental-heavy epilogue overlaps the other's MFMAs / loads.
// ------------------------------------------------------------------------------------------------
struct AnchorMultiArgs {
    int M, A, i_lo, i_hi, nsplit;
    const float* Z[4];
    const float* beta;             // [M]
    const double* sums;            // [(M+1)][8]
    const float* inv;              // [(M+1)][8] = 1/(sums + 1e-9) as floats (inv_sums_kernel): uniform global loads -> SGPRs
    float alpha, kc, ki, itc, iti;
    double* out;                   // fwd: [(M+1) + 2M] (+ slots)
    const float* coef;             // bwd: dL/d(out)
    float* M1[4];                  // bwd: M1[m][j*ns + (i - i_lo)] = dL/dS_m[i,j] (+ beta_m dL/dS_J)
    double* gs;                    // bwd: [(M+1)][8] (+ slots)
    double* gamma;                 // bwd: [M] (+ slots)
    int j_lo;                      // bwd: first column (a multiple of 16); stash rows are j - j_lo.  0 except in the symmetric mode
    float* M2[4];                  // symmetric mode: M2[m][(j - mir)*ns + (i - i_lo)] = the MIRRORED coefficient dL/dS_m[j,i], j >= mir
    int j_hi, mir;                 // symmetric mode: columns [j_lo, j_hi); tiles at j >= mir also produce the mirrored element (one GPU: A, i_hi)
};

template <int M, bool BWD>
__global__ __launch_bounds__(CT_THREADS, M <= 3 ? 2 : 1) void anchor_multi_kernel(AnchorMultiArgs a) {
    constexpr int DP = 104, NQ = 13, NT = M + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [M][2][32][DP] own rows + inv_s[NT*8]
    float* inv_s = lds + M * 2 * 32 * DP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int A = a.A, ns = a.i_hi - a.i_lo;
    const int ib = blockIdx.x / a.nsplit, split = blockIdx.x % a.nsplit;
    const int i0 = a.i_lo + ib * 32;
    const int my_i = i0 + l31;
    const bool iv = my_i < a.i_hi;

    // ---- stage the I block (rows past the shard end are clamped; masked in the epilogue)
    for (int e = tid; e < M * 2 * 32 * (DP / 4); e += CT_THREADS) {
        const int c = (e % (DP / 4)) * 4, r = (e / (DP / 4)) % 32, side = (e / (DP / 4) / 32) % 2, m = e / (DP / 4) / 64;
        const int row = min(i0 + r, a.i_hi - 1) + side * A;
        *reinterpret_cast<f32x4*>(lds + ((m * 2 + side) * 32 + r) * DP + c) = *reinterpret_cast<const f32x4*>(a.Z[m] + (size_t)row * DP + c);
    }
    for (int e = tid; e < NT * 8; e += CT_THREADS) inv_s[e] = (float)(1.0 / (a.sums[e] + 1e-9));
    __syncthreads();
    float beta[M];
#pragma unroll
    for (int m = 0; m < M; ++m) beta[m] = a.beta[m];

    // per-lane partial sums: fp32 within a tile (16 elements), fp64 across this wave's tiles.  (All-fp32 partials lost 1.7e-4 of the IAL
    // terms at configs[2] -- 19 456 nearly equal addends per lane round with a bias, not a random walk; tools/dbg/aa_check64.py.)
    double acc_d[NT + 2 * M];
#pragma unroll
    for (int e = 0; e < NT + 2 * M; ++e) acc_d[e] = 0.0;

    const int ntile = (A + 31) / 32;
    for (int jt = split * 4 + wave; jt < ntile; jt += a.nsplit * 4) {
        const int j0 = jt * 32;
        float acc_out[NT + 2 * M];
#pragma unroll
        for (int e = 0; e < NT + 2 * M; ++e) acc_out[e] = 0.f;
        const int jrow = min(j0 + l31, A - 1);                       // this lane's J row as an MFMA A-operand row
        // ---- S tiles: P[m][r] = S_m[i = lane, j = j0 + row(r,h)],  Q[m][r] = S_m[j, i]
        f32x16 P[M], Q[M];
        zero_acc<M>(P);
        zero_acc<M>(Q);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float* gp = a.Z[m] + (size_t)(A + jrow) * DP + 4 * h;      // X2[j] for P
            const float* gq = a.Z[m] + (size_t)jrow * DP + 4 * h;            // X1[j] for Q
            const float* bp = lds + ((m * 2 + 0) * 32 + l31) * DP + 4 * h;   // X1[i]
            const float* bq = lds + ((m * 2 + 1) * 32 + l31) * DP + 4 * h;   // X2[i]
            // J-side fragments are prefetched exactly one K-group ahead into the other of two register pairs; the
            // sched_barrier per group stops the scheduler from hoisting all 2*13 global loads of the table (spills).
            f32x4 apA = *reinterpret_cast<const f32x4*>(gp), aqA = *reinterpret_cast<const f32x4*>(gq), apB = apA, aqB = aqA;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
                    if (q & 1) { apA = *reinterpret_cast<const f32x4*>(gp + 8 * (q + 1)); aqA = *reinterpret_cast<const f32x4*>(gq + 8 * (q + 1)); }
                    else { apB = *reinterpret_cast<const f32x4*>(gp + 8 * (q + 1)); aqB = *reinterpret_cast<const f32x4*>(gq + 8 * (q + 1)); }
                }
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 8 * q);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(bq + 8 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    P[m] = __builtin_amdgcn_mfma_f32_32x32x2f32((q & 1) ? apB[r] : apA[r], b1[r], P[m], 0, 0, 0);
                    Q[m] = __builtin_amdgcn_mfma_f32_32x32x2f32((q & 1) ? aqB[r] : aqA[r], b2[r], Q[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // From here on the S tiles are handled as SCALARS: in-place element updates of the 16-wide accumulator vectors
        // (Q[m][r] = ...) make hipcc keep several versions of each vector alive -- >6 KB of scratch per lane.
        float xs[M][16], ys[M][16];
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) { xs[m][r] = P[m][r]; ys[m][r] = Q[m][r]; }
        // ---- epilogue, one element at a time (forward)
        const float* js = inv_s + M * 8;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + mfma32_row(r, h);
            const bool ok = iv && j < A;
            float xj = 0.f, yj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) { xj = fmaf(beta[m], xs[m][r], xj); yj = fmaf(beta[m], ys[m][r], yj); }
            {
                const float dji = fexp2(xj * a.ki);
                const float qma = g_val(dji, js[1], js[3]), qmb = g_val(dji, js[5], js[7]);
                const float lqma = flog(qma), lqmb = flog(qmb);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const float x = k < M ? xs[k < M ? k : 0][r] : xj, y = k < M ? ys[k < M ? k : 0][r] : yj;
                    const float* is = inv_s + k * 8;
                    const float qa = g_val(fexp2(x * a.kc), is[0], is[2]);
                    const float qb = g_val(fexp2(y * a.kc), is[4], is[6]);
                    const float term = -flog(a.alpha * qa + (1.f - a.alpha) * qb);
                    acc_out[k] += ok ? term : 0.f;
                    if (k < M) {
                        const float dm = fexp2(x * a.ki);
                        const float qoa = g_val(dm, is[1], is[3]), qob = g_val(dm, is[5], is[7]);
                        const float ta = __expf(qoa) * (qoa - lqma), tb = __expf(qob) * (qob - lqmb);
                        acc_out[NT + (k < M ? k : 0)] += ok ? ta : 0.f;
                        acc_out[NT + M + (k < M ? k : 0)] += ok ? tb : 0.f;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < NT + 2 * M; ++e) asm volatile("" : "+v"(acc_out[e]));   // keep the updates out of the loop latch
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < NT + 2 * M; ++e) acc_d[e] += (double)acc_out[e];
    }
    // ---- flush the wave's partial sums into its slot
    const int slot = my_slot();
#pragma unroll
    for (int e = 0; e < NT + 2 * M; ++e) {
        const double v = wave_sum_d(acc_d[e]);
        if (lane == 0 && v != 0.0) atomicAdd(a.out + (NT + 2 * M) * (1 + slot) + e, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the fused anchors x anchors terms on 16x16x4 MFMA tiles.
// The 32x32 form of this epilogue (16 elements per lane, 4 table-major passes, everything unrolled) is ~25 000
// instructions in one loop body and hipcc's register allocator collapses on it (6 KB/lane of scratch, 50 ms).  With
// v_mfma_f32_16x16x4_f32 a lane holds 4 elements of a 16x16 tile, the J loop stays ROLLED, and the body is 4x smaller:
// no scratch, <= 128 registers.  Same geometry otherwise: 32 anchor rows of all M tables resident in LDS (MFMA B
// operand, "lane & 15 = anchor row"), J-side fragments straight from global/L2, waves = (anchor half, J interleave).
// ------------------------------------------------------------------------------------------------
__global__ void inv_sums_kernel(const double* __restrict__ sums, float* __restrict__ inv, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[i] = (float)(1.0 / (sums[i] + 1e-9));
}

// RB = anchor rows staged per workgroup: 32 (two wave pairs, each walking its own J tiles) for M <= 3; 16 for M = 4, where 32 rows of
// four tables are 106 KiB of LDS = one workgroup per CU (all four waves then share the 16 rows and split the J tiles four ways).
// TERMS: the same launch also accumulates the forward TERM values (what anchor_multi_kernel<M,false> returns): the epilogue already holds
// every q they are made of, so a training step whose dL/d(terms) is known at forward time (ops.FusedContrastiveFn one-pass mode) runs the
// A x A similarities once instead of twice.
// SYM (M <= 3, TERMS): every UNORDERED anchor pair is visited once.  The block's rows [i_lo, i_hi) meet the columns j >= i_lo only; in a
// tile right of the block (j >= i_hi) a lane holds x = S[i,j] and y = S[j,i] anyway, so it also produces the mirrored element (j, i) --
// its terms, its sum gradients and its coefficient dL/dS[j,i], which goes to a second stash M2 -- instead of leaving it to the block that
// owns row j.  The ICL halves of the two elements share every exp2 / g() evaluation and both denominators; the IAL halves are
// independent.  Half the MFMAs and J-operand loads, ~0.78 of the VALU work per pair.  Tiles inside the block's own
// column range (the diagonal square) run the ordinary epilogue.
template <int M, bool TERMS = false, int RB = (M <= 3 ? 32 : 16), bool SYM = false>
__global__ __launch_bounds__(CT_THREADS, 2) void anchor_multi_bwd16_kernel(AnchorMultiArgs a) {
    static_assert(!SYM || TERMS, "symmetric mode: one-pass build");
    constexpr int DP = 104, NT = M + 1, NSUB = RB / 16, TW = 4 / NSUB;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [M][2][RB][DP] own rows
    // The (M+1)*8 sum coefficients and the 3M+1 upstream coefficients are read from global memory at uniform addresses, per element
    // of the epilogue (the compiler re-issues them as vector loads after every stash store: it cannot rule out aliasing).  Measured
    // alternatives, 2048 x 155 648 block: as is 9.50 ms; loaded once before the loop (the compiler turns them into s_loads, 82 SGPRs)
    // 10.43 ms; pinned in SGPRs by readfirstlane 11.05 ms -- two-SGPR-operand VALU forms do not exist on gfx9, so the uniform values
    // cost v_movs in the arithmetic, more than the L1-hit loads they replace (tools/bench_aa.py).  As LDS reads each value cost an
    // address VGPR + a data VGPR and pushed the kernel into scratch.
    const float* __restrict__ inv_s = a.inv;
    auto CF = [&](int e) { return a.coef[e]; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int A = a.A, ns = a.i_hi - a.i_lo;
    const int JH = SYM ? a.j_hi : A;                                 // column end (the symmetric walk of a rank stops where another rank's starts)
    const int ib = blockIdx.x / a.nsplit, split = blockIdx.x % a.nsplit;
    const int i0 = a.i_lo + ib * RB;
    const int ih = wave % NSUB, tw = wave / NSUB;                    // which 16 anchor rows of the block / which share of the J tiles
    const int my_i = i0 + ih * 16 + l15;
    const bool iv = my_i < a.i_hi;

    for (int e = tid; e < M * 2 * RB * (DP / 4); e += CT_THREADS) {
        const int c = (e % (DP / 4)) * 4, r = (e / (DP / 4)) % RB, side = (e / (DP / 4) / RB) % 2, m = e / (DP / 4) / (2 * RB);
        const int row = min(i0 + r, a.i_hi - 1) + side * A;
        *reinterpret_cast<f32x4*>(lds + ((m * 2 + side) * RB + r) * DP + c) = *reinterpret_cast<const f32x4*>(a.Z[m] + (size_t)row * DP + c);
    }
    __syncthreads();
    float beta[M];
#pragma unroll
    for (int m = 0; m < M; ++m) beta[m] = a.beta[m];

    float acc_gs[NT][8], acc_gam[M];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc_gs[k][e] = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) acc_gam[m] = 0.f;
    float acc_out[TERMS ? NT + 2 * M : 1];                           // TERMS: [ICL_0..M | IAL_a 0..M-1 | IAL_b 0..M-1] partial sums
#pragma unroll
    for (int e = 0; e < (TERMS ? NT + 2 * M : 1); ++e) acc_out[e] = 0.f;
    int tiles_done = 0;
    // All running sums are fp32 per lane and leave for the fp64 slots every 32 tiles (<= 128 addends per partial): at configs[2] a lane
    // sees thousands of nearly equal addends, whose fp32 rounding is a bias, not a random walk (1.7e-4 on the IAL terms with
    // whole-sweep fp32 partials; tools/dbg/aa_check64.py).
    const int slot = my_slot();
    auto flush = [&]() {
        if (TERMS) {
#pragma unroll
            for (int e = 0; e < NT + 2 * M; ++e) {
                const float v = wave_sum(acc_out[TERMS ? e : 0]);
                if (lane == 0 && v != 0.f) atomicAdd(a.out + (NT + 2 * M) * (1 + slot) + e, (double)v);
                acc_out[TERMS ? e : 0] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float iv2 = inv_s[k * 8 + e];
                const float v = -iv2 * iv2 * wave_sum(acc_gs[k][e]);          // dg/dsum = -d inv^2 (g/u)^2: the uniform factor, once
                if (lane == 0 && v != 0.f) atomicAdd(a.gs + NT * 8 * (1 + slot) + k * 8 + e, (double)v);
                acc_gs[k][e] = 0.f;
            }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float v = wave_sum(acc_gam[m]);
            if (lane == 0 && v != 0.f) atomicAdd(a.gamma + M * (1 + slot) + m, (double)v);
            acc_gam[m] = 0.f;
        }
    };
    const float* js = inv_s + M * 8;

    const int ntile = (JH + 15) / 16;
#pragma unroll 1
    for (int jt = (a.j_lo >> 4) + split * TW + tw; jt < ntile; jt += a.nsplit * TW) {
        const int j0 = jt * 16;
        const int jrow = min(j0 + l15, A - 1);
        // The anchor-row operands are loop invariant; left alone, LICM parks all M*2*26 of them in registers
        // (156 for M = 3) and the kernel drops to one wave per SIMD with AGPR/scratch spills.  An opaque zero
        // offset keeps the ds_reads inside the loop: ~100 registers, two waves per SIMD hide each other's loads.
        int lofs = 0;
        asm volatile("" : "+v"(lofs));
        f32x4 P[M], Q[M];
        // J-side operands of table m + 1 are requested (all 12 quads + tails) before table m's MFMAs start: a whole table of flight time
        // for loads that come straight from L2 (compiler-scheduled two loads ahead: 12.80 ms per symmetric 2048 x 155 648 block; this: 12.39)
        struct JOps { f32x4 p[6], q[6]; float pt[2], qt[2]; };
        auto jload = [&](int m, JOps& o) {
            const float* gp = a.Z[m] + (size_t)(A + jrow) * DP;              // X2[j] for P
            const float* gq = a.Z[m] + (size_t)jrow * DP;                    // X1[j] for Q
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                o.p[q] = *reinterpret_cast<const f32x4*>(gp + 16 * q + 4 * g);
                o.q[q] = *reinterpret_cast<const f32x4*>(gq + 16 * q + 4 * g);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) { o.pt[t] = gp[96 + 4 * t + g]; o.qt[t] = gq[96 + 4 * t + g]; }
        };
        constexpr bool JDB = true;        // (M = 4, symmetric: 54 VGPRs go to scratch with or without the second buffer -- cold values, 2 reloads per element)
        JOps jb[JDB ? 2 : 1];
        jload(0, jb[0]);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (JDB) { if (m + 1 < M) jload(m + 1, jb[JDB ? (m + 1) & 1 : 0]); }
            else if (m > 0) jload(m, jb[0]);
            __builtin_amdgcn_sched_barrier(0);
            P[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            Q[m] = P[m];
            const JOps& o = jb[JDB ? m & 1 : 0];
            const float* bp = lds + lofs + ((m * 2 + 0) * RB + ih * 16 + l15) * DP;   // X1[i]
            const float* bq = lds + lofs + ((m * 2 + 1) * RB + ih * 16 + l15) * DP;   // X2[i]
#pragma unroll
            for (int q = 0; q < 6; ++q) {                                    // k = 16q + 4g + r
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 16 * q + 4 * g);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(bq + 16 * q + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    P[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.p[q][r], b1[r], P[m], 0, 0, 0);
                    Q[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.q[q][r], b2[r], Q[m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kk = 96 + 4 * t + g;
                P[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.pt[t], bp[kk], P[m], 0, 0, 0);
                Q[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.qt[t], bq[kk], Q[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // P[m][r] = S_m[i = lane&15, j = j0 + 4g + r], Q[m][r] = S_m[j, i].  One element (r) at a time, with a
        // scheduling barrier between elements: interleaving the four independent chains keeps ~4x the temporaries
        // live and pushes the loop into scratch.  Masks are multiplied in (rows/columns past the end are clamped
        // copies of valid rows, so every intermediate is finite) -- selects here become 28 exec-mask branches.
        // Interior tiles (all 16 anchor rows and all 16 columns valid: everything but the last row block / column tile) run the
        // mask-free instantiation -- the ~20 multiplications by okf and the predicated stores are 4 % of this VALU-bound loop.
        const float cJ = CF(M);
        auto epilogue = [&](auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * g + r;
            const bool ok = !MASKED || (iv && (j < JH));
            const float okf = (!MASKED || ok) ? 1.f : 0.f;
            float xj = 0.f, yj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) { xj = fmaf(beta[m], P[m][r], xj); yj = fmaf(beta[m], Q[m][r], yj); }
            float gJ, EA = 0.f, EB = 0.f;
            // joint ICL
            {
                const float dx = fexp2(xj * a.kc), dy = fexp2(yj * a.kc);
                const GP Ax = g_parts(dx, js[0], js[2]), Bx = g_parts(dx, js[4], js[6]);
                const float qAy = g_val(dy, js[0], js[2]), qBy = g_val(dy, js[4], js[6]);
                const float denA = a.alpha * Ax.q + (1.f - a.alpha) * qBy;
                const float wA = okf * (-cJ * a.alpha) * frcp(denA) * dx;      // weight * d
                const float wB = okf * (-cJ * (1.f - a.alpha)) * frcp(a.alpha * qAy + (1.f - a.alpha) * Bx.q) * dx;
                if (TERMS) acc_out[TERMS ? M : 0] = fmaf(okf, -flog(denA), acc_out[TERMS ? M : 0]);     // -log(a qA(x) + (1-a) qB(y)), losses.py:55-57
                gJ = fmaf(wA, Ax.dd, wB * Bx.dd) * a.itc;
                acc_gs[M][0] = fmaf(wA, Ax.p, acc_gs[M][0]); acc_gs[M][2] = fmaf(wA, Ax.r, acc_gs[M][2]);
                acc_gs[M][4] = fmaf(wB, Bx.p, acc_gs[M][4]); acc_gs[M][6] = fmaf(wB, Bx.r, acc_gs[M][6]);
            }
            // joint IAL reference distribution (qm), shared by every modality
            const float dji = fexp2(xj * a.ki);
            const GP MA = g_parts(dji, js[1], js[3]), MB = g_parts(dji, js[5], js[7]);
            const float lqma = flog(MA.q), lqmb = flog(MB.q);
            float gx[M];
            // per modality ICL + IAL (qo part)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float* is = inv_s + m * 8;
                const float c = CF(m), ca = CF(NT + m), cb = CF(NT + M + m);
                const float x = P[m][r], y = Q[m][r];
                const float dx = fexp2(x * a.kc), dy = fexp2(y * a.kc);
                const GP Ax = g_parts(dx, is[0], is[2]), Bx = g_parts(dx, is[4], is[6]);
                const float qAy = g_val(dy, is[0], is[2]), qBy = g_val(dy, is[4], is[6]);
                const float denA = a.alpha * Ax.q + (1.f - a.alpha) * qBy;
                const float wA = okf * (-c * a.alpha) * frcp(denA) * dx;
                const float wB = okf * (-c * (1.f - a.alpha)) * frcp(a.alpha * qAy + (1.f - a.alpha) * Bx.q) * dx;
                if (TERMS) acc_out[TERMS ? m : 0] = fmaf(okf, -flog(denA), acc_out[TERMS ? m : 0]);
                float gxm = fmaf(wA, Ax.dd, wB * Bx.dd) * a.itc;
                acc_gs[m][0] = fmaf(wA, Ax.p, acc_gs[m][0]); acc_gs[m][2] = fmaf(wA, Ax.r, acc_gs[m][2]);
                acc_gs[m][4] = fmaf(wB, Bx.p, acc_gs[m][4]); acc_gs[m][6] = fmaf(wB, Bx.r, acc_gs[m][6]);
                const float dm = fexp2(x * a.ki);
                const GP OA = g_parts(dm, is[1], is[3]), OB = g_parts(dm, is[5], is[7]);
                const float xA = okf * __expf(OA.q), xB = okf * __expf(OB.q);
                const float eA = ca * xA, eB = cb * xB;
                if (TERMS) {                                                       // exp(qo) (qo - log qm): KLDiv with log_target, losses.py:90-94
                    acc_out[TERMS ? NT + m : 0] = fmaf(xA, OA.q - lqma, acc_out[TERMS ? NT + m : 0]);
                    acc_out[TERMS ? NT + M + m : 0] = fmaf(xB, OB.q - lqmb, acc_out[TERMS ? NT + M + m : 0]);
                }
                const float tA = eA * (OA.q - lqma + 1.f) * dm, tB = eB * (OB.q - lqmb + 1.f) * dm;
                gxm = fmaf(fmaf(tA, OA.dd, tB * OB.dd), a.iti, gxm);
                acc_gs[m][1] = fmaf(tA, OA.p, acc_gs[m][1]); acc_gs[m][3] = fmaf(tA, OA.r, acc_gs[m][3]);
                acc_gs[m][5] = fmaf(tB, OB.p, acc_gs[m][5]); acc_gs[m][7] = fmaf(tB, OB.r, acc_gs[m][7]);
                EA += eA; EB += eB;
                gx[m] = gxm;
            }
            // joint IAL (qm part), totals + stash
            {
                const float uA = -EA * frcp(MA.q) * dji, uB = -EB * frcp(MB.q) * dji;
                gJ = fmaf(fmaf(uA, MA.dd, uB * MB.dd), a.iti, gJ);
                acc_gs[M][1] = fmaf(uA, MA.p, acc_gs[M][1]); acc_gs[M][3] = fmaf(uA, MA.r, acc_gs[M][3]);
                acc_gs[M][5] = fmaf(uB, MB.p, acc_gs[M][5]); acc_gs[M][7] = fmaf(uB, MB.r, acc_gs[M][7]);
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                acc_gam[m] = fmaf(gJ, P[m][r], acc_gam[m]);
#ifdef SGA_DBG_AA_NOSTORE
                acc_gam[m] += fmaf(beta[m], gJ, gx[m]);
#else
                if (ok) a.M1[m][(size_t)(j - a.j_lo) * ns + (my_i - a.i_lo)] = fmaf(beta[m], gJ, gx[m]);
#endif
            }
            // pin the running sums here: otherwise their updates are sunk into the loop latch (they are only
            // consumed by the next iteration) and every factor of all four elements stays live until then
#pragma unroll
            for (int k = 0; k < NT; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(acc_gs[k][e]));
#pragma unroll
            for (int m = 0; m < M; ++m) asm volatile("" : "+v"(acc_gam[m]));
            if (TERMS) {
#pragma unroll
                for (int e = 0; e < NT + 2 * M; ++e) asm volatile("" : "+v"(acc_out[TERMS ? e : 0]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        };
        // Symmetric epilogue: elements (i, j) [x = P, "dir 0", stash M1] and (j, i) [y = Q, "dir 1", stash M2] together.
        auto epilogue_sym = [&](auto) {      // generic: only instantiated where it is called
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * g + r;
            const bool ok = iv && (j < JH);
            const float okf = ok ? 1.f : 0.f;
            float xj = 0.f, yj = 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) { xj = fmaf(beta[m], P[m][r], xj); yj = fmaf(beta[m], Q[m][r], yj); }
            float gci[NT][2];                                        // ICL part of dL/dx, dL/dy per table (joint = M)
            // ---- ICL, every table and the joint: term(i,j) = -log(a qA(x) + (1-a) qB(y)), term(j,i) = -log(a qA(y) + (1-a) qB(x))
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const float* is = inv_s + k * 8;
                const float x = k < M ? P[k < M ? k : 0][r] : xj, y = k < M ? Q[k < M ? k : 0][r] : yj;
                const float c = CF(k);
                const float dx = fexp2(x * a.kc), dy = fexp2(y * a.kc);
                const GP Ax = g_parts(dx, is[0], is[2]), Bx = g_parts(dx, is[4], is[6]);
                const GP Ay = g_parts(dy, is[0], is[2]), By = g_parts(dy, is[4], is[6]);
                const float den1 = a.alpha * Ax.q + (1.f - a.alpha) * By.q;
                const float den2 = a.alpha * Ay.q + (1.f - a.alpha) * Bx.q;
                const float r1 = okf * -c * frcp(den1), r2 = okf * -c * frcp(den2);
                const float wAx = a.alpha * r1 * dx, wBx = (1.f - a.alpha) * r2 * dx;
                const float wAy = a.alpha * r2 * dy, wBy = (1.f - a.alpha) * r1 * dy;
                acc_out[TERMS ? k : 0] = fmaf(okf, -(flog(den1) + flog(den2)), acc_out[TERMS ? k : 0]);
                gci[k][0] = fmaf(wAx, Ax.dd, wBx * Bx.dd) * a.itc;
                gci[k][1] = fmaf(wAy, Ay.dd, wBy * By.dd) * a.itc;
                acc_gs[k][0] = fmaf(wAx, Ax.p, fmaf(wAy, Ay.p, acc_gs[k][0])); acc_gs[k][2] = fmaf(wAx, Ax.r, fmaf(wAy, Ay.r, acc_gs[k][2]));
                acc_gs[k][4] = fmaf(wBx, Bx.p, fmaf(wBy, By.p, acc_gs[k][4])); acc_gs[k][6] = fmaf(wBx, Bx.r, fmaf(wBy, By.r, acc_gs[k][6]));
#pragma unroll
                for (int e = 0; e < 8; e += 2) asm volatile("" : "+v"(acc_gs[k][e]));
                asm volatile("" : "+v"(acc_out[TERMS ? k : 0]));
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- IAL, one direction at a time (nothing shared between x and y here)
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                const float vj = dir ? yj : xj;
                const float dji = fexp2(vj * a.ki);
                const GP MA = g_parts(dji, js[1], js[3]), MB = g_parts(dji, js[5], js[7]);
                const float lqma = flog(MA.q), lqmb = flog(MB.q);
                float gx[M], EA = 0.f, EB = 0.f;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float* is = inv_s + m * 8;
                    const float ca = CF(NT + m), cb = CF(NT + M + m);
                    const float v = dir ? Q[m][r] : P[m][r];
                    const float dm = fexp2(v * a.ki);
                    const GP OA = g_parts(dm, is[1], is[3]), OB = g_parts(dm, is[5], is[7]);
                    const float xA = okf * __expf(OA.q), xB = okf * __expf(OB.q);
                    const float eA = ca * xA, eB = cb * xB;
                    acc_out[TERMS ? NT + m : 0] = fmaf(xA, OA.q - lqma, acc_out[TERMS ? NT + m : 0]);
                    acc_out[TERMS ? NT + M + m : 0] = fmaf(xB, OB.q - lqmb, acc_out[TERMS ? NT + M + m : 0]);
                    const float tA = eA * (OA.q - lqma + 1.f) * dm, tB = eB * (OB.q - lqmb + 1.f) * dm;
                    gx[m] = fmaf(fmaf(tA, OA.dd, tB * OB.dd), a.iti, gci[m][dir]);
                    acc_gs[m][1] = fmaf(tA, OA.p, acc_gs[m][1]); acc_gs[m][3] = fmaf(tA, OA.r, acc_gs[m][3]);
                    acc_gs[m][5] = fmaf(tB, OB.p, acc_gs[m][5]); acc_gs[m][7] = fmaf(tB, OB.r, acc_gs[m][7]);
                    EA += eA; EB += eB;
                }
                const float uA = -EA * frcp(MA.q) * dji, uB = -EB * frcp(MB.q) * dji;
                const float gJ = fmaf(fmaf(uA, MA.dd, uB * MB.dd), a.iti, gci[M][dir]);
                acc_gs[M][1] = fmaf(uA, MA.p, acc_gs[M][1]); acc_gs[M][3] = fmaf(uA, MA.r, acc_gs[M][3]);
                acc_gs[M][5] = fmaf(uB, MB.p, acc_gs[M][5]); acc_gs[M][7] = fmaf(uB, MB.r, acc_gs[M][7]);
                float* const* dst = dir ? a.M2 : a.M1;
                const size_t off = (size_t)(j - (dir ? a.mir : a.j_lo)) * ns + (my_i - a.i_lo);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    acc_gam[m] = fmaf(gJ, dir ? Q[m][r] : P[m][r], acc_gam[m]);
                    if (ok) dst[m][off] = fmaf(beta[m], gJ, gx[m]);
                }
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int e = 1; e < 8; e += 2) asm volatile("" : "+v"(acc_gs[k][e]));
#pragma unroll
                for (int m = 0; m < M; ++m) asm volatile("" : "+v"(acc_gam[m]));
#pragma unroll
                for (int e = NT; e < NT + 2 * M; ++e) asm volatile("" : "+v"(acc_out[TERMS ? e : 0]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        };
#ifdef SGA_DBG_AA_NOEPI
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_gam[m] += P[m][r] * Q[m][r];
#else
        if constexpr (SYM) {
            if (j0 >= a.mir) epilogue_sym(0); else epilogue(std::true_type{});                               // uniform
        } else {
            if (j0 + 16 <= A && i0 + RB <= a.i_hi) epilogue(std::false_type{}); else epilogue(std::true_type{});   // uniform
        }
#endif
        if ((++tiles_done & (SYM ? 15 : 31)) == 0) flush();          // uniform
    }
    flush();
}

int rows_grid(int R) {
    int g = (R + 3) / 4;
    const int cap = sga_num_cus() * 8;
    return g > cap ? cap : (g < 1 ? 1 : g);
}

}  // namespace

extern "C" int sga_loss_gather(const float* E, int T, int D, const int32_t* idx, int R, float* Z, int Dp, float* nrm,
                               void* stream) {
    (void)T;
    SGA_CHECK_ARG(D >= 1 && Dp >= D && Dp % 8 == 0 && R >= 0, "sga_loss_gather: bad argument (Dp must be a multiple of 8 >= D)");
    if (R == 0) return SGA_OK;
    SGA_CHECK_ARG(E && idx && Z && nrm, "sga_loss_gather: null pointer");
    hipLaunchKernelGGL(gather_normalize_kernel, dim3(rows_grid(R)), dim3(256), 0, static_cast<hipStream_t>(stream), E, D, idx, R, Z, Dp, nrm);
    SGA_CHECK_LAUNCH("sga_loss_gather");
    return SGA_OK;
}

extern "C" int sga_loss_scatter(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int R, int D,
                                int Dp, float* dE, void* stream) {
    SGA_CHECK_ARG(D >= 1 && Dp >= D && R >= 0, "sga_loss_scatter: bad argument");
    if (R == 0) return SGA_OK;
    SGA_CHECK_ARG(dZ && Z && nrm && idx && dE, "sga_loss_scatter: null pointer");
    hipLaunchKernelGGL(scatter_normalize_bwd_kernel, dim3(rows_grid(R)), dim3(256), 0, static_cast<hipStream_t>(stream), dZ, Z, nrm, idx, R, D, Dp, dE);
    SGA_CHECK_LAUNCH("sga_loss_scatter");
    return SGA_OK;
}

// Owner/other row groups of the sweeps.  [a_lo, a_hi) is the anchor shard this process owns (one process per GPU shards
// the anchors; 0..A on a single GPU): anchor-owner groups cover only the shard, negative-owner groups see only the shard's
// anchors as "others" -- summing the ranks' outputs gives the unsharded result.
static void fill_groups(SweepArgs& a, int A, int J1, int J2, bool grad, int a_lo, int a_hi) {
    const int x1 = 0, x2 = A, n1 = 2 * A, n2 = 2 * A + J1, ns = a_hi - a_lo;
    int blk = 0, g = 0;
    auto add = [&](int own0, int nown, SweepSeg s0, SweepSeg s1) {
        if (nown <= 0) return;
        SweepGroup& G = a.grp[g++];
        G.own0 = own0; G.nown = nown; G.blk0 = blk; G.nseg = 2; G.seg[0] = s0; G.seg[1] = s1; G.nsplit = 1;
        blk += (nown + 127) / 128;
    };
    add(x1 + a_lo, ns, SweepSeg{n1, J1, 0}, SweepSeg{n2, J2, 1});       // s11, s12
    add(x2 + a_lo, ns, SweepSeg{n2, J2, 2}, SweepSeg{n1, J1, 3});       // s22, s21
    if (grad) {
        add(n1, J1, SweepSeg{x1 + a_lo, ns, 0}, SweepSeg{x2 + a_lo, ns, 3});
        add(n2, J2, SweepSeg{x1 + a_lo, ns, 1}, SweepSeg{x2 + a_lo, ns, 2});
    }
    a.ngroups = g;
}

static int total_blocks(const SweepArgs& a) {
    int n = 0;
    for (int g = 0; g < a.ngroups; ++g) n += (a.grp[g].nown + 127) / 128;
    return n;
}

extern "C" int sga_loss_neg_sums(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, double* sums8,
                                 void* stream) {
    return sga_loss_neg_sums_shard(Z, Dp, A, J1, J2, tau0, tau1, sums8, 0, A, stream);
}

extern "C" int sga_loss_neg_sums_shard(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, double* sums8,
                                       int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Z && sums8 && Dp % 8 == 0 && A >= 0 && J1 >= 0 && J2 >= 0 && tau0 > 0 && tau1 > 0, "sga_loss_neg_sums: bad argument");
    SGA_CHECK_ARG(a_lo >= 0 && a_lo <= a_hi && a_hi <= A, "sga_loss_neg_sums: anchor shard [%d,%d) outside [0,%d]", a_lo, a_hi, A);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = zero_slots(sums8, 8, s, "sga_loss_neg_sums")) return rc;
    if (A == 0 || a_hi == a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    SweepArgs a{};
    a.Z = Z; a.Dp = Dp; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1;
    a.sums = sums8; a.gs = nullptr; a.dZ = nullptr; a.col0 = 0;
    fill_groups(a, A, J1, J2, false, a_lo, a_hi);
    const int nblk = total_blocks(a);
    const int jt = ((J1 > J2 ? J1 : J2) + 127) / 128;
    int gy = (8 * sga_num_cus() + nblk - 1) / nblk;
    if (gy > jt) gy = jt;
    if (gy < 1) gy = 1;
    if (Dp == 104) launch_sweep_fast<13, false>(a, nblk, gy, s);
    else if (Dp == 128) launch_sweep_fast<16, false>(a, nblk, gy, s);
    else launch_sweep<4, 1, false>(a, nblk, gy, s);
    fold_slots(sums8, 8, s);
    SGA_CHECK_LAUNCH("sga_loss_neg_sums");
    return SGA_OK;
}

extern "C" int sga_loss_neg_grad(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1,
                                 const double* gs8, float* dZ, void* stream) {
    return sga_loss_neg_grad_shard(Z, Dp, A, J1, J2, tau0, tau1, gs8, dZ, 0, A, stream);
}

extern "C" int sga_loss_neg_grad_shard(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1,
                                       const double* gs8, float* dZ, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Z && gs8 && dZ && Dp % 8 == 0 && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_neg_grad: bad argument");
    SGA_CHECK_ARG(a_lo >= 0 && a_lo <= a_hi && a_hi <= A, "sga_loss_neg_grad: anchor shard [%d,%d) outside [0,%d]", a_lo, a_hi, A);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (A == 0 || a_hi == a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    SweepArgs a{};
    a.Z = Z; a.Dp = Dp; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1;
    a.sums = nullptr; a.gs = gs8; a.dZ = dZ;
    fill_groups(a, A, J1, J2, true, a_lo, a_hi);
    const int nblk = total_blocks(a);
    int mx = A > J1 ? A : J1;
    if (J2 > mx) mx = J2;
    int gy = (6 * sga_num_cus() + nblk - 1) / nblk;
    if (Dp <= 128) {
        const int jt = (mx + 127) / 128;
        if (gy > jt) gy = jt;
        if (gy < 1) gy = 1;
        a.col0 = 0;
        if (Dp == 104) launch_sweep_fast<13, true>(a, nblk, gy, s);
        else if (Dp == 128) launch_sweep_fast<16, true>(a, nblk, gy, s);
        else launch_sweep<4, 4, true>(a, nblk, gy, s);
    } else {
        const int jt = (mx + 63) / 64;
        if (gy > jt) gy = jt;
        if (gy < 1) gy = 1;
        for (int col0 = 0; col0 < Dp; col0 += 320) {       // 10 column tiles per pass: one pass for the 300-d joint table
            a.col0 = col0;
            launch_sweep<2, 10, true>(a, nblk, gy, s);
        }
    }
    SGA_CHECK_LAUNCH("sga_loss_neg_grad");
    return SGA_OK;
}

extern "C" size_t sga_loss_neg_grad_wide_floats(int A, int J1, int J2) {
    return (size_t)2 * (size_t)(J1 + J2) * (size_t)((A + 31) / 32 * 32);       // the whole batch in one block; less is allowed
}

extern "C" int sga_loss_neg_grad_wide(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, const double* gs8,
                                      float* dZ, float* stash, size_t stash_floats, void* stream) {
    SGA_CHECK_ARG(Z && gs8 && dZ && stash && Dp % 8 == 0 && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_neg_grad_wide: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int J = J1 + J2;
    if (A == 0 || J == 0) return SGA_OK;
    size_t rows = stash_floats / ((size_t)2 * J);
    if (rows >= (size_t)A) rows = A; else rows = rows / 32 * 32;
    SGA_CHECK_ARG(rows >= 32 || rows == (size_t)A, "sga_loss_neg_grad_wide: workspace holds fewer than 32 anchor rows (%zu floats for J = %d)", stash_floats, J);
    const int n1 = 2 * A, n2 = 2 * A + J1;
    for (int lo = 0; lo < A; lo += (int)rows) {
        const int hi = lo + (int)rows < A ? lo + (int)rows : A, ns = hi - lo;
        CoefArgs a{};
        a.Z = Z; a.Dp = Dp; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1; a.gs = gs8;
        a.ld = ns; a.n1 = n1;
        a.stash[0] = stash; a.stash[1] = stash + (size_t)J * ns;
        const int nb = (ns + 127) / 128;
        a.grp[0] = SweepGroup{lo, ns, 0, 2, {SweepSeg{n1, J1, 0}, SweepSeg{n2, J2, 1}}, 1};            // X1 anchors: s11, s12
        a.grp[1] = SweepGroup{A + lo, ns, nb, 2, {SweepSeg{n2, J2, 2}, SweepSeg{n1, J1, 3}}, 1};       // X2 anchors: s22, s21
        const int mx = J1 > J2 ? J1 : J2;
        int gy = (6 * sga_num_cus() + 2 * nb - 1) / (2 * nb);
        const int jt = (mx + 63) / 64;
        if (gy > jt) gy = jt;
        if (gy < 1) gy = 1;
        hipLaunchKernelGGL(sweep_coef_kernel<2>, dim3(2 * nb, gy), dim3(CT_THREADS), 0, s, a);
        SGA_CHECK_LAUNCH("sga_loss_neg_grad_wide");
        for (int g = 0; g < 2; ++g) {
            const float* C = a.stash[g];
            const size_t own_row = (size_t)(g == 0 ? lo : A + lo);
            // dZ[anchors of the block] += Ct^T Z[negatives]
            int rc = sga_gemm(1, 0, ns, Dp, J, C, ns, 0, Z + (size_t)n1 * Dp, Dp, dZ + own_row * Dp, Dp, nullptr, 1, stream);
            if (rc) return rc;
            // dZ[negatives] += Ct Z[anchors of the block]
            rc = sga_gemm(0, 0, J, Dp, ns, C, ns, 0, Z + own_row * Dp, Dp, dZ + (size_t)n1 * Dp, Dp, nullptr, 1, stream);
            if (rc) return rc;
        }
    }
    return SGA_OK;
}

static int fill_anchor(AnchorArgs& a, const float* const* Z, const int* Dp, int NT, int A, const double* sums,
                       float alpha, float tau_icl, float tau_ial, int a_lo, int a_hi) {
    if (NT < 1 || NT > CT_MAXT) { sga_set_error("sga_loss_anchor: NT=%d outside [1,%d]", NT, CT_MAXT); return SGA_ERR_ARG; }
    if (a_lo < 0 || a_hi > A || a_lo > a_hi) { sga_set_error("sga_loss_anchor: anchor shard [%d,%d) outside [0,%d]", a_lo, a_hi, A); return SGA_ERR_ARG; }
    a.NT = NT; a.A = A; a.sums = sums; a.alpha = alpha; a.i_lo = a_lo; a.i_hi = a_hi;
    a.kc = LOG2E / tau_icl; a.ki = LOG2E / tau_ial; a.itc = 1.f / tau_icl; a.iti = 1.f / tau_ial;
    for (int k = 0; k < NT; ++k) {
        if (!Z[k] || Dp[k] % 8) { sga_set_error("sga_loss_anchor: table %d null or Dp %% 8 != 0", k); return SGA_ERR_ARG; }
        a.Z[k] = Z[k]; a.Dp[k] = Dp[k];
    }
    return SGA_OK;
}

extern "C" int sga_loss_anchor_fwd(const float* const* Z, const int* Dp, int NT, int A, const double* sums,
                                   float alpha, float tau_icl, float tau_ial, double* out, int a_lo, int a_hi, void* stream) {
    return sga_loss_anchor_fwd_f16(Z, nullptr, Dp, NT, A, sums, alpha, tau_icl, tau_ial, out, a_lo, a_hi, nullptr, 0, stream);
}

// A workspace given (the caller's choice: wide tables): the 2 NT similarity blocks of the anchor shard are formed first -- tables with an
// fp16 copy Zh[k] on wide16.hip's fp16 tile core (up to 8 blocks per launch), the others by the exact-fp32 NT GEMM of gemm.hip -- and the
// epilogue-only form of the kernel reads them.
static size_t anchor_ws_ldp(int ns) { return (size_t)(ns + 3) / 4 * 4; }
extern "C" size_t sga_loss_anchor_f16_ws_bytes(int NT, int A, int ns) {
    if (NT < 1 || A < 1 || ns < 1) return 256;
    return (size_t)NT * 2 * A * anchor_ws_ldp(ns) * sizeof(float) + 256;
}
static int anchor_pre_blocks(AnchorArgs& a, const void* const* Zh, void* ws, size_t ws_bytes, hipStream_t s, bool& pre) {
    pre = false;
    if (!ws) return SGA_OK;
    const int ns = a.i_hi - a.i_lo, A = a.A;
    if (ws_bytes < sga_loss_anchor_f16_ws_bytes(a.NT, A, ns)) {
        sga_set_error("sga_loss_anchor (f16): workspace of %zu bytes, %zu needed", ws_bytes, sga_loss_anchor_f16_ws_bytes(a.NT, A, ns));
        return SGA_ERR_WORKSPACE;
    }
    const size_t ldp = anchor_ws_ldp(ns);
    float* w = static_cast<float*>(ws);
    SgaW16Store e[8];
    int n = 0;
    for (int k = 0; k < a.NT; ++k) {
        const _Float16* zh = Zh ? static_cast<const _Float16*>(Zh[k]) : nullptr;
        const long dp = a.Dp[k];
        float* sp = w + (size_t)(2 * k) * A * ldp;
        float* sq = w + (size_t)(2 * k + 1) * A * ldp;
        a.SP[k] = sp; a.SQ[k] = sq;
        if (zh) {
            e[n++] = SgaW16Store{zh + (size_t)A * dp, dp, A, zh + (size_t)a.i_lo * dp, dp, ns, (int)dp, sp, (long)ldp};      // X2[j] . X1[i]
            e[n++] = SgaW16Store{zh, dp, A, zh + (size_t)(A + a.i_lo) * dp, dp, ns, (int)dp, sq, (long)ldp};                  // X1[j] . X2[i]
        } else {
            const float* z = a.Z[k];
            if (int rc = sga_gemm(0, 1, A, ns, (int)dp, z + (size_t)A * dp, dp, 0, z + (size_t)a.i_lo * dp, dp, sp, (long)ldp, nullptr, 0, s)) return rc;
            if (int rc = sga_gemm(0, 1, A, ns, (int)dp, z, dp, 0, z + (size_t)(A + a.i_lo) * dp, dp, sq, (long)ldp, nullptr, 0, s)) return rc;
        }
        if (n == 8 || (k == a.NT - 1 && n > 0)) {
            if (int rc = sga_wide16_store_batch(e, n, s)) return rc;
            n = 0;
        }
    }
    a.ldp = (long)ldp;
    pre = true;
    return SGA_OK;
}

extern "C" int sga_loss_anchor_fwd_f16(const float* const* Z, const void* const* Zh, const int* Dp, int NT, int A, const double* sums,
                                       float alpha, float tau_icl, float tau_ial, double* out, int a_lo, int a_hi, void* ws, size_t ws_bytes,
                                       void* stream) {
    SGA_CHECK_ARG(Z && Dp && sums && out && A >= 0, "sga_loss_anchor_fwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = NT > 1 ? NT - 1 : 0;
    if (int rc0 = zero_slots(out, NT + 2 * M, s, "sga_loss_anchor_fwd")) return rc0;
    if (A == 0 || a_hi <= a_lo) return SGA_OK;
    AnchorArgs a{};
    int rc = fill_anchor(a, Z, Dp, NT, A, sums, alpha, tau_icl, tau_ial, a_lo, a_hi);
    if (rc) return rc;
    a.out = out;
    for (int k = 0; k < NT; ++k) a.Zh[k] = Zh ? static_cast<const _Float16*>(Zh[k]) : nullptr;
    bool pre = false;
    if (int rcp = anchor_pre_blocks(a, Zh, ws, ws_bytes, s, pre)) return rcp;
    if (pre) hipLaunchKernelGGL((anchor_kernel<false, true>), dim3((a_hi - a_lo + 127) / 128, (A + 31) / 32), dim3(CT_THREADS), 0, s, a);
    else hipLaunchKernelGGL(anchor_kernel<false>, dim3((a_hi - a_lo + 127) / 128, (A + 63) / 64), dim3(CT_THREADS), 0, s, a);
    fold_slots(out, NT + 2 * M, s);
    SGA_CHECK_LAUNCH("sga_loss_anchor_fwd");
    return SGA_OK;
}

extern "C" int sga_loss_anchor_bwd(const float* const* Z, const int* Dp, int NT, int A, const double* sums,
                                   float alpha, float tau_icl, float tau_ial, const float* coef, float* const* M1,
                                   double* gs, int a_lo, int a_hi, void* stream) {
    return sga_loss_anchor_bwd_f16(Z, nullptr, Dp, NT, A, sums, alpha, tau_icl, tau_ial, coef, M1, gs, a_lo, a_hi, nullptr, 0, stream);
}

extern "C" int sga_loss_anchor_bwd_f16(const float* const* Z, const void* const* Zh, const int* Dp, int NT, int A, const double* sums,
                                       float alpha, float tau_icl, float tau_ial, const float* coef, float* const* M1,
                                       double* gs, int a_lo, int a_hi, void* ws, size_t ws_bytes, void* stream) {
    SGA_CHECK_ARG(Z && Dp && sums && coef && M1 && gs && A >= 0, "sga_loss_anchor_bwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(gs, NT * 8, s, "sga_loss_anchor_bwd")) return rc0;
    if (A == 0 || a_hi <= a_lo) return SGA_OK;
    AnchorArgs a{};
    int rc = fill_anchor(a, Z, Dp, NT, A, sums, alpha, tau_icl, tau_ial, a_lo, a_hi);
    if (rc) return rc;
    a.coef = coef; a.gs = gs;
    for (int k = 0; k < NT; ++k) { SGA_CHECK_ARG(M1[k], "sga_loss_anchor_bwd: null stash %d", k); a.M1[k] = M1[k]; }
    for (int k = 0; k < NT; ++k) a.Zh[k] = Zh ? static_cast<const _Float16*>(Zh[k]) : nullptr;
    bool pre = false;
    if (int rcp = anchor_pre_blocks(a, Zh, ws, ws_bytes, s, pre)) return rcp;
    if (pre) hipLaunchKernelGGL((anchor_kernel<true, true>), dim3((a_hi - a_lo + 127) / 128, (A + 31) / 32), dim3(CT_THREADS), 0, s, a);
    else hipLaunchKernelGGL(anchor_kernel<true>, dim3((a_hi - a_lo + 127) / 128, (A + 63) / 64), dim3(CT_THREADS), 0, s, a);
    fold_slots(gs, NT * 8, s);
    SGA_CHECK_LAUNCH("sga_loss_anchor_bwd");
    return SGA_OK;
}

// ---- fused multi-table entry points (joint table == fusion of the M tables) ----------------------------
static int fill_multi(MultiArgs& a, const float* const* Z, int M, int D, const float* beta, int A, int J1, int J2, float tau0,
                      float tau1, bool grad, int a_lo, int a_hi) {
    if (D < 1 || D > 104) { sga_set_error("sga_loss_multi: D=%d outside [1,104] (the fused sweeps take Dp = 104 tables)", D); return SGA_ERR_ARG; }
    a.ktail = D > 100 ? 2 : (D > 96 ? 1 : 0);
    if (a_lo < 0 || a_hi > A || a_lo > a_hi) { sga_set_error("sga_loss_multi: anchor shard [%d,%d) outside [0,%d]", a_lo, a_hi, A); return SGA_ERR_ARG; }
    if (M < 2 || M > 4) { sga_set_error("sga_loss_multi: M=%d outside [2,4]", M); return SGA_ERR_ARG; }
    a.M = M;
    for (int m = 0; m < M; ++m) { if (!Z[m]) { sga_set_error("sga_loss_multi: null table"); return SGA_ERR_ARG; } a.Z[m] = Z[m]; }
    a.beta = beta; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1;
    SweepArgs tmp{};
    fill_groups(tmp, A, J1, J2, grad, a_lo, a_hi);
    a.ngroups = tmp.ngroups;
    for (int g = 0; g < 4; ++g) a.grp[g] = tmp.grp[g];
    return SGA_OK;
}
// Split every group's other-tile list so that all workgroups run ~`target` 32-row steps: uniform work units keep
// the 256 CUs busy to the end (anchor-owner blocks see 2.4x more other rows than negative-owner blocks).
static int plan_multi(MultiArgs& a, int target_steps, int own_rows = 128) {
    int nwg = 0;
    for (int g = 0; g < a.ngroups; ++g) {
        SweepGroup& G = a.grp[g];
        int steps = 0;
        for (int sg = 0; sg < G.nseg; ++sg) steps += (G.seg[sg].n + 31) / 32;
        int ns = (steps + target_steps - 1) / target_steps;
        if (ns < 1) ns = 1;
        G.nsplit = ns;
        G.blk0 = nwg;
        nwg += ((((G.nown + own_rows - 1) / own_rows) * ns + 7) / 8) * 8;      // 8 per-XCD chunks (sweep16_kernel's work order)
    }
    return nwg;
}

static int multi_sums_impl(const float* const* Z, int M, int D, bool centred, const float* beta, int A, int J1, int J2, float tau0,
                           float tau1, double* sums, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Z && beta && sums && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_sums: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(sums, (M + 1) * 8, s, "sga_loss_multi_sums")) return rc0;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    MultiArgs a{};
    int rc = fill_multi(a, Z, M, D, beta, A, J1, J2, tau0, tau1, false, a_lo, a_hi);
    if (rc) return rc;
    a.swap_tail = centred ? 1 : 0;
    a.sums = sums;
    const int nwg = plan_multi(a, 160, S16_OWN);
    if (M == 2) launch_sweep16<2, false>(a, nwg, s);
    else if (M == 3) launch_sweep16<3, false>(a, nwg, s);
    else launch_sweep16x2<false>(a, nwg, s);
    fold_slots(sums, (M + 1) * 8, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_sums");
    return SGA_OK;
}
extern "C" int sga_loss_multi_sums(const float* const* Z, int M, int D, const float* beta, int A, int J1, int J2, float tau0,
                                   float tau1, double* sums, int a_lo, int a_hi, void* stream) {
    return multi_sums_impl(Z, M, D, false, beta, A, J1, J2, tau0, tau1, sums, a_lo, a_hi, stream);
}
// The same sweeps over CENTRED tables Zc (sga_loss_centre_tables: 100 data columns of z - zbar, column 100 = b, 101 = 1): identical
// similarities (the owner's swapped K tail adds b_i + b_j), gradient delivered in two parts -- dZ[:, 0..99] = sum c (z - zbar),
// dZ[:, 101] = sum c -- for sga_loss_scatter_tangent_stat.
extern "C" int sga_loss_multi_sums_centred(const float* const* Zc, int M, const float* beta, int A, int J1, int J2, float tau0,
                                           float tau1, double* sums, int a_lo, int a_hi, void* stream) {
    return multi_sums_impl(Zc, M, 102, true, beta, A, J1, J2, tau0, tau1, sums, a_lo, a_hi, stream);
}

static int multi_grad_impl(const float* const* Z, int M, int D, bool centred, const float* beta, int A, int J1, int J2, float tau0,
                           float tau1, const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi,
                           void* stream) {
    SGA_CHECK_ARG(Z && beta && gs && dZ && gamma && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_grad: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rcz = zero_slots(gamma, M > 0 ? M : 1, s, "sga_loss_multi_grad")) return rcz;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    MultiArgs a{};
    int rc = fill_multi(a, Z, M, D, beta, A, J1, J2, tau0, tau1, true, a_lo, a_hi);
    if (rc) return rc;
    a.swap_tail = centred ? 1 : 0;
    a.gs = gs; a.gamma = gamma;
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(dZ[m], "sga_loss_multi_grad: null dZ"); a.dZ[m] = dZ[m]; }
    const int nwg = plan_multi(a, 160, S16_OWN);
    if (M == 2) launch_sweep16<2, true>(a, nwg, s);
    else if (M == 3) launch_sweep16<3, true>(a, nwg, s);
    else launch_sweep16x2<true>(a, nwg, s);            // paired waves: two tables per wave, 8 waves per workgroup
    fold_slots(gamma, M, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_grad");
    return SGA_OK;
}
extern "C" int sga_loss_multi_grad(const float* const* Z, int M, int D, const float* beta, int A, int J1, int J2, float tau0,
                                   float tau1, const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi,
                                   void* stream) {
    return multi_grad_impl(Z, M, D, false, beta, A, J1, J2, tau0, tau1, gs, dZ, gamma, a_lo, a_hi, stream);
}
extern "C" int sga_loss_multi_grad_centred(const float* const* Zc, int M, const float* beta, int A, int J1, int J2, float tau0,
                                           float tau1, const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi,
                                           void* stream) {
    return multi_grad_impl(Zc, M, 102, true, beta, A, J1, J2, tau0, tau1, gs, dZ, gamma, a_lo, a_hi, stream);
}

extern "C" int sga_loss_build_joint(const float* const* Z, int M, const float* beta, int rows, float* ZJ, void* stream) {
    SGA_CHECK_ARG(Z && beta && ZJ && M >= 2 && M <= 4 && rows >= 0, "sga_loss_build_joint: bad argument");
    if (rows == 0) return SGA_OK;
    MultiArgs a{};
    a.M = M; a.beta = beta;
    for (int m = 0; m < M; ++m) a.Z[m] = Z[m];
    hipLaunchKernelGGL(build_joint_kernel, dim3(rows_grid(rows)), dim3(256), 0, static_cast<hipStream_t>(stream), a, ZJ, rows);
    SGA_CHECK_LAUNCH("sga_loss_build_joint");
    return SGA_OK;
}

extern "C" int sga_loss_fold_joint(const float* const* Z, int M, const float* beta, const float* dZJ, int rows,
                                   float* const* dZ, double* gamma2, void* stream) {
    SGA_CHECK_ARG(Z && beta && dZJ && dZ && gamma2 && M >= 2 && M <= 4 && rows >= 0, "sga_loss_fold_joint: bad argument");
    if (rows == 0) return SGA_OK;
    MultiArgs a{};
    a.M = M; a.beta = beta;
    for (int m = 0; m < M; ++m) { a.Z[m] = Z[m]; a.dZ[m] = dZ[m]; }
    hipLaunchKernelGGL(fold_joint_kernel, dim3(rows_grid(rows)), dim3(256), 0, static_cast<hipStream_t>(stream), a, dZJ, rows, gamma2);
    SGA_CHECK_LAUNCH("sga_loss_fold_joint");
    return SGA_OK;
}

extern "C" int sga_loss_check_norms(const float* nrm, int n, float* poison, void* stream) {
    SGA_CHECK_ARG(nrm && poison && n >= 0, "sga_loss_check_norms: bad argument");
    if (n == 0) return SGA_OK;
    hipLaunchKernelGGL(check_norms_kernel, dim3(64), dim3(256), 0, static_cast<hipStream_t>(stream), nrm, n, poison);
    SGA_CHECK_LAUNCH("sga_loss_check_norms");
    return SGA_OK;
}

extern "C" int sga_loss_slots(void) { return SGA_SLOTS; }

/* For one table (Z = [X1 | X2 | ...] rows of width Dp) and the anchor shard [a_lo, a_hi) that produced M1 [A, a_hi-a_lo]:
 * dZ[a_lo:a_hi] += M1^T X2   and   dZ[A:2A] += M1 X1[a_lo:a_hi] */
static void launch_stash(bool trans, int nct10, const float* M1, const float* X, float* out, int MR, int KR, int ldm, int ld,
                         int w, hipStream_t s) {
    const int gx = (MR + 127) / 128;
    int splits = (6 * sga_num_cus() + gx - 1) / gx;
    int kper = ((KR + splits - 1) / splits + 7) / 8 * 8;
    if (kper < 64) kper = 64;
    splits = (KR + kper - 1) / kper;
    dim3 grid(gx, splits), blk(256);
    if (trans) {
        if (nct10) hipLaunchKernelGGL((stash_gemm_kernel<10, true>), grid, blk, 0, s, M1, X, out, MR, KR, ldm, ld, w, kper);
        else hipLaunchKernelGGL((stash_gemm_kernel<4, true>), grid, blk, 0, s, M1, X, out, MR, KR, ldm, ld, w, kper);
    } else {
        if (nct10) hipLaunchKernelGGL((stash_gemm_kernel<10, false>), grid, blk, 0, s, M1, X, out, MR, KR, ldm, ld, w, kper);
        else hipLaunchKernelGGL((stash_gemm_kernel<4, false>), grid, blk, 0, s, M1, X, out, MR, KR, ldm, ld, w, kper);
    }
}

extern "C" int sga_loss_stash_grad(const float* M1, const float* Z, int A, int Dp, float* dZ, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(M1 && Z && dZ && A >= 0 && Dp >= 8 && Dp % 8 == 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi, "sga_loss_stash_grad: bad argument");
    const int ns = a_hi - a_lo;
    if (A == 0 || ns == 0) return SGA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* X1 = Z + (size_t)a_lo * Dp;             // the shard's X1 rows
    const float* X2 = Z + (size_t)A * Dp;                // all X2 rows
    if (ns % 4 == 0 && Dp % 4 == 0 && reinterpret_cast<uintptr_t>(M1) % 16 == 0 && reinterpret_cast<uintptr_t>(Z) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(dZ) % 16 == 0) {
        // the two products as plain GEMMs on the stash M1 [A (j), ns (i)] = G^T (gemm.hip: row-major TN / NN kernels,
        // split over the contraction with atomic accumulation into dZ):
        //   dX1[i, :] += sum_j M1[j, i] X2[j, :]      (TN)          dX2[j, :] += sum_i M1[j, i] X1[i, :]      (NN)
        int rc = sga_gemm(1, 0, ns, Dp, A, M1, ns, 0, X2, Dp, dZ + (size_t)a_lo * Dp, Dp, nullptr, 1, stream);
        if (rc) return rc;
        return sga_gemm(0, 0, A, Dp, ns, M1, ns, 0, X1, Dp, dZ + (size_t)A * Dp, Dp, nullptr, 1, stream);
    }
    for (int c0 = 0; c0 < Dp; c0 += 320) {               // column blocks of <= 320 (the 104*M-wide joint operand takes one or two)
        const int w = Dp - c0 < 320 ? Dp - c0 : 320;
        launch_stash(true, w > 128, M1, X2 + c0, dZ + (size_t)a_lo * Dp + c0, ns, A, ns, Dp, w, s);
        launch_stash(false, w > 128, M1, X1 + c0, dZ + (size_t)A * Dp + c0, A, ns, ns, Dp, w, s);
    }
    SGA_CHECK_LAUNCH("sga_loss_stash_grad");
    return SGA_OK;
}

// ---- fused anchors x anchors entry points -------------------------------------------------------------------------
static int fill_anchor_multi(AnchorMultiArgs& a, const float* const* Z, int M, const float* beta, int A, const double* sums,
                             float alpha, float tau_icl, float tau_ial, int a_lo, int a_hi) {
    if (M < 2 || M > 4) { sga_set_error("sga_loss_anchor_multi: M=%d not in {2,3,4} (use the per-table kernels)", M); return SGA_ERR_ARG; }
    if (a_lo < 0 || a_hi > A || a_lo > a_hi) { sga_set_error("sga_loss_anchor_multi: anchor shard [%d,%d) outside [0,%d]", a_lo, a_hi, A); return SGA_ERR_ARG; }
    a.M = M; a.A = A; a.i_lo = a_lo; a.i_hi = a_hi; a.beta = beta; a.sums = sums; a.alpha = alpha;
    a.kc = LOG2E / tau_icl; a.ki = LOG2E / tau_ial; a.itc = 1.f / tau_icl; a.iti = 1.f / tau_ial;
    for (int m = 0; m < M; ++m) { if (!Z[m]) { sga_set_error("sga_loss_anchor_multi: null table"); return SGA_ERR_ARG; } a.Z[m] = Z[m]; }
    const int nib = (a_hi - a_lo + 31) / 32, ntile = (A + 31) / 32;
    int ns = (4 * sga_num_cus() + nib - 1) / (nib > 0 ? nib : 1);
    if (ns > (ntile + 3) / 4) ns = (ntile + 3) / 4;
    if (ns < 1) ns = 1;
    a.nsplit = ns;
    return SGA_OK;
}

template <int M, bool BWD>
static void launch_anchor_multi(const AnchorMultiArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(M * 2 * 32 * 104 + (M + 1) * 8) * sizeof(float);
    auto k = anchor_multi_kernel<M, BWD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nib = (a.i_hi - a.i_lo + 31) / 32;
    hipLaunchKernelGGL(k, dim3(nib * a.nsplit), dim3(CT_THREADS), lds, s, a);
}

extern "C" int sga_loss_anchor_multi_fwd(const float* const* Z, int M, const float* beta, int A, const double* sums,
                                         float alpha, float tau_icl, float tau_ial, double* out, int a_lo, int a_hi,
                                         void* stream) {
    SGA_CHECK_ARG(Z && beta && sums && out && A >= 0, "sga_loss_anchor_multi_fwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n = (M + 1) + 2 * M;
    if (int rc0 = zero_slots(out, n, s, "sga_loss_anchor_multi_fwd")) return rc0;
    if (A == 0 || a_hi <= a_lo) return SGA_OK;
    AnchorMultiArgs a{};
    int rc = fill_anchor_multi(a, Z, M, beta, A, sums, alpha, tau_icl, tau_ial, a_lo, a_hi);
    if (rc) return rc;
    a.out = out;
    if (M == 2) launch_anchor_multi<2, false>(a, s); else if (M == 3) launch_anchor_multi<3, false>(a, s); else launch_anchor_multi<4, false>(a, s);
    fold_slots(out, n, s);
    SGA_CHECK_LAUNCH("sga_loss_anchor_multi_fwd");
    return SGA_OK;
}

extern "C" int sga_loss_anchor_multi_bwd(const float* const* Z, int M, const float* beta, int A, const double* sums,
                                         float alpha, float tau_icl, float tau_ial, const float* coef, float* const* M1,
                                         double* gs, double* gamma, int a_lo, int a_hi, double* out_terms, void* stream) {
    SGA_CHECK_ARG(Z && beta && sums && coef && M1 && gs && gamma && A >= 0, "sga_loss_anchor_multi_bwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(gs, (M + 1) * 8, s, "sga_loss_anchor_multi_bwd")) return rc0;
    if (int rc1 = zero_slots(gamma, M, s, "sga_loss_anchor_multi_bwd")) return rc1;
    if (out_terms) { if (int rc2 = zero_slots(out_terms, (M + 1) + 2 * M, s, "sga_loss_anchor_multi_bwd")) return rc2; }
    if (A == 0 || a_hi <= a_lo) return SGA_OK;
    AnchorMultiArgs a{};
    int rc = fill_anchor_multi(a, Z, M, beta, A, sums, alpha, tau_icl, tau_ial, a_lo, a_hi);
    if (rc) return rc;
    a.coef = coef; a.gs = gs; a.gamma = gamma;
    // float copy of 1/(sums+eps): lives in the block after the gs slots (gs buffers hold (2 + slots) * (M+1)*8 doubles)
    float* inv = reinterpret_cast<float*>(gs + (size_t)(1 + SGA_SLOTS) * (M + 1) * 8);
    hipLaunchKernelGGL(inv_sums_kernel, dim3(1), dim3(64), 0, s, sums, inv, (M + 1) * 8);
    a.inv = inv;
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(M1[m], "sga_loss_anchor_multi_bwd: null stash"); a.M1[m] = M1[m]; }
    {
        const int RB = M <= 3 ? 32 : 16, TW = M <= 3 ? 2 : 4;
        const size_t lds = (size_t)(M * 2 * RB * 104 + (M + 1) * 8) * sizeof(float);
        const int nib = (a.i_hi - a.i_lo + RB - 1) / RB, ntile16 = (A + 15) / 16;
        int nsp = (6 * sga_num_cus() + nib - 1) / (nib > 0 ? nib : 1);
        if (nsp > (ntile16 + TW - 1) / TW) nsp = (ntile16 + TW - 1) / TW;
        if (nsp < 1) nsp = 1;
        a.nsplit = nsp;
        a.out = out_terms;
        auto go = [&](auto k) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(nib * nsp), dim3(CT_THREADS), lds, s, a);
        };
        if (out_terms) { if (M == 2) go(anchor_multi_bwd16_kernel<2, true>); else if (M == 3) go(anchor_multi_bwd16_kernel<3, true>); else go(anchor_multi_bwd16_kernel<4, true>); }
        else { if (M == 2) go(anchor_multi_bwd16_kernel<2, false>); else if (M == 3) go(anchor_multi_bwd16_kernel<3, false>); else go(anchor_multi_bwd16_kernel<4, false>); }
    }
    if (out_terms) fold_slots(out_terms, (M + 1) + 2 * M, s);
    fold_slots(gs, (M + 1) * 8, s);
    fold_slots(gamma, M, s);
    SGA_CHECK_LAUNCH("sga_loss_anchor_multi_bwd");
    return SGA_OK;
}

/* Symmetric form of sga_loss_anchor_multi_bwd for an UNSHARDED anchor set walked in blocks (M = 2, 3; terms always returned): block
 * [a_lo, a_hi) meets the columns j >= a_lo only and also produces the mirrored elements (j, i), j >= a_hi, i in the block, so every
 * unordered pair is evaluated once over the whole walk.  a_lo must be a multiple of 32, a_hi a multiple of 32 or == A.
 *   M1[m][(j - a_lo) * ns + (i - a_lo)] = dL/dS_m[i, j],  j in [a_lo, A)          ([A - a_lo, ns] floats)
 *   M2[m][(j - a_hi) * ns + (i - a_lo)] = dL/dS_m[j, i],  j in [a_hi, A)          ([A - a_hi, ns] floats)
 * out_terms / gs / gamma as in sga_loss_anchor_multi_bwd: this block's share (both elements of every pair it visits). */
static int symx_impl(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                     float tau_icl, float tau_ial, const float* coef, float* const* M1, float* const* M2,
                     double* gs, double* gamma, int a_lo, int a_hi, int j_lo, int j_hi, int mir, double* out_terms,
                     void* stream);

extern "C" int sga_loss_anchor_multi_bwd_symx(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                                              float tau_icl, float tau_ial, const float* coef, float* const* M1, float* const* M2,
                                              double* gs, double* gamma, int a_lo, int a_hi, int j_lo, int j_hi, int mir, double* out_terms,
                                              void* stream) {
    return symx_impl(Z, M, beta, A, sums, alpha, tau_icl, tau_ial, coef, M1, M2, gs, gamma, a_lo, a_hi, j_lo, j_hi, mir, out_terms, stream);
}

static int symx_impl(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                     float tau_icl, float tau_ial, const float* coef, float* const* M1, float* const* M2,
                     double* gs, double* gamma, int a_lo, int a_hi, int j_lo, int j_hi, int mir, double* out_terms,
                     void* stream) {
    SGA_CHECK_ARG(Z && beta && sums && coef && M1 && M2 && gs && gamma && out_terms && A >= 0, "sga_loss_anchor_multi_bwd_symx: bad argument");
    SGA_CHECK_ARG(M >= 2 && M <= 4, "sga_loss_anchor_multi_bwd_symx: M=%d (2, 3 or 4)", M);
    SGA_CHECK_ARG(a_lo % 32 == 0 && (a_hi % 32 == 0 || a_hi == A), "sga_loss_anchor_multi_bwd_symx: block [%d,%d) not on 32-row boundaries", a_lo, a_hi);
    SGA_CHECK_ARG(j_lo >= 0 && j_lo % 16 == 0 && j_hi <= A && j_lo <= j_hi && (j_hi % 16 == 0 || j_hi == A) && mir >= j_lo && (mir % 16 == 0 || mir >= j_hi),
                  "sga_loss_anchor_multi_bwd_symx: columns [%d,%d) / mirror start %d not on 16-column boundaries", j_lo, j_hi, mir);
    // columns left of the mirror start are visited in the ordered way: they must lie in the block's own square
    SGA_CHECK_ARG(mir <= j_lo || (j_lo >= a_lo && (mir < j_hi ? mir : j_hi) <= a_hi), "sga_loss_anchor_multi_bwd_symx: ordered columns [%d,%d) outside the block's square [%d,%d)",
                  j_lo, mir, a_lo, a_hi);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(gs, (M + 1) * 8, s, "sga_loss_anchor_multi_bwd_symx")) return rc0;
    if (int rc1 = zero_slots(gamma, M, s, "sga_loss_anchor_multi_bwd_symx")) return rc1;
    if (int rc2 = zero_slots(out_terms, (M + 1) + 2 * M, s, "sga_loss_anchor_multi_bwd_symx")) return rc2;
    if (A == 0 || a_hi <= a_lo || j_hi <= j_lo) return SGA_OK;
    AnchorMultiArgs a{};
    int rc = fill_anchor_multi(a, Z, M, beta, A, sums, alpha, tau_icl, tau_ial, a_lo, a_hi);
    if (rc) return rc;
    a.coef = coef; a.gs = gs; a.gamma = gamma; a.out = out_terms; a.j_lo = j_lo; a.j_hi = j_hi; a.mir = mir;
    float* inv = reinterpret_cast<float*>(gs + (size_t)(1 + SGA_SLOTS) * (M + 1) * 8);
    hipLaunchKernelGGL(inv_sums_kernel, dim3(1), dim3(64), 0, s, sums, inv, (M + 1) * 8);
    a.inv = inv;
    for (int m = 0; m < M; ++m) {
        SGA_CHECK_ARG(M1[m] && (M2[m] || mir >= j_hi), "sga_loss_anchor_multi_bwd_symx: null stash");
        a.M1[m] = M1[m]; a.M2[m] = M2[m];
    }
    const int RB = M <= 3 ? 32 : 16, TW = M <= 3 ? 2 : 4;
    const size_t lds = (size_t)(M * 2 * RB * 104 + (M + 1) * 8) * sizeof(float);
    const int nib = (a_hi - a_lo + RB - 1) / RB, ntile16 = (j_hi - j_lo + 15) / 16;
    int nsp = (6 * sga_num_cus() + nib - 1) / nib;
    if (nsp > (ntile16 + TW - 1) / TW) nsp = (ntile16 + TW - 1) / TW;
    if (nsp < 1) nsp = 1;
    a.nsplit = nsp;
    auto go = [&](auto k) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(nib * nsp), dim3(CT_THREADS), lds, s, a);
    };
    if (M == 2) go(anchor_multi_bwd16_kernel<2, true, 32, true>);
    else if (M == 3) go(anchor_multi_bwd16_kernel<3, true, 32, true>);
    else go(anchor_multi_bwd16_kernel<4, true, 16, true>);
    fold_slots(out_terms, (M + 1) + 2 * M, s);
    fold_slots(gs, (M + 1) * 8, s);
    fold_slots(gamma, M, s);
    SGA_CHECK_LAUNCH("sga_loss_anchor_multi_bwd_symx");
    return SGA_OK;
}

extern "C" int sga_loss_anchor_multi_bwd_sym(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                                             float tau_icl, float tau_ial, const float* coef, float* const* M1, float* const* M2,
                                             double* gs, double* gamma, int a_lo, int a_hi, double* out_terms, void* stream) {
    return sga_loss_anchor_multi_bwd_symx(Z, M, beta, A, sums, alpha, tau_icl, tau_ial, coef, M1, M2, gs, gamma, a_lo, a_hi, a_lo, A, a_hi, out_terms, stream);
}

/* The four products of a symmetric block's two stashes for one table (Z = [X1 | X2 | ...] rows of width Dp; R = [a_lo, a_hi), C = [j_lo, j_hi),
 * C' = [mir, j_hi)):    dX1[R] += M1^T X2[C]     dX2[C] += M1 X1[R]     dX1[C'] += M2 X2[R]     dX2[R] += M2^T X1[C'] */
extern "C" int sga_loss_stash_grad_symx(const float* M1, const float* M2, const float* Z, int A, int Dp, float* dZ, int a_lo, int a_hi,
                                        int j_lo, int j_hi, int mir, void* stream) {
    SGA_CHECK_ARG(M1 && Z && dZ && A >= 0 && Dp >= 8 && Dp % 8 == 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi && j_lo >= 0 && j_lo <= j_hi && j_hi <= A &&
                  mir >= j_lo && (M2 || mir >= j_hi), "sga_loss_stash_grad_symx: bad argument");
    const int ns = a_hi - a_lo, c1 = j_hi - j_lo, c2 = mir < j_hi ? j_hi - mir : 0;
    if (A == 0 || ns == 0 || c1 == 0) return SGA_OK;
    // (a last block whose row count is not a multiple of 4 takes sga_gemm's general kernel: correct, slower)
    const float* X1 = Z;
    const float* X2 = Z + (size_t)A * Dp;
    float* d1 = dZ;
    float* d2 = dZ + (size_t)A * Dp;
    int rc = sga_gemm(1, 0, ns, Dp, c1, M1, ns, 0, X2 + (size_t)j_lo * Dp, Dp, d1 + (size_t)a_lo * Dp, Dp, nullptr, 1, stream);
    if (!rc) rc = sga_gemm(0, 0, c1, Dp, ns, M1, ns, 0, X1 + (size_t)a_lo * Dp, Dp, d2 + (size_t)j_lo * Dp, Dp, nullptr, 1, stream);
    if (!rc && c2 > 0) rc = sga_gemm(0, 0, c2, Dp, ns, M2, ns, 0, X2 + (size_t)a_lo * Dp, Dp, d1 + (size_t)mir * Dp, Dp, nullptr, 1, stream);
    if (!rc && c2 > 0) rc = sga_gemm(1, 0, ns, Dp, c2, M2, ns, 0, X1 + (size_t)mir * Dp, Dp, d2 + (size_t)a_lo * Dp, Dp, nullptr, 1, stream);
    return rc;
}

extern "C" int sga_loss_stash_grad_sym(const float* M1, const float* M2, const float* Z, int A, int Dp, float* dZ, int a_lo, int a_hi,
                                       void* stream) {
    return sga_loss_stash_grad_symx(M1, M2, Z, A, Dp, dZ, a_lo, a_hi, a_lo, A, a_hi, stream);
}
