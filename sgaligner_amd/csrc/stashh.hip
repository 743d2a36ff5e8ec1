// 'f16x2' form of the stash products of the anchors x anchors backward (sga_loss_stash_grad_symx in contrastive.hip is the exact-fp32
// form: four GEMMs on the xf32 MFMA, 0.36 s of a configs[2] step at ~85 TFLOP/s).  Reference: autograd of src/aligner/losses.py:6,50-57,81-94
// -- dL/dX = dL/dS . X for the similarity matrices S = X1 X2^T of every table.
//
// The stash T [rows j][ns columns i] holds dL/dS[i, j] (fp32, written by anchor_multi_bwd16_kernel); the products are
//     NN:  out[j, :] += sum_i T[j, i] X[i, :]        TN:  out[i, :] += sum_j T[j, i] X[j, :]
// with X rows of a table's unit-norm [X1 | X2] block (104 columns, 100..103 zero).  Both operands enter v_mfma_f32_16x16x32_f16 as
// fp16 hi + lo of SCALED values (22 significand bits; products hi.hi + hi.lo + lo.hi, fp32 accumulate -- the split of the loss sweeps,
// csrc/sweeph.hip):
//   * X: 4096 x, split once per step and table into TRANSPOSED planes Xt[side][hi|lo][112][ldp] (stash_planes_kernel), so that a lane's
//     8 contraction slots are 16 contiguous bytes for either product;
//   * T: scaled by a power of two that puts the launch's largest |coefficient| (cmax: found by the caller) into [2^13, 2^14), split
//     in registers on the way from global memory.  An element below 2^-17 of the maximum keeps >= 12 bits, below 2^-38 it vanishes
//     (relative to the maximum: 2^-38).
// CENTRING.  A row of T sums to nearly zero against nearly identical X rows (the 'rel' table: meta-embedding of a few distinct relation
// sets), and the 2^-23 representation error of each scaled T[j, i] is then as large as the result (77 % of the parameter gradient's maximum
// at configs[2] without it; the fp32 GEMM has no such error: its products are exact).  So the planes hold x' = x - xbar (xbar = the side's
// column mean, kept behind the planes), the MFMAs produce T x' -- small operands, small errors -- and the missing (sum_k T[m, k]) xbar is
// added from EXACT fp32 row sums of the stash slots, accumulated on the VALU as they pass through the registers.
// No LDS: the stash is read once, straight into the A-operand layout (NN: two 16-byte loads per lane and 16-row tile along a row; TN:
// eight 4-byte loads down a column, 64-byte segments that the neighbouring row tiles of the same wave complete); the planes come from
// L2.  A wave owns 64 output rows x all 7 column tiles (112 accumulators), a workgroup 256 rows; the contraction is split over blockIdx.y
// with fp32 atomics (the outputs are accumulated into anyway).  The kernel is bound by the HBM stream of the stash (4 B per 3 x 2 x 104
// executed fp16 FLOPs).
#include <stdlib.h>
#include "sga_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_RT = 4;                       // 16-row tiles per wave
constexpr int SG_WAVES = 4;
constexpr int SG_THREADS = SG_WAVES * 64;
constexpr int SG_ROWS = SG_WAVES * SG_RT * 16;  // output rows per workgroup
constexpr int SG_NCT = 7;                      // 7 x 16 = 112 >= 104 columns
constexpr float SG_XS = 4096.f;                // scale of the unit rows: lo stays a normal fp16 number down to |x| ~ 3e-5

__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// 8 scaled floats -> packed fp16 hi / lo (element 2p in the low half of dword p); lo = fp16(v - hi) by v_fma_mix
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = v[2 * p], b = v[2 * p + 1];
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, f16x2v));
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hp), "v"(a));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hp), "v"(b));
        hi[p] = hp;
        lo[p] = l;
    }
}

// planes[side][hi|lo][112][ldp] <- 4096 x rows of Z = [X1 (A rows) | X2 (A rows)], width Dp = 104; the caller zeroed the planes
// column sums of each side's A rows -> zsum[side][104] (fp32 atomics: any vector near the mean does; the SAME vector is used on both ends)
__global__ __launch_bounds__(256) void stash_colsum_kernel(const float* __restrict__ Z, int A, int Dp, float* __restrict__ zsum) {
    const int side = blockIdx.y, d = threadIdx.x & 127, part = threadIdx.x >> 7;
    if (d >= Dp) return;
    float sum = 0.f;
    for (int r = blockIdx.x * 2 + part; r < A; r += gridDim.x * 2) sum += Z[((size_t)side * A + r) * Dp + d];
    atomicAdd(zsum + side * 104 + d, sum);
}
__global__ void stash_mean_kernel(float* __restrict__ zsum, int A) {
    const int i = threadIdx.x;
    if (i < 2 * 104) zsum[i] = zsum[i] / (float)A;
}
__global__ __launch_bounds__(256) void stash_planes_kernel(const float* __restrict__ Z, int A, int Dp, _Float16* __restrict__ planes, long ldp,
                                                           const float* __restrict__ zbar) {
    __shared__ float tile[64][105];
    const int side = blockIdx.y, r0 = blockIdx.x * 64;
    const float* src = Z + ((size_t)side * A + r0) * Dp;
    for (int e = threadIdx.x; e < 64 * 104; e += 256) {
        const int r = e / 104, d = e - r * 104;
        tile[r][d] = (r0 + r < A && d < Dp) ? (src[(size_t)r * Dp + d] - zbar[side * 104 + d]) * SG_XS : 0.f;
    }
    __syncthreads();
    _Float16* hi = planes + (size_t)(side * 2 + 0) * 112 * ldp;
    _Float16* lo = planes + (size_t)(side * 2 + 1) * 112 * ldp;
    for (int e = threadIdx.x; e < 64 * 104; e += 256) {
        const int d = e >> 6, r = e & 63;
        if (r0 + r < A) {
            const float v = tile[r][d];
            const _Float16 h = (_Float16)v;
            hi[(size_t)d * ldp + r0 + r] = h;
            lo[(size_t)d * ldp + r0 + r] = (_Float16)(v - (float)h);
        }
    }
}

struct StashHArgs {
    const float* T; long ldt;                 // NN: T[m * ldt + k]; TN: T[k * ldt + m]
    const _Float16* xhi; const _Float16* xlo; // planes of the X side, [112][ldp]; contraction index k <-> plane column xoff + k
    long ldp; int xoff;
    float* out;                               // out[m * 104 + d] += ...
    int MR, K, kper;
    const unsigned* cmax;                     // float bits of the launch's largest |T|
    const float* xbar;                        // [104] the X side's column mean (what the planes were centred by)
};

template <bool TRANS>
__global__ __launch_bounds__(SG_THREADS, 2) void stashh_kernel(StashHArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.x * SG_ROWS + wave * (SG_RT * 16);
    const int kbeg = blockIdx.y * a.kper, kend = min(a.K, kbeg + a.kper);
    if (m0 >= a.MR || kbeg >= kend) return;
    // power-of-two scale: largest |T| -> [2^13, 2^14); inf / NaN in the stash poison the output
    const float cm = __builtin_bit_cast(float, *a.cmax);
    int ex = 0;
    (void)frexpf(cm, &ex);
    const float sc = cm > 0.f ? ldexpf(1.f, 14 - ex) : 1.f;
    const float inv = (cm == cm && cm < INFINITY) ? 1.f / (sc * SG_XS) : __builtin_nanf("");

    f32x4 acc[SG_RT][SG_NCT];
#pragma unroll
    for (int rt = 0; rt < SG_RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < SG_NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A-operand slots of this lane.  NN: tile rt's row l15 is output row m0 + 16 rt + l15 (a stash row: its 8 contraction slots are 32
    // contiguous bytes).  TN: the contraction runs DOWN the stash rows, so the four tiles interleave -- tile rt's row l15 is output row
    // m0 + 4 l15 + rt -- and one 16-byte load along a stash row feeds slot e of all four tiles (256-byte runs per 16 lanes instead of
    // four 4-byte loads in 64-byte runs).
    const float* tp[SG_RT];
#pragma unroll
    for (int rt = 0; rt < SG_RT; ++rt) {
        const int m = min(m0 + rt * 16 + l15, a.MR - 1);
        tp[rt] = a.T + (size_t)m * a.ldt;
    }
    const float* tq = a.T + min(m0 + 4 * l15, a.MR - 4);          // TN (MR % 4 == 0)
    const _Float16* bh = a.xhi + (size_t)l15 * a.ldp + a.xoff + 8 * g;
    const _Float16* bl = a.xlo + (size_t)l15 * a.ldp + a.xoff + 8 * g;

    float raw[SG_RT][8], rs[SG_RT];
#pragma unroll
    for (int rt = 0; rt < SG_RT; ++rt) rs[rt] = 0.f;
    auto load_a = [&](int k0) {
#pragma unroll
        for (int rt = 0; rt < SG_RT; ++rt) {
            if (TRANS) {
                if (rt == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = k0 + 8 * g + e;
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        const f32x4 v = k < kend ? *reinterpret_cast<const f32x4*>(tq + (size_t)k * a.ldt) : z;
#pragma unroll
                        for (int t = 0; t < SG_RT; ++t) raw[t][e] = v[t];
                    }
                }
            } else {
                const int k = k0 + 8 * g;                 // K % 8 == 0 (checked by the host): a slot group is inside or outside as a whole
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 v0 = k < kend ? *reinterpret_cast<const f32x4*>(tp[rt] + k) : z;
                const f32x4 v1 = k < kend ? *reinterpret_cast<const f32x4*>(tp[rt] + k + 4) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) { raw[rt][e] = v0[e]; raw[rt][4 + e] = v1[e]; }
            }
        }
    };
    load_a(kbeg);
#pragma unroll 1
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        u32x4 ah[SG_RT], al[SG_RT];
#pragma unroll
        for (int rt = 0; rt < SG_RT; ++rt) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = raw[rt][e] * sc;
            rs[rt] += ((raw[rt][0] + raw[rt][1]) + (raw[rt][2] + raw[rt][3])) + ((raw[rt][4] + raw[rt][5]) + (raw[rt][6] + raw[rt][7]));   // exact-fp32 row sum
            split8(v, ah[rt], al[rt]);
        }
        if (k0 + 32 < kend) load_a(k0 + 32);            // the next step's stash slots travel under this step's MFMAs
        // (holding the NEXT step's plane operands in registers as well -- requested tile by tile behind the MFMAs that free them -- spills and
        // is no faster: tools/bench_aa.py)
#pragma unroll
        for (int ct = 0; ct < SG_NCT; ++ct) {
            const u32x4 xh = *reinterpret_cast<const u32x4*>(bh + (size_t)ct * 16 * a.ldp + k0);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(bl + (size_t)ct * 16 * a.ldp + k0);
#pragma unroll
            for (int rt = 0; rt < SG_RT; ++rt) {
                acc[rt][ct] = mfma_h(al[rt], xh, acc[rt][ct]);
                acc[rt][ct] = mfma_h(ah[rt], xl, acc[rt][ct]);
                acc[rt][ct] = mfma_h(ah[rt], xh, acc[rt][ct]);
            }
        }
    }
    // acc[rt][ct][r] = out[m0 + 16 rt + 4 g + r][16 ct + l15]  (TN: row m0 + 4 (4 g + r) + rt)
    float xb[SG_NCT];
#pragma unroll
    for (int ct = 0; ct < SG_NCT; ++ct) xb[ct] = ct * 16 + l15 < 104 ? a.xbar[ct * 16 + l15] : 0.f;
#pragma unroll
    for (int rt = 0; rt < SG_RT; ++rt) {
        rs[rt] += __shfl_xor(rs[rt], 16, 64);                       // the four k groups of tile row l15
        rs[rt] += __shfl_xor(rs[rt], 32, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = TRANS ? m0 + 4 * (4 * g + r) + rt : m0 + rt * 16 + 4 * g + r;
            const float rsum = __shfl(rs[rt], 4 * g + r, 64);       // tile row 4 g + r
            if (m >= a.MR) continue;
#pragma unroll
            for (int ct = 0; ct < SG_NCT; ++ct) {
                const int d = ct * 16 + l15;
                if (d < 104) atomicAdd(a.out + (size_t)m * 104 + d, fmaf(acc[rt][ct][r], inv, rsum * xb[ct]));
            }
        }
    }
}

int launch_stashh(bool trans, const float* T, long ldt, const _Float16* planes, long ldp, int side, int xoff, float* out, int MR, int K,
                  const unsigned* cmax, hipStream_t s) {
    if (MR <= 0 || K <= 0) return SGA_OK;
    static const int dbg = getenv("SGA_DBG_STASHH") ? atoi(getenv("SGA_DBG_STASHH")) : 0;     // timing: 1 = TN launches only, 2 = NN only
    if ((dbg == 1 && !trans) || (dbg == 2 && trans)) return SGA_OK;
    StashHArgs a{};
    a.T = T; a.ldt = ldt; a.ldp = ldp; a.xoff = xoff; a.out = out; a.MR = MR; a.K = K; a.cmax = cmax;
    a.xbar = reinterpret_cast<const float*>(planes + (size_t)4 * 112 * ldp) + side * 104;
    a.xhi = planes + (size_t)(side * 2 + 0) * 112 * ldp;
    a.xlo = planes + (size_t)(side * 2 + 1) * 112 * ldp;
    const int gx = (MR + SG_ROWS - 1) / SG_ROWS;
    // ~4 workgroups per CU; at least 256 contraction slots per split
    int splits = (4 * sga_num_cus() + gx - 1) / gx;
    const int maxs = (K + 255) / 256;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int kper = ((K + splits - 1) / splits + 31) / 32 * 32;
    splits = (K + kper - 1) / kper;
    a.kper = kper;
    if (trans) hipLaunchKernelGGL(stashh_kernel<true>, dim3(gx, splits), dim3(SG_THREADS), 0, s, a);
    else hipLaunchKernelGGL(stashh_kernel<false>, dim3(gx, splits), dim3(SG_THREADS), 0, s, a);
    return SGA_OK;
}

inline long planes_ld(int A) { return ((long)A + 63) / 64 * 64 + 64; }

}  // namespace

extern "C" size_t sga_loss_stash_planes_bytes(int A) { return A > 0 ? (size_t)4 * 112 * planes_ld(A) * sizeof(_Float16) + 2 * 104 * sizeof(float) : 0; }

extern "C" int sga_loss_stash_planes(const float* Z, int A, int Dp, void* planes, void* stream) {
    SGA_CHECK_ARG(A >= 0 && Dp == 104, "sga_loss_stash_planes: rows of width 104 (got Dp=%d)", Dp);
    if (A == 0) return SGA_OK;
    SGA_CHECK_ARG(Z && planes, "sga_loss_stash_planes: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(planes, 0, sga_loss_stash_planes_bytes(A), s) != hipSuccess) { sga_set_error("sga_loss_stash_planes: memset failed"); return SGA_ERR_HIP; }
    float* zbar = reinterpret_cast<float*>(static_cast<_Float16*>(planes) + (size_t)4 * 112 * planes_ld(A));
    const int gcs = A / 2 < 512 ? (A + 1) / 2 : 512;
    hipLaunchKernelGGL(stash_colsum_kernel, dim3(gcs > 0 ? gcs : 1, 2), dim3(256), 0, s, Z, A, Dp, zbar);
    hipLaunchKernelGGL(stash_mean_kernel, dim3(1), dim3(256), 0, s, zbar, A);
    hipLaunchKernelGGL(stash_planes_kernel, dim3((A + 63) / 64, 2), dim3(256), 0, s, Z, A, Dp, static_cast<_Float16*>(planes), planes_ld(A), zbar);
    SGA_CHECK_LAUNCH("sga_loss_stash_planes");
    return SGA_OK;
}

extern "C" int sga_loss_stash_grad_symx_f16x2(const float* M1, const float* M2, const void* planes, const uint32_t* cmax, int A, float* dZ,
                                              int a_lo, int a_hi, int j_lo, int j_hi, int mir, void* stream) {
    SGA_CHECK_ARG(M1 && planes && cmax && dZ && A >= 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi && j_lo >= 0 && j_lo <= j_hi && j_hi <= A &&
                  mir >= j_lo && (M2 || mir >= j_hi), "sga_loss_stash_grad_symx_f16x2: bad argument");
    const int ns = a_hi - a_lo, c1 = j_hi - j_lo, c2 = mir < j_hi ? j_hi - mir : 0;
    if (A == 0 || ns == 0 || c1 == 0) return SGA_OK;
    SGA_CHECK_ARG(ns % 8 == 0 && a_lo % 8 == 0 && j_lo % 8 == 0 && (c2 == 0 || mir % 8 == 0),
                  "sga_loss_stash_grad_symx_f16x2: block [%d,%d) x [%d,%d) (mirror %d) not on 8-row boundaries (the exact-fp32 sga_loss_stash_grad_symx takes it)",
                  a_lo, a_hi, j_lo, j_hi, mir);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const _Float16* pl = static_cast<const _Float16*>(planes);
    const long ldp = planes_ld(A);
    float* d1 = dZ;
    float* d2 = dZ + (size_t)A * 104;
    // dX1[R] += M1^T X2[C]     dX2[C] += M1 X1[R]     dX1[C'] += M2 X2[R]     dX2[R] += M2^T X1[C']
    launch_stashh(true, M1, ns, pl, ldp, 1, j_lo, d1 + (size_t)a_lo * 104, ns, c1, cmax, s);
    launch_stashh(false, M1, ns, pl, ldp, 0, a_lo, d2 + (size_t)j_lo * 104, c1, ns, cmax, s);
    if (c2 > 0) {
        launch_stashh(false, M2, ns, pl, ldp, 1, a_lo, d1 + (size_t)mir * 104, c2, ns, cmax, s);
        launch_stashh(true, M2, ns, pl, ldp, 0, mir, d2 + (size_t)a_lo * 104, ns, c2, cmax, s);
    }
    SGA_CHECK_LAUNCH("sga_loss_stash_grad_symx_f16x2");
    return SGA_OK;
}
