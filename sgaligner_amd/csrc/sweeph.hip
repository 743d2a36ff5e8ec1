// Opt-in, fp32-FAITHFUL split-fp16 form of the anchors x negatives loss sweeps (sga_set_mfma_mode(3), ops.set_mfma_mode('f16x2'); the default
// stays the exact-fp32 sweep16_kernel in contrastive.hip).  Same mathematics and the same two-owner-sweep structure as sweepb.hip:
//     pass 1 (sums)   s_fam,temp[table] = sum exp(S / tau)                 S = X_own . X_other^T per modality table,
//     backward (grad) dZ[own] += C . Z[other],  C = dL/dS_m + beta_m dL/dS_J    S_J = sum_m beta_m S_m (joint table derived)
// (reference src/aligner/losses.py:5-15 and its autograd).
//
// Arithmetic.  Every fp32 operand z (|z| <= 1: the rows are L2-normalised) enters the matrix cores as TWO fp16 terms of 4096 z:
// hi = fp16(4096 z), lo = fp16(4096 z - hi) -- 22 significand bits, and with the 2^12 pre-scale lo stays a NORMAL fp16 number for every
// |z| >= 6e-5 (absolute error <= 2^-37 below that), so no second accumulator and no dependence on subnormal handling.  A product is
// hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16 into one fp32 accumulator (fp16 x fp16 is exact in fp32; the dropped lo.lo term is
// 2^-24 relative).  Measured on the MI355X against an fp64 sum (tools/micro/f16x2_probe.hip, 2 M similarities of unit rows, D = 100):
// max |dS| 1.7e-7 / rms 1.5e-8 (exact-fp32 MFMA chain: 2.1e-7 / 1.8e-8) for uncorrelated rows, 4.4e-7 / 9.2e-8 (6.6e-7 / 1.3e-7) for
// rows with S ~ 0.99 -- the fp32 ACCUMULATION rounding dominates both, i.e. the similarities carry fp32's own error (the bf16 hi+lo
// split of sweepb.hip: 16 bits, 20x worse, amplified by 1/tau = 10 in the coefficients).
// The coefficient C (gradient GEMM A operand) is scaled per table by a power of two so that |C| <= 2^14 fits fp16 whatever the loss
// scale (bound from dL/d(sums), |S| <= 1) and split the same way; the accumulators are unscaled by the exact inverse at the end.
//
// On gfx950 an MFMA occupies the SIMD's vector issue port for all of its passes: VALU, LDS and LDS-DMA instructions of the same AND of the
// other resident waves add to the matrix time, they never hide under it (tools/micro/f16_valu_overlap.hip: an MFMA-only wave and an
// FMA-only wave sharing a SIMD take the SUM of their solo times, 64.9 + 37.7 -> 102.5 cycles per iteration; in one stream every v_fma
// costs 2.2-2.7 and every v_exp 4.5 cycles on top of the MFMAs).  So the kernel is built to minimise the instruction count per pair:
//   * one wave owns 32 owner rows x ALL M tables (one wave per SIMD, up to 512 registers): every LDS read of the "other" tile feeds two
//     owner halves (sweepb: 16 rows per wave, twice the reads per MFMA, owner lo planes re-read from LDS every tile);
//   * the 8-column K tail (columns 96..103) packs its three products into the 32 k slots of ONE MFMA (k group 0: hi.hi, 1: hi.lo, 2: lo.hi,
//     3: lo.lo): 10 instead of 12 MFMAs per 16 x 16 similarity tile, same accumulator chain;
//   * the 16-byte slots of a K step are stored XOR-swizzled (slot i ^ 12 for odd k groups) so that BOTH the lane-linear ds_read_b128 of
//     the S product and the ds_read_b64_tr_b16 transpose reads of the gradient GEMM are bank-conflict free (sweepb: 2-way conflicts on
//     every transpose read).
//
// Data layout.  sga_loss_split16_tables turns a packed fp32 table Z [X1 | X2 | N1 | N2] into 32-row BLOCKS (each segment padded to whole
// blocks) of 14 336 B: [hi plane 6 144 | lo plane 6 144 | packed tail 2 048], a plane = [K step q (3)][half jh (2)][64 slots][8 fp16],
// slot(g, i) = 16 g + (i ^ 12 (g & 1)) holds columns 32 q + 8 g .. + 7 of row 8 (i >> 2) + 4 jh + (i & 3); the tail = [jh][64 slots][8 fp16]
// with k groups 0, 1 = hi and 2, 3 = lo of columns 96 .. 103.  A tile of "other" rows is ONE contiguous 14 KiB copy per table (14
// global_load_lds DMA chunks), double-buffered in LDS (3 tables: 84 KiB).
// MFMA bookkeeping (v_mfma_f32_16x16x32_f16; A: lane&15 = row, B: lane&15 = column, lane>>4 = k group of 8 slots):
//   S^T tile: A = other rows from LDS, B = owner rows (registers).  Half jh of a 32-row tile uses A row i <-> other row
//   8 (i>>2) + 4 jh + (i&3), so that a lane's 8 accumulator values (2 halves x 4) are the 8 CONSECUTIVE other rows 8 g4 .. 8 g4+7:
//   exactly the k slots of the gradient MFMA  dZ[own] += C[own, other] Z[other, cols]  whose A operand is therefore the coefficient
//   registers (split into fp16 hi/lo) and whose B operand comes from transpose reads of the same planes.
#include <stdlib.h>
#include <type_traits>

#include "loss_math.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int SH_OWN = 128;                      // owner rows per workgroup (8 waves x 16 or 4 waves x 32)
constexpr int SH_DP = 104;
constexpr int SH_PLANE = 3 * 2 * 1024;           // 6144 B
constexpr int SH_TAIL = 2 * SH_PLANE;            // byte offset of the packed tail in a block
constexpr int SH_BLOCK = SH_TAIL + 2 * 1024;     // 14336 B
constexpr int SH_NCH = SH_BLOCK / 1024;          // 14 DMA chunks
constexpr float SH_PRE = 4096.f;                 // operand pre-scale 2^12: S accumulates 2^24 S
constexpr float SH_UNPRE = 1.f / (4096.f * 4096.f);

// v0, v1 -> packed fp16 hi pair and lo pair (v = hi + lo to 22 bits)
__device__ __forceinline__ void split_pair16(float v0, float v1, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(f32x2{v0, v1}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    // lo = fp16(v - hi), the difference exact in fp32: v_fma_mixlo / mixhi_f16 read the fp16 hi half directly and write the fp16 result
    // (3 instead of 6 VALU per pair: no v_cvt_f32_f16 x 2, v_sub x 2, v_cvt_pk)
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v1));
    lo = l;
}
__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 tr_read16(const unsigned char* p) {     // ds_read_b64_tr_b16
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p)));
}
__host__ __device__ constexpr int sh_slot(int g, int i) { return 16 * g + (i ^ (12 * (g & 1))); }

#ifndef SH_BDEPTH
#define SH_BDEPTH 2
#endif
struct HLayout { int nbA, nb1, nb2; };
__host__ __device__ inline HLayout make_hlayout(int A, int J1, int J2) { return HLayout{(A + 31) / 32, (J1 + 31) / 32, (J2 + 31) / 32}; }

// ---- CENTRING.  The planes hold z' = z - zbar (zbar = the table's column mean over its R packed rows) plus two bookkeeping columns:
//   column 100: b = zbar . z' + |zbar|^2 / 2        column 101: 1
// so that with the owner's two columns SWAPPED (1, b_i) the K tail of the S product adds b_j + b_i and the MFMAs still deliver
//   S_ij = z_i . z_j = z'_i . z'_j + zbar . z'_i + zbar . z'_j + |zbar|^2   exactly as before,
// and the gradient GEMM's output column 101 is rowsum_i = sum_j c_ij, from which dZ_i = sum_j c_ij z'_j + rowsum_i zbar.
// Why: a table whose rows are nearly identical (meta_embedding_rel: bag-of-words rows that are almost all alike, S ~ 1 for every pair) has a
// loss gradient that is the small TANGENTIAL remainder of a large radial sum.  The 22-bit hi + lo image of z_j carries the same rounding
// residue for every such row, so sum_j c_ij (z_j + eps_j) grows a coherent (sum_j c_ij) eps term with a tangential part of the remainder's own
// size: at 1024 pairs x 128 objects the un-centred kernel missed d(meta_embedding_rel.weight) by 1.1 x its maximum against fp64 where the
// exact-fp32 sweep misses it by 2.6 % (tests/test_fp64_chunked_gpu.py).  Centred, the residue is relative to |z'| (the spread of the rows),
// the accumulators hold the small sums only, and the radial part enters once, exactly, through rowsum x zbar.
// Statistics block of a table, behind its blocks and the slack block: float zbar[104] | float nbh (= |zbar|^2 / 2) ... | at +512 B: double colsum[104].
constexpr int SH_STAT_BYTES = 2048;
constexpr int SH_DREAL = 100;                    // data columns; 100, 101 are the bookkeeping columns (emb_dim <= 100 in this mode)

__global__ __launch_bounds__(128) void split16_colsum_kernel(const float* __restrict__ Z, int R, double* __restrict__ colsum) {
    const int c = threadIdx.x;
    const int per = (R + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    if (c >= SH_DP) return;
    double acc = 0.0;
    for (int r = r0; r < r1; ++r) acc += (double)Z[(size_t)r * SH_DP + c];
    if (r1 > r0) atomicAdd(colsum + c, acc);
}
__global__ __launch_bounds__(128) void split16_stats_kernel(const double* __restrict__ colsum, int R, float* __restrict__ stat) {
    __shared__ double sq[128];
    const int c = threadIdx.x;
    const float zb = (c < SH_DREAL && R > 0) ? (float)(colsum[c] / (double)R) : 0.f;
    if (c < SH_DP) stat[c] = zb;
    sq[c] = (double)zb * (double)zb;
    __syncthreads();
    if (c == 0) {
        double t = 0.0;
        for (int i = 0; i < 128; ++i) t += sq[i];
        stat[SH_DP] = (float)(0.5 * t);
    }
}

// fp32 packed table -> blocked fp16 hi/lo planes + packed tail of the CENTRED rows (one workgroup per 32-row block)
__global__ __launch_bounds__(256) void split16_tables_kernel(const float* __restrict__ Z, int A, int J1, int J2, unsigned char* __restrict__ Zb,
                                                             const float* __restrict__ stat) {
    __shared__ float tile[32 * SH_DP];
    __shared__ float zbar[SH_DP];
    const HLayout L = make_hlayout(A, J1, J2);
    int b = blockIdx.x, old0, len;
    if (b < L.nbA) { old0 = 0; len = A; }
    else if (b < 2 * L.nbA) { b -= L.nbA; old0 = A; len = A; }
    else if (b < 2 * L.nbA + L.nb1) { b -= 2 * L.nbA; old0 = 2 * A; len = J1; }
    else { b -= 2 * L.nbA + L.nb1; old0 = 2 * A + J1; len = J2; }
    const int nvalid = min(32, len - 32 * b);
    const float* src = Z + (size_t)(old0 + 32 * b) * SH_DP;
    if (threadIdx.x < SH_DP) zbar[threadIdx.x] = stat[threadIdx.x];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * SH_DP; e += 256) {
        const int r = e / SH_DP, c = e - r * SH_DP;
        tile[e] = (r < nvalid && c < SH_DREAL) ? src[e] - zbar[c] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 32) {                       // the two bookkeeping columns of a valid row (a padding row stays all zero: S = 0 as before)
        const int r = threadIdx.x;
        if (r < nvalid) {
            double a = 0.0;
            for (int c = 0; c < SH_DREAL; ++c) a += (double)zbar[c] * (double)tile[r * SH_DP + c];
            tile[r * SH_DP + SH_DREAL] = (float)(a + (double)stat[SH_DP]);
            tile[r * SH_DP + SH_DREAL + 1] = 1.f;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * SH_DP; e += 256) tile[e] *= SH_PRE;
    __syncthreads();
    unsigned* out = reinterpret_cast<unsigned*>(Zb + (size_t)blockIdx.x * SH_BLOCK);
    // planes: dword e = (slot s of [q][jh][64], pair p of 4): stored slot (g, i ^ swz) <- logical (g, i)
    for (int e = threadIdx.x; e < 3 * 2 * 64 * 4; e += 256) {
        const int p = e & 3, st = (e >> 2) & 63, jh = (e >> 8) & 1, q = e >> 9;
        const int g = st >> 4, i = (st & 15) ^ (12 * (g & 1));
        const int row = 8 * (i >> 2) + 4 * jh + (i & 3), col = 32 * q + 8 * g + 2 * p;
        unsigned hi, lo;
        split_pair16(tile[row * SH_DP + col], tile[row * SH_DP + col + 1], hi, lo);
        out[e] = hi; out[SH_PLANE / 4 + e] = lo;
    }
    // packed tail: dword e = (slot of [jh][64], pair p of 4): k groups 0, 1 = hi, 2, 3 = lo of columns 96 + 2 p, + 1
    for (int e = threadIdx.x; e < 2 * 64 * 4; e += 256) {
        const int p = e & 3, st = (e >> 2) & 63, jh = e >> 8;
        const int g = st >> 4, i = (st & 15) ^ (12 * (g & 1));
        const int row = 8 * (i >> 2) + 4 * jh + (i & 3), col = 96 + 2 * p;
        unsigned hi, lo;
        split_pair16(tile[row * SH_DP + col], tile[row * SH_DP + col + 1], hi, lo);
        out[SH_TAIL / 4 + e] = g < 2 ? hi : lo;
    }
}

#ifdef SH_DBG_TIMING
__device__ unsigned long long g_sh_dbg[16];
#define SH_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; }
#else
#define SH_T(i)
#endif
struct HSeg { int blk0, jt_lo, jt_hi, old0, lo, hi, fam; };   // others: block blk0 + jt holds old rows old0 + 32 jt + w; valid rows in [lo, hi)
struct HGroup { int own0, nown, own_old0, own_blk0, blk0, nsplit, nseg; HSeg seg[2]; };
struct HArgs {
    int M; const unsigned char* Zb[4]; int ngroups; HGroup grp[4];
    float k0, k1, it0, it1;
    const float* beta;
    double* sums;                    // SUM out  [(M+1)][8] (+ slots)
    const double* gs;                // GRAD in  [(M+1)][8]
    float* dZ[4];                    // GRAD out (fp32, old row order), atomic accumulate
    const float* stat[4];            // per table: zbar[104], |zbar|^2 / 2 (behind the blocks of Zb)
    double* gamma;                   // GRAD out [M] (+ slots)
};

// CLO: the coefficients' lo terms (C = hi + lo, third gradient MFMA).  Without them C is rounded to fp16's 11 bits: an independent,
// unbiased rounding error of <= 2^-12 per (owner, other) pair.
// OH: owner halves (of 16 rows) per wave.  OH = 1: 8 waves x 16 rows, two waves per SIMD at <= 256 registers -- the CU's issue logic hands a
// SIMD one instruction per TYPE per 4-cycle slot from DIFFERENT waves, so one wave's LDS / scalar / wait / DMA instructions (a third of the
// instruction stream) issue in the shadow of the other's MFMAs and a parked wave costs nothing; OH = 2: 4 waves x 32 rows, one wave per SIMD
// at <= 512 registers -- half the LDS reads per MFMA, but every instruction of the stream then takes its own 4-cycle issue slot (SQ
// counters, profiles/r04_f_sweeph_sq.txt: 246 MFMAs = 3 936 matrix cycles + ~990 other instructions = 3 900 issue cycles per tile).
// WV: waves per workgroup.  8 / OH (128 owner rows) for M <= 3; M = 4 runs OH = 1 with FOUR waves (64 owner rows, one wave per SIMD at up to
// 512 registers): four tables of owner operands (112) and gradient accumulators (112) do not fit the 256 registers of a two-wave SIMD.
template <int M, bool GRAD, bool CLO, int OH, int WV = 8 / OH>
__global__ __launch_bounds__(WV * 64, (WV * OH == 8) ? 2 / OH : 1) void sweeph_kernel(HArgs a) {
    constexpr int NCT = 7;
    constexpr int WAVES = WV, THREADS = WAVES * 64;
    constexpr int OWN = WV * 16 * OH;                                 // owner rows per workgroup
    constexpr int KMAX = (SH_NCH + WAVES - 1) / WAVES;               // DMA chunk slots per table and wave
    constexpr int BUF = M * SH_BLOCK;
    constexpr bool PIPE = OH == 2;                                    // one wave per SIMD: operand prefetch pinned inside the MFMA stream
    // Forward sums (GRAD = false) with CLO = false: the S product's lo terms are dropped for the 96 main columns (4 instead of 10 MFMAs per
    // 16 x 16 tile; the K tail -- columns 96..103 and the centring's bookkeeping columns -- keeps all four products).  A sum of >= 2^24
    // exponentials only needs every S to ~1e-5 with an UNBIASED error: the 11-bit rounding of the centred rows is random per element, the
    // sum's relative error is that of one term (3e-5) over the root of the term count, and the second-order bias 50 dS^2 ~ 6e-10.
    constexpr bool SLO = GRAD || CLO;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsh[];      // [2][M][SH_BLOCK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const HGroup& grp = a.grp[g];
    // XCD-aware work order (as sweepb / sweep16): the group's (split major, owner block minor) work list in 8 contiguous per-XCD chunks
    const int wg_in_grp = (int)blockIdx.x - grp.blk0;
    const int nsplit = grp.nsplit, n_ob = (grp.nown + OWN - 1) / OWN, n_units = n_ob * nsplit;
    const int unit = (wg_in_grp & 7) * ((n_units + 7) >> 3) + (wg_in_grp >> 3);
    if ((wg_in_grp >> 3) >= ((n_units + 7) >> 3) || unit >= n_units) return;
    const int split = unit / n_ob;
    const int own0 = grp.own0 + (unit - split * n_ob) * OWN;
    const int own_end = grp.own0 + grp.nown;
    const int wrow0 = own0 + wave * 16 * OH;                          // this wave's first owner row
#ifdef SH_DBG_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif

    // ---- owner rows (OH halves of 16) as the S product's B operand: 3 K = 32 steps hi / lo + the packed tail (k group 0, 2: hi; 1, 3: lo)
    u32x4 ohi[M][OH][3], olo[M][OH][3], otl[M][OH];
    float beta[M];
#pragma unroll
    for (int oh = 0; oh < OH; ++oh) {
        const int my_i = wrow0 + oh * 16 + l15;
        const bool iv = my_i < own_end;
        const int rel = (iv ? my_i : own0) - grp.own_old0;
        const int o = rel & 31, oi = 4 * (o >> 3) + (o & 3), ojh = (o >> 2) & 1;       // row o sits at (half ojh, operand row oi) of its block
        const size_t off = (size_t)(grp.own_blk0 + (rel >> 5)) * SH_BLOCK;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned char* base = a.Zb[m] + off;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int so = ((q * 2 + ojh) * 64 + sh_slot(g4, oi)) * 16;
                ohi[m][oh][q] = iv ? *reinterpret_cast<const u32x4*>(base + so) : u32x4{0, 0, 0, 0};
                olo[m][oh][q] = iv ? *reinterpret_cast<const u32x4*>(base + SH_PLANE + so) : u32x4{0, 0, 0, 0};
            }
            // tail B operand: k group 0 -> own hi, 1 -> own lo, 2 -> own hi, 3 -> own lo (against A = other hi, hi, lo, lo)
            otl[m][oh] = iv ? *reinterpret_cast<const u32x4*>(base + SH_TAIL + (ojh * 64 + sh_slot((g4 & 1) * 2, oi)) * 16) : u32x4{0, 0, 0, 0};
            otl[m][oh][2] = (otl[m][oh][2] >> 16) | (otl[m][oh][2] << 16);      // columns 100, 101: the owner holds (1, b_i) against the other's (b_j, 1)
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) beta[m] = a.beta[m];

    f32x4 gacc[GRAD ? M : 1][OH][NCT];
#pragma unroll
    for (int m = 0; m < (GRAD ? M : 1); ++m)
#pragma unroll
        for (int oh = 0; oh < OH; ++oh)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) gacc[m][oh][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gam[M];
#pragma unroll
    for (int m = 0; m < M; ++m) gam[m] = 0.f;

    // per-table power-of-two scale of the coefficients: |c_m| <= cmax_m (|S| <= 1, any family) -> |c_m sig_m| <= 2^14
    float sig[M], isig[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { sig[m] = 1.f; isig[m] = 1.f; }
    if (GRAD) {
        const float e0 = __builtin_amdgcn_exp2f(a.k0 * 1.0000005f), e1 = __builtin_amdgcn_exp2f(a.k1 * 1.0000005f);   // e^{1/tau}, a hair above
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float cmax = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float t0 = (fabsf((float)a.gs[m * 8 + f * 2]) + beta[m] * fabsf((float)a.gs[M * 8 + f * 2])) * a.it0 * e0;
                const float t1 = (fabsf((float)a.gs[m * 8 + f * 2 + 1]) + beta[m] * fabsf((float)a.gs[M * 8 + f * 2 + 1])) * a.it1 * e1;
                cmax = fmaxf(cmax, t0 + t1);
            }
            if (cmax > 0.f && cmax < 3.0e38f) {
                int ex;
                (void)frexpf(cmax, &ex);                      // cmax = f 2^ex, f in [0.5, 1)
                ex = min(max(ex, -100), 100);
                sig[m] = ldexpf(1.f, 14 - ex);
                isig[m] = ldexpf(1.f, ex - 14 - 12);          // ... and the 2^12 of the B operand
            } else {
                isig[m] = 1.f / SH_PRE;
            }
        }
    }

    // Tile transport: ONE contiguous 14-KiB copy per table by LDS-DMA (global_load_lds, 1 KiB per wave instruction), slot (table m, k) ->
    // chunk (wave + m) % WAVES + WAVES k: the table index of every DMA is a compile-time constant (no kernel-argument reload + s_waitcnt in
    // front of it).  Issued in one burst at the top of a tile for the NEXT tile.  (Tried and dropped, profiles/r04_e_sweeph_transport.txt:
    // the chunks through registers -- buffer_load_dwordx4 + ds_write_b128, 1-3 in flight, spread between the MFMA groups: 7.0-7.8 ms vs 6.9;
    // the DMAs themselves spread through the tile: hipcc orders every ds_read_b64_tr_b16 behind ALL outstanding LDS-DMA with vmcnt(0).)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // M0 (the DMA's LDS address) must be provably uniform
    auto issue = [&](int blk, unsigned char* buf) {
#ifdef SH_DBG_NODMA
        return;
#endif
        int l16 = threadIdx.x;
        asm volatile("" : "+v"(l16));
        l16 = (l16 & 63) * 16;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int rot = (wave_u + m) & (WAVES - 1);
            const unsigned char* src = a.Zb[m] + ((size_t)blk * SH_BLOCK + rot * 1024) + l16;
            unsigned char* dst = buf + m * SH_BLOCK + rot * 1024;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if ((k + 1) * WAVES > SH_NCH && rot + k * WAVES >= SH_NCH) break;       // uniform; only the last k can fall off the block
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * WAVES * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + k * WAVES * 1024), 16, 0, 0);
            }
        }
    };

    // lane-derived LDS offsets: S product (lane-linear up to the swizzle) and the transpose reads of the gradient GEMM's B operand:
    // lane i of a 16-lane group addresses the 8-byte piece (row 8 g4 + 4 rd + (i >> 2), columns 16 ct + 4 (i & 3) ..)
    const int aoff = sh_slot(g4, l15) * 16;
    const int tr_io = 4 * g4 + (l15 >> 2), tr_cs = l15 & 3;
    const int tr_main = sh_slot(tr_cs >> 1, tr_io) * 16 + (tr_cs & 1) * 8;       // + (ct >> 1) * 2048 + rd * 1024 + (ct & 1) * 512
    const int tr_tail = SH_TAIL + tr_io * 16 + (tr_cs & 1) * 8;                   // hi: k group 0; lo: + 512 (k group 2); + rd * 1024

#pragma unroll 1
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const HSeg seg = grp.seg[sg];
        // coefficient constants of this segment's family, pre-multiplied by the table's scale; the joint coefficient stays unscaled (Gamma)
        float c0[M + 1], c1[M + 1], bsig[M];
#pragma unroll
        for (int m = 0; m <= M; ++m) {
            const float s_ = m < M ? sig[m < M ? m : 0] : 1.f;
            c0[m] = GRAD ? (float)(a.gs[m * 8 + seg.fam * 2 + 0] * (double)a.it0) * s_ : 0.f;
            c1[m] = GRAD ? (float)(a.gs[m * 8 + seg.fam * 2 + 1] * (double)a.it1) * s_ : 0.f;
        }
#pragma unroll
        for (int m = 0; m < M; ++m) bsig[m] = beta[m] * sig[m];
        const float k0s = a.k0 * SH_UNPRE, k1s = a.k1 * SH_UNPRE;         // exp2 arguments straight from the 2^24-scaled accumulators
        double dsum[M + 1][2];
#pragma unroll
        for (int m = 0; m <= M; ++m) { dsum[m][0] = 0.0; dsum[m][1] = 0.0; }

        __syncthreads();
        if (seg.jt_lo + split < seg.jt_hi) issue(seg.blk0 + seg.jt_lo + split, ldsh);
        int it = 0;
#pragma unroll 1
        for (int jt = seg.jt_lo + split; jt < seg.jt_hi; jt += nsplit, ++it) {
            unsigned char* buf = ldsh + (it & 1) * BUF;
            const int j0 = seg.old0 + 32 * jt;                 // old row of the tile's first row
            SH_T(5)
            __syncthreads();                                   // tile `it` has landed (every wave waited for its own chunks), buffer it + 1 is free
            SH_T(0)
            if (jt + nsplit < seg.jt_hi) issue(seg.blk0 + jt + nsplit, ldsh + ((it + 1) & 1) * BUF);
            SH_T(1)
            if (GRAD) {
                // A segment's first / last tile may hold rows outside [lo, hi) (uniform test).  Zeroing those rows' 16-byte slots in the
                // operand-order image (both planes + tail, all tables) makes their contributions vanish by themselves -- S = 0, c * 0 into the
                // owner gradient, 0 into Gamma -- so the gradient epilogue carries no validity mask.
                const int vlo = max(seg.lo - j0, 0), vhi = min(seg.hi - j0, 32);
                if (vlo > 0 || vhi < 32) {
                    for (int x = tid; x < M * 32 * 28; x += THREADS) {
                        const int pc = x % 28, w = (x / 28) % 32, m = x / (28 * 32);
                        if (w >= vlo && w < vhi) continue;
                        const int wjh = (w >> 2) & 1, wi = 4 * (w >> 3) + (w & 3);
                        unsigned char* base = buf + m * SH_BLOCK;
                        if (pc < 24) {
                            const int pl = pc / 12, q = (pc % 12) >> 2, gq = pc & 3;
                            *reinterpret_cast<u32x4*>(base + pl * SH_PLANE + ((q * 2 + wjh) * 64 + sh_slot(gq, wi)) * 16) = u32x4{0, 0, 0, 0};
                        } else {
                            *reinterpret_cast<u32x4*>(base + SH_TAIL + (wjh * 64 + sh_slot(pc - 24, wi)) * 16) = u32x4{0, 0, 0, 0};
                        }
                    }
                    __syncthreads();
                }
            }

            // ---- S^T tiles: sacc[m][oh][jh][r] = 2^24 S_m[own = 16 oh + lane&15, other = 8 g4 + 4 jh + r], one SUB-STEP = (table, other half):
            // 7 A operands from LDS, OH accumulator chains (a dependent v_mfma_f32_16x16x32_f16 issues back to back at 16 cycles,
            // tools/micro/mfma_dep_chain.hip).  PIPE (one wave per SIMD: nobody else covers an LDS read's latency): the operands are prefetched
            // on a ROLLING schedule inside the MFMA stream -- the next sub-step's tail + lo operands are requested once this one's tail /
            // lo.hi products (their last readers) have issued, its hi operands after the hi.lo / hi.hi products.
            f32x4 sacc[M][OH][2];
            u32x4 at, al[3], ah[3];
            auto ld_tl = [&](int ss) {
                const unsigned char* ar = buf + (ss >> 1) * SH_BLOCK + (ss & 1) * 1024 + aoff;
                at = *reinterpret_cast<const u32x4*>(ar + SH_TAIL);
                if (SLO) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) al[q] = *reinterpret_cast<const u32x4*>(ar + SH_PLANE + q * 2048);
                }
            };
            auto ld_h = [&](int ss) {
                const unsigned char* ar = buf + (ss >> 1) * SH_BLOCK + (ss & 1) * 1024 + aoff;
#pragma unroll
                for (int q = 0; q < 3; ++q) ah[q] = *reinterpret_cast<const u32x4*>(ar + q * 2048);
            };
            if (PIPE) {
                ld_tl(0);
                ld_h(0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ss = 0; ss < 2 * M; ++ss) {
                const int m = ss >> 1, jh = ss & 1;
                if (!PIPE) { ld_tl(ss); ld_h(ss); }
                f32x4 acc[OH];
#pragma unroll
                for (int oh = 0; oh < OH; ++oh) acc[oh] = mfma_h(at, otl[m][oh], f32x4{0.f, 0.f, 0.f, 0.f});
                if (SLO) {
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int oh = 0; oh < OH; ++oh) acc[oh] = mfma_h(al[q], ohi[m][oh][q], acc[oh]);
                }
                if (PIPE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ss + 1 < 2 * M) ld_tl(ss + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (SLO) {
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int oh = 0; oh < OH; ++oh) acc[oh] = mfma_h(ah[q], olo[m][oh][q], acc[oh]);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)                    // the large hi.hi terms last
#pragma unroll
                    for (int oh = 0; oh < OH; ++oh) acc[oh] = mfma_h(ah[q], ohi[m][oh][q], acc[oh]);
                if (PIPE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ss + 1 < 2 * M) ld_h(ss + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int oh = 0; oh < OH; ++oh) sacc[m][oh][jh] = acc[oh];
            }

            SH_T(2)
            if (!GRAD) {
                float p0[M + 1], p1[M + 1];
#pragma unroll
                for (int m = 0; m <= M; ++m) { p0[m] = 0.f; p1[m] = 0.f; }
                // forward sums: exp2(0) = 1 of a padded / foreign row would count, so edge tiles are masked; interior tiles add unmasked
                auto sums_tile = [&](auto masked_c) {
                    constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                    for (int oh = 0; oh < OH; ++oh) {
                        const bool iv = wrow0 + oh * 16 + l15 < own_end;
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = j0 + 8 * g4 + 4 * jh + r;
                                const float okf = (!MASKED || (iv && row >= seg.lo && row < seg.hi)) ? 1.f : 0.f;
                                float sj = 0.f;
#pragma unroll
                                for (int m = 0; m < M; ++m) {
                                    const float sv = sacc[m][oh][jh][r];
                                    sj = fmaf(beta[m], sv, sj);
                                    p0[m] = MASKED ? fmaf(okf, fexp2(sv * k0s), p0[m]) : p0[m] + fexp2(sv * k0s);
                                    p1[m] = MASKED ? fmaf(okf, fexp2(sv * k1s), p1[m]) : p1[m] + fexp2(sv * k1s);
                                }
                                p0[M] = MASKED ? fmaf(okf, fexp2(sj * k0s), p0[M]) : p0[M] + fexp2(sj * k0s);
                                p1[M] = MASKED ? fmaf(okf, fexp2(sj * k1s), p1[M]) : p1[M] + fexp2(sj * k1s);
                            }
                    }
                };
                if (j0 >= seg.lo && j0 + 32 <= seg.hi && own0 + OWN <= own_end) sums_tile(std::false_type{}); else sums_tile(std::true_type{});   // uniform
#pragma unroll
                for (int m = 0; m <= M; ++m) { dsum[m][0] += (double)p0[m]; dsum[m][1] += (double)p1[m]; }
            } else {
                // Gradient GEMM B operands (8 consecutive other rows 8 g4 .. + 7 of column 16 ct + c) by LDS transpose reads of the row planes,
                // one STEP = (table, column tile), OH accumulator chains.  PIPE: the operands of step k + 1 are requested before the MFMAs of
                // step k issue (double-buffered: the first step's before the joint epilogue, a table's first step's before the previous
                // table's last MFMAs), so their latency hides under matrix / VALU work of this wave itself.
                constexpr int BD = PIPE ? SH_BDEPTH : 1;          // B operand ring: steps k .. k + BD - 1 requested
                u32x4 bh[BD], bl[BD];
                auto ld_b = [&](int k) {
                    const int m = k / NCT, ct = k % NCT, par = k % BD;
                    const unsigned char* ph = ct < 6 ? buf + m * SH_BLOCK + tr_main + (ct >> 1) * 2048 + (ct & 1) * 512 : buf + m * SH_BLOCK + tr_tail;
                    const unsigned char* pl = ct < 6 ? ph + SH_PLANE : ph + 512;
                    const u32x2 h0 = tr_read16(ph), h1 = tr_read16(ph + 1024);
                    const u32x2 l0 = tr_read16(pl), l1 = tr_read16(pl + 1024);
                    bh[par] = u32x4{h0[0], h0[1], h1[0], h1[1]};
                    bl[par] = u32x4{l0[0], l0[1], l1[0], l1[1]};
                };
                if (PIPE) {
#pragma unroll
                    for (int k = 0; k + 1 < BD; ++k) ld_b(k);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // joint coefficient dL/dS_J (unscaled) for this lane's OH x 8 pairs.  Plain VALU on purpose: beside MFMAs a v_pk_*_f32 costs 7.8
                // cycles against 2.6 for v_fma / v_mul and 6.4 for v_exp (tools/micro/valu_issue.hip), so neither packed pairs nor
                // e^{S/tau0} = (e^{S/tau1})^10 by four multiplies pay (both built and measured: profiles/r04_k_sweeph_final_variants.txt)
                float cj[OH][2][4];
#pragma unroll
                for (int oh = 0; oh < OH; ++oh)
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float sj = 0.f;
#pragma unroll
                            for (int m = 0; m < M; ++m) sj = fmaf(beta[m], sacc[m][oh][jh][r], sj);
#ifdef SH_DBG_NOEPI
                            cj[oh][jh][r] = sj;
#else
                            cj[oh][jh][r] = c0[M] * fexp2(sj * k0s) + c1[M] * fexp2(sj * k1s);
#endif
                        }
                if (g < 2) {                                   // Gamma_m = sum dL/dS_J * S_m, each pair once (anchor-owner sweep); 2^24 folded out at the end
#pragma unroll
                    for (int m = 0; m < M; ++m) {
#pragma unroll
                        for (int oh = 0; oh < OH; ++oh)
#pragma unroll
                            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                                for (int r = 0; r < 4; ++r) gam[m] = fmaf(cj[oh][jh][r], sacc[m][oh][jh][r], gam[m]);
                        asm volatile("" : "+v"(gam[m]));
                    }
                }
                SH_T(3)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    // sig_m c_m for this lane's 8 consecutive other rows (k slot j = 4 jh + r), split into fp16 hi / lo: the A operand
                    u32x4 chi[OH], clo[OH];
#pragma unroll
                    for (int oh = 0; oh < OH; ++oh)
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int jh = p >> 1, r = (p & 1) * 2;
                            const float sv0 = sacc[m][oh][jh][r], sv1 = sacc[m][oh][jh][r + 1];
#ifdef SH_DBG_NOEPI
                            const float v0 = sv0 + cj[oh][jh][r], v1 = sv1 + cj[oh][jh][r + 1];
#else
                            const float v0 = fmaf(bsig[m], cj[oh][jh][r], fmaf(c0[m], fexp2(sv0 * k0s), c1[m] * fexp2(sv0 * k1s)));
                            const float v1 = fmaf(bsig[m], cj[oh][jh][r + 1], fmaf(c0[m], fexp2(sv1 * k0s), c1[m] * fexp2(sv1 * k1s)));
#endif
                            if (CLO) {
                                unsigned hi, lo;
                                split_pair16(v0, v1, hi, lo);
                                chi[oh][p] = hi; clo[oh][p] = lo;
                            } else {
                                chi[oh][p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, f16x2));
                            }
                        }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const int k = NCT * m + ct, par = k % BD;
                        if (PIPE) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (k + BD - 1 < NCT * M) ld_b(k + BD - 1);
                            __builtin_amdgcn_sched_barrier(0);
                        } else {
                            ld_b(k);
                        }
#pragma unroll
                        for (int oh = 0; oh < OH; ++oh) {
                            f32x4 acc = gacc[GRAD ? m : 0][oh][ct];
                            if (CLO) acc = mfma_h(clo[oh], bh[par], acc);
                            acc = mfma_h(chi[oh], bl[par], acc);
                            acc = mfma_h(chi[oh], bh[par], acc);
                            gacc[GRAD ? m : 0][oh][ct] = acc;
                        }
                    }
                }
                SH_T(4)
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
#ifdef SH_DBG_TIMING
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_sh_dbg[(GRAD ? 8 : 0) + i], tacc[i]);
#endif
    if (GRAD) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float* dz = a.dZ[m];
            float zb[NCT];                                     // zbar of this lane's columns
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) zb[ct] = (ct * 16 + l15 < SH_DREAL) ? a.stat[m][ct * 16 + l15] : 0.f;
#pragma unroll
            for (int oh = 0; oh < OH; ++oh) {
                // rowsum_i = sum_j c_ij: output column 101 = lane 5 of the last column tile (same row group g4)
                float rs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rs[r] = __shfl(gacc[GRAD ? m : 0][oh][NCT - 1][r], (lane & 48) | 5, 64);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int d = ct * 16 + l15;
                    if (d < SH_DREAL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = wrow0 + oh * 16 + 4 * g4 + r;
                            if (i < own_end) atomicAdd(dz + (size_t)i * SH_DP + d, fmaf(rs[r], zb[ct], gacc[GRAD ? m : 0][oh][ct][r]) * isig[m]);
                        }
                    }
                }
            }
        }
        if (g < 2) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(gam[m]) * SH_UNPRE;
                if (lane == 0 && v != 0.f) atomicAdd(a.gamma + M * (1 + my_slot()) + m, (double)v);
            }
        }
    }
}

int fill_h(HArgs& a, const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1, bool grad,
           int a_lo, int a_hi, const char* who) {
    if (M < 2 || M > 4) { sga_set_error("%s: M=%d (the split-fp16 sweeps are built for 2, 3 or 4 modality tables)", who, M); return SGA_ERR_ARG; }
    const int own_rows = M <= 3 ? SH_OWN : SH_OWN / 2;               // owner rows per workgroup (sweeph_kernel: WV)
    if (a_lo < 0 || a_hi > A || a_lo > a_hi) { sga_set_error("%s: anchor shard [%d,%d) outside [0,%d]", who, a_lo, a_hi, A); return SGA_ERR_ARG; }
    a.M = M;
    {
        const HLayout L0 = make_hlayout(A, J1, J2);
        const size_t stat_off = (size_t)(2 * L0.nbA + L0.nb1 + L0.nb2 + 1) * SH_BLOCK;
        for (int m = 0; m < M; ++m) {
            if (!Zb[m]) { sga_set_error("%s: null table", who); return SGA_ERR_ARG; }
            a.Zb[m] = static_cast<const unsigned char*>(Zb[m]);
            a.stat[m] = reinterpret_cast<const float*>(a.Zb[m] + stat_off);
        }
    }
    a.beta = beta; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1;
    const HLayout L = make_hlayout(A, J1, J2);
    const int ns = a_hi - a_lo;
    const int bx1 = 0, bx2 = L.nbA, bn1 = 2 * L.nbA, bn2 = 2 * L.nbA + L.nb1;
    const int ox1 = 0, ox2 = A, on1 = 2 * A, on2 = 2 * A + J1;
    const HSeg N1a{bn1, 0, L.nb1, on1, on1, on1 + J1, 0}, N2a{bn2, 0, L.nb2, on2, on2, on2 + J2, 1};
    const HSeg N2b{bn2, 0, L.nb2, on2, on2, on2 + J2, 2}, N1b{bn1, 0, L.nb1, on1, on1, on1 + J1, 3};
    int g = 0;
    auto add = [&](int own0, int nown, int own_old0, int own_blk0, HSeg s0, HSeg s1) {
        if (nown <= 0) return;
        HGroup& G = a.grp[g++];
        G.own0 = own0; G.nown = nown; G.own_old0 = own_old0; G.own_blk0 = own_blk0; G.nseg = 2; G.seg[0] = s0; G.seg[1] = s1; G.nsplit = 1; G.blk0 = 0;
    };
    add(ox1 + a_lo, ns, ox1, bx1, N1a, N2a);                       // s11, s12
    add(ox2 + a_lo, ns, ox2, bx2, N2b, N1b);                       // s22, s21
    if (grad) {
        const int jl = a_lo / 32, jh = (a_hi + 31) / 32;
        const HSeg X1f0{bx1, jl, jh, ox1, ox1 + a_lo, ox1 + a_hi, 0}, X2f3{bx2, jl, jh, ox2, ox2 + a_lo, ox2 + a_hi, 3};
        const HSeg X1f1{bx1, jl, jh, ox1, ox1 + a_lo, ox1 + a_hi, 1}, X2f2{bx2, jl, jh, ox2, ox2 + a_lo, ox2 + a_hi, 2};
        add(on1, J1, on1, bn1, X1f0, X2f3);
        add(on2, J2, on2, bn2, X1f1, X2f2);
    }
    a.ngroups = g;
    // uniform ~target-step work units (one 4-wave workgroup per CU at a time); see sweepb.hip for the XCD argument
    int nwg = 0;
    for (int i = 0; i < g; ++i) {
        HGroup& G = a.grp[i];
        int steps = 0;
        for (int sg = 0; sg < G.nseg; ++sg) steps += G.seg[sg].jt_hi - G.seg[sg].jt_lo;
        int nsp = (steps + 159) / 160;
        if (nsp > steps) nsp = steps;
        if (nsp < 1) nsp = 1;
        G.nsplit = nsp;
        G.blk0 = nwg;
        nwg += (((G.nown + own_rows - 1) / own_rows) * nsp + 7) / 8 * 8;
    }
    return -nwg;                                                    // negative: number of workgroups (0 is a valid "nothing to do")
}

#ifndef SH_OH
#define SH_OH 1
#endif
template <int M, bool GRAD, bool CLO>
void launch_h(const HArgs& a, int nwg, hipStream_t s) {
    const size_t lds = (size_t)2 * M * SH_BLOCK;
    if constexpr (M == 4) {
        auto k = sweeph_kernel<M, GRAD, CLO, 1, 4>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, s, a);
    } else {
        auto k = sweeph_kernel<M, GRAD, CLO, SH_OH>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(512 / SH_OH), lds, s, a);
    }
}

}  // namespace

#ifdef SH_DBG_TIMING
extern "C" int sga_dbg_sweeph(unsigned long long* host16) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_sh_dbg), sizeof(g_sh_dbg)) != hipSuccess) return 1;
    static unsigned long long z[16];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sh_dbg), z, sizeof(z)) != hipSuccess;
}
#endif
extern "C" size_t sga_loss_split16_bytes(int A, int J1, int J2) {
    const HLayout L = make_hlayout(A, J1, J2);
    return (size_t)(2 * L.nbA + L.nb1 + L.nb2 + 1) * SH_BLOCK + SH_STAT_BYTES;      // + one block of slack + the table's statistics
}

extern "C" int sga_loss_split16_tables(const float* Z, int A, int J1, int J2, void* Zb, void* stream) {
    SGA_CHECK_ARG(A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_split16_tables: bad sizes");
    const HLayout L = make_hlayout(A, J1, J2);
    const int nb = 2 * L.nbA + L.nb1 + L.nb2;
    if (nb == 0) return SGA_OK;
    SGA_CHECK_ARG(Z && Zb, "sga_loss_split16_tables: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned char* tail = static_cast<unsigned char*>(Zb) + (size_t)nb * SH_BLOCK;
    if (hipMemsetAsync(tail, 0, SH_BLOCK + SH_STAT_BYTES, s) != hipSuccess) { sga_set_error("sga_loss_split16_tables: memset failed"); return SGA_ERR_HIP; }
    float* stat = reinterpret_cast<float*>(tail + SH_BLOCK);
    double* colsum = reinterpret_cast<double*>(tail + SH_BLOCK + 512);
    const int R = 2 * A + J1 + J2;
    hipLaunchKernelGGL(split16_colsum_kernel, dim3(R < 4096 ? (R + 63) / 64 : 1024), dim3(128), 0, s, Z, R, colsum);
    hipLaunchKernelGGL(split16_stats_kernel, dim3(1), dim3(128), 0, s, colsum, R, stat);
    hipLaunchKernelGGL(split16_tables_kernel, dim3(nb), dim3(256), 0, s, Z, A, J1, J2, static_cast<unsigned char*>(Zb), stat);
    SGA_CHECK_LAUNCH("sga_loss_split16_tables");
    return SGA_OK;
}

extern "C" int sga_loss_multi_sums_f16x2(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                         double* sums, int a_lo, int a_hi, int s_lo, void* stream) {
    SGA_CHECK_ARG(Zb && beta && sums && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_sums_f16x2: bad argument");
    SGA_CHECK_ARG(M >= 2 && M <= 4, "sga_loss_multi_sums_f16x2: M=%d (2, 3 or 4 modality tables)", M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(sums, (M + 1) * 8, s, "sga_loss_multi_sums_f16x2")) return rc0;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    HArgs a{};
    const int r = fill_h(a, Zb, M, beta, A, J1, J2, tau0, tau1, false, a_lo, a_hi, "sga_loss_multi_sums_f16x2");
    if (r > 0) return r;
    a.sums = sums;
    if (M == 2) { if (s_lo) launch_h<2, false, true>(a, -r, s); else launch_h<2, false, false>(a, -r, s); }
    else if (M == 3) { if (s_lo) launch_h<3, false, true>(a, -r, s); else launch_h<3, false, false>(a, -r, s); }
    else { if (s_lo) launch_h<4, false, true>(a, -r, s); else launch_h<4, false, false>(a, -r, s); }
    fold_slots(sums, (M + 1) * 8, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_sums_f16x2");
    return SGA_OK;
}

extern "C" int sga_loss_multi_grad_f16x2(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                         const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi, int coef_lo, void* stream) {
    SGA_CHECK_ARG(Zb && beta && gs && dZ && gamma && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_grad_f16x2: bad argument");
    SGA_CHECK_ARG(M >= 2 && M <= 4, "sga_loss_multi_grad_f16x2: M=%d (2, 3 or 4 modality tables)", M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rcz = zero_slots(gamma, M > 0 ? M : 1, s, "sga_loss_multi_grad_f16x2")) return rcz;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    HArgs a{};
    const int r = fill_h(a, Zb, M, beta, A, J1, J2, tau0, tau1, true, a_lo, a_hi, "sga_loss_multi_grad_f16x2");
    if (r > 0) return r;
    a.gs = gs; a.gamma = gamma;
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(dZ[m], "sga_loss_multi_grad_f16x2: null dZ"); a.dZ[m] = dZ[m]; }
    if (M == 2) { if (coef_lo) launch_h<2, true, true>(a, -r, s); else launch_h<2, true, false>(a, -r, s); }
    else if (M == 3) { if (coef_lo) launch_h<3, true, true>(a, -r, s); else launch_h<3, true, false>(a, -r, s); }
    else { if (coef_lo) launch_h<4, true, true>(a, -r, s); else launch_h<4, true, false>(a, -r, s); }
    fold_slots(gamma, M, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_grad_f16x2");
    return SGA_OK;
}
