// The anchors x negatives loss sweeps with every fp32 operand split EXACTLY into three bf16 terms (ops.set_mfma_mode('bf16x6'); SURVEY 7:
// "parity configs use fp32 MFMA or split-bf16 x3").  Same mathematics and the same two-owner-sweep structure as sweep16_kernel
// (contrastive.hip, exact-fp32 MFMA):
//     pass 1 (sums)   s_fam,temp[table] = sum exp(S / tau)                 S = X_own . X_other^T per modality table,
//     backward (grad) dZ[own] += C . Z[other],  C = dL/dS_m + beta_m dL/dS_J    S_J = sum_m beta_m S_m (joint table derived)
// (reference src/aligner/losses.py:5-15 and its autograd).
//
// Arithmetic.  x = h + m + l with h = bf16(x), m = bf16(x - h), l = x - h - m: 8 + 8 + 8 significand bits (round to nearest at every step,
// so |m| <= 2^-8 |x|, |l| <= 2^-16 |x|), and bf16 has fp32's exponent range -- the three terms represent EVERY fp32 value exactly, no
// pre-scale, no range condition.  A product x y is the six partial products
//     h h' + (h m' + m h') + (m m' + h l' + l h')
// on v_mfma_f32_16x16x32_bf16 into ONE fp32 accumulator (a bf16 x bf16 product is exact in fp32); the three dropped ones (m l', l m', l l')
// are <= 2^-23 |x y| in the worst case, 2^-27 typically -- below the rounding of the fp32 accumulation that both this kernel and the
// fp32 MFMA perform.  So the similarities and the gradient sums are fp32 arithmetic on the exact fp32 operands: six bf16 MFMAs of 16 cycles
// where v_mfma_f32_16x16x4_f32 needs eight of 32 cycles for the same 32 k slots (6/16 of the matrix time).
// The coefficient C = dL/dS (A operand of the gradient GEMM) is an fp32 value computed in fp32 exactly as in sweep16_kernel and is split the
// same way inside the loop: v_cvt_pk_bf16_f32 + two v_dot2c_f32_bf16 (residual = x - h, exact) per plane and pair, 7 VALU per pair.
//
// Rows are CENTRED (z' = z - zbar, bookkeeping columns 100: b = zbar . z' + |zbar|^2 / 2, 101: 1; the owner holds
// (1, b_i) so that the K tail adds b_i + b_j: the MFMAs deliver S_ij = z_i . z_j; the gradient GEMM's column 101 is rowsum_i = sum_j c_ij
// and dZ_i = sum_j c_ij z'_j + rowsum_i zbar).  With exact operands this is not needed for correctness; it keeps the accumulators of a
// table of nearly identical rows (meta_embedding_rel) small, so that its tangential gradient is not the rounding residue of a large radial
// sum -- there the centred sweep is MORE accurate than the plain fp32 one (tests/test_fp64_chunked_gpu.py).
//
// Data layout.  sga_loss_split3_tables turns a packed fp32 table Z [X1 | X2 | N1 | N2] into 32-row BLOCKS (each segment padded to whole
// blocks) of 20 480 B: [h plane 6 144 | m plane | l plane | tail image 2 048]; a plane = [K step q (3)][half jh (2)]
// [64 slots][8 bf16], slot(g, i) = 16 g + (i ^ 12 (g & 1)) holds columns 32 q + 8 g .. + 7 of row 8 (i >> 2) + 4 jh + (i & 3) (the XOR
// swizzle makes both the lane-linear ds_read_b128 of the S product and the ds_read_b64_tr_b16 transpose reads of the gradient GEMM bank-
// conflict free); the tail image = [jh][64 slots][8 bf16] of columns 96 .. 103 with k groups (h, h, m, l): against the owner's (h, m, h, h)
// and (l, 0, m, 0) two MFMAs give the six partial products of the K tail (until round 5: two images, 22 528 B per block -- every 1-KiB
// LDS-DMA costs the issuing wave ~60 cycles).
// MFMA bookkeeping: S^T tile with A = other rows from LDS, B = owner rows (registers); half jh of a 32-row tile uses A row
// i <-> other row 8 (i >> 2) + 4 jh + (i & 3), so a lane's 8 accumulator values are the 8 consecutive other rows 8 g4 .. 8 g4 + 7 = the k
// slots of the gradient MFMA, whose A operand is therefore the coefficient registers (split into three planes) and whose B operand
// comes from transpose reads of the same planes.
//
// Geometry.  Per 16 owner rows a wave holds 44 operand + 28 gradient-accumulator + 8 S registers PER TABLE; with three tables (240) plus the
// operands in flight that is more than the 256 registers of a two-wave SIMD, so the M = 3 gradient sweep runs ONE wave per SIMD (4 waves x
// 16 owner rows, <= 512 registers) with every LDS operand requested a group of MFMAs ahead inside the matrix stream (sched_barrier-pinned); M = 2 and the forward sums (no accumulators) run 8 waves, two per SIMD.
#include <stdlib.h>
#include <type_traits>

#include "loss_math.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int S3_DP = 104;
constexpr int S3_PLANE = 3 * 2 * 1024;           // 6144 B
constexpr int S3_TAIL = 3 * S3_PLANE;            // byte offset of the tail image in a block
constexpr int S3_BLOCK = S3_TAIL + 2048;         // 20480 B
constexpr int S3_NCH = S3_BLOCK / 1024;          // 20 DMA chunks
constexpr int S3_ROWSLOTS = 3 * 12 + 4;          // 16-byte slots that hold one row: 3 planes x (3 K steps x 4 k groups) + the tail image's 4 k groups

// v0, v1 -> three packed bf16 pairs, v = h + m + l EXACTLY (round to nearest at each step; the residuals are exact in fp32, the last one has
// at most 8 significant bits).  Residual = v - float(bf16): the bf16 pair is unpacked by a shift / a mask (plain VALU: beside MFMAs they cost
// their issue slot only, where v_dot2c_f32_bf16 -- "v - h" in one instruction -- costs ~10 cycles, MI355X_MICROARCH.md; S3_SPLIT_DOT2 builds
// that form.  NB hipcc 7.2 folds the packed constant 0x0000BF80 = (-1, 0) into the inline constant "-1.0", which the hardware reads as
// 0xBF800000 = (0, -1): the constants must stay opaque in SGPRs).
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    const bf16x2 H = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
    const unsigned hu = __builtin_bit_cast(unsigned, H);
#ifdef S3_SPLIT_DOT2
    unsigned c0 = 0x0000BF80u, c1 = 0xBF800000u;
    asm("" : "+s"(c0), "+s"(c1));
    const bf16x2 n0 = __builtin_bit_cast(bf16x2, c0), n1 = __builtin_bit_cast(bf16x2, c1);
    const float r0 = __builtin_amdgcn_fdot2_f32_bf16(H, n0, v0, false);
    const float r1 = __builtin_amdgcn_fdot2_f32_bf16(H, n1, v1, false);
#else
    const float r0 = v0 - __builtin_bit_cast(float, hu << 16);
    const float r1 = v1 - __builtin_bit_cast(float, hu & 0xffff0000u);
#endif
    const bf16x2 Mi = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
    const unsigned mu = __builtin_bit_cast(unsigned, Mi);
#ifdef S3_SPLIT_DOT2
    const float s0 = __builtin_amdgcn_fdot2_f32_bf16(Mi, n0, r0, false);
    const float s1 = __builtin_amdgcn_fdot2_f32_bf16(Mi, n1, r1, false);
#else
    const float s0 = r0 - __builtin_bit_cast(float, mu << 16);
    const float s1 = r1 - __builtin_bit_cast(float, mu & 0xffff0000u);
#endif
    const bf16x2 Lo = __builtin_convertvector(f32x2{s0, s1}, bf16x2);
    h = hu; m = mu; l = __builtin_bit_cast(unsigned, Lo);
}
__device__ __forceinline__ f32x4 mfma_b(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2 tr_read16(const unsigned char* p) {     // ds_read_b64_tr_b16
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p)));
}
__device__ __forceinline__ unsigned long long sreg64(unsigned long long v) {       // a wave-uniform 64-bit value, provably in SGPRs
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// scheduling directives of one column-tile group of the gradient phase: per MFMA one transpose read (the first NR MFMAs) and two VALU
template <int X, int END, int NR, bool VALU_ALL> __device__ __forceinline__ void sgb_seq() {
    if constexpr (X < END) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (X < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (VALU_ALL || X < 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        sgb_seq<X + 1, END, NR, VALU_ALL>();
    }
}
__host__ __device__ constexpr int s3_slot(int g, int i) { return 16 * g + (i ^ (12 * (g & 1))); }

struct TLayout { int nbA, nb1, nb2; };
__host__ __device__ inline TLayout make_tlayout(int A, int J1, int J2) { return TLayout{(A + 31) / 32, (J1 + 31) / 32, (J2 + 31) / 32}; }

// Statistics block of a table, behind its blocks and the slack block: float zbar[104] | float nbh (= |zbar|^2 / 2) ... | at +512 B: double colsum[104].
constexpr int S3_STAT_BYTES = 2048;
constexpr int S3_DREAL = 100;                    // data columns; 100, 101 are the bookkeeping columns (emb_dim <= 100 in this mode)

// column sums in a FIXED order (deterministic): block b sums its row range into part[b][c], one
// workgroup folds the partials in index order.
constexpr int S3_CS_BLOCKS = 256;
__global__ __launch_bounds__(128) void split3_colsum_kernel(const float* __restrict__ Z, int R, double* __restrict__ part) {
    const int c = threadIdx.x;
    const int per = (R + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    if (c >= S3_DP) return;
    double acc = 0.0;
    for (int r = r0; r < r1; ++r) acc += (double)Z[(size_t)r * S3_DP + c];
    part[(size_t)blockIdx.x * S3_DP + c] = acc;
}
__global__ __launch_bounds__(128) void split3_stats_kernel(const double* __restrict__ part, int nblocks, int R, float* __restrict__ stat) {
    __shared__ double sq[128];
    const int c = threadIdx.x;
    double cs = 0.0;
    if (c < S3_DP) for (int b = 0; b < nblocks; ++b) cs += part[(size_t)b * S3_DP + c];
    float zb = (c < S3_DREAL && R > 0) ? (float)(cs / (double)R) : 0.f;
    sq[c] = (double)zb * (double)zb;
    __syncthreads();
    __shared__ double tot;
    if (c == 0) {
        double t = 0.0;
        for (int i = 0; i < 128; ++i) t += sq[i];
        tot = t;
    }
    __syncthreads();
    // Centre only a table whose rows point the same way (|mean row|^2 >= 1/4: meta_embedding_rel's nearly identical rows have ~1).  For any
    // other table zbar = 0: the planes then hold the fp32 operands themselves -- z - zbar would round each of them once more (2^-25 |z|,
    // the same for every owner row, which the column sums of the gradient see), for no gain when the rows are spread out.
    const bool centre = tot >= 0.25;
    if (!centre) zb = 0.f;
    if (c < S3_DP) stat[c] = zb;
    if (c == 0) { stat[S3_DP] = centre ? (float)(0.5 * tot) : 0.f; stat[S3_DP + 1] = centre ? 1.f : 0.f; }
}

// fp32 packed table -> blocked bf16 h / m / l planes + the tail image of the CENTRED rows (one workgroup per 32-row block)
__global__ __launch_bounds__(256) void split3_tables_kernel(const float* __restrict__ Z, int A, int J1, int J2, unsigned char* __restrict__ Zb,
                                                            const float* __restrict__ stat, float* __restrict__ Zc) {
    __shared__ float tile[32 * S3_DP];
    __shared__ float zbar[S3_DP];
    const TLayout L = make_tlayout(A, J1, J2);
    int b = blockIdx.x, old0, len;
    if (b < L.nbA) { old0 = 0; len = A; }
    else if (b < 2 * L.nbA) { b -= L.nbA; old0 = A; len = A; }
    else if (b < 2 * L.nbA + L.nb1) { b -= 2 * L.nbA; old0 = 2 * A; len = J1; }
    else { b -= 2 * L.nbA + L.nb1; old0 = 2 * A + J1; len = J2; }
    const int nvalid = min(32, len - 32 * b);
    const float* src = Z + (size_t)(old0 + 32 * b) * S3_DP;
    if (threadIdx.x < S3_DP) zbar[threadIdx.x] = stat[threadIdx.x];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * S3_DP; e += 256) {
        const int r = e / S3_DP, c = e - r * S3_DP;
        tile[e] = (r < nvalid && c < S3_DREAL) ? src[e] - zbar[c] : 0.f;
        // the anchor rows once more as fp32 [2A, 104]: z - zbar, column 101 = 1 -- the B operand of the A x A stash products, whose output
        // column 101 is then the row sum of the coefficients (sga_loss_scatter_tangent)
        if (Zc && old0 < 2 * A && r < nvalid) Zc[(size_t)(old0 + 32 * b) * S3_DP + e] = c == S3_DREAL + 1 ? 1.f : tile[e];
    }
    __syncthreads();
    if (threadIdx.x < 32) {                       // the two bookkeeping columns of a valid row (a padding row stays all zero: S = 0)
        const int r = threadIdx.x;
        if (r < nvalid) {
            double a = 0.0;
            for (int c = 0; c < S3_DREAL; ++c) a += (double)zbar[c] * (double)tile[r * S3_DP + c];
            tile[r * S3_DP + S3_DREAL] = (float)(a + (double)stat[S3_DP]);
            tile[r * S3_DP + S3_DREAL + 1] = 1.f;
        }
    }
    __syncthreads();
    unsigned* out = reinterpret_cast<unsigned*>(Zb + (size_t)blockIdx.x * S3_BLOCK);
    // planes: dword e = (slot s of [q][jh][64], pair p of 4): stored slot (g, i ^ swz) <- logical (g, i)
    for (int e = threadIdx.x; e < 3 * 2 * 64 * 4; e += 256) {
        const int p = e & 3, st = (e >> 2) & 63, jh = (e >> 8) & 1, q = e >> 9;
        const int g = st >> 4, i = (st & 15) ^ (12 * (g & 1));
        const int row = 8 * (i >> 2) + 4 * jh + (i & 3), col = 32 * q + 8 * g + 2 * p;
        unsigned h, m, l;
        split3_pair(tile[row * S3_DP + col], tile[row * S3_DP + col + 1], h, m, l);
        out[e] = h; out[S3_PLANE / 4 + e] = m; out[2 * S3_PLANE / 4 + e] = l;
    }
    // tail image: dword e = (slot of [jh][64], pair p of 4) of columns 96 + 2 p, + 1; k groups (h, h, m, l)
    for (int e = threadIdx.x; e < 2 * 64 * 4; e += 256) {
        const int p = e & 3, st = (e >> 2) & 63, jh = (e >> 8) & 1;
        const int g = st >> 4, i = (st & 15) ^ (12 * (g & 1));
        const int row = 8 * (i >> 2) + 4 * jh + (i & 3), col = 96 + 2 * p;
        unsigned h, m, l;
        split3_pair(tile[row * S3_DP + col], tile[row * S3_DP + col + 1], h, m, l);
        out[S3_TAIL / 4 + e] = g < 2 ? h : (g == 2 ? m : l);
    }
}


// fp32 form of the same centring, for the fp32-MFMA sweeps (contrastive.hip: sga_loss_multi_sums_centred / _grad_centred): row r of Zc =
// (z - zbar [100] | b = zbar . (z - zbar) + |zbar|^2 / 2 | 1 | 0 | 0).  The owner side of those kernels reads columns 100 and 101 swapped,
// so that its K tail adds b_i + b_j and S_ij = z_i . z_j; the gradient GEMM's output column 101 is then rho_i = sum_j c_ij and columns
// 0..99 hold sum_j c_ij (z_j - zbar): the two-part gradient sga_loss_scatter_tangent projects.  One wave per row.
__global__ __launch_bounds__(256) void centre_tables_kernel(const float* __restrict__ Z, int R, const float* __restrict__ stat, float* __restrict__ Zc) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
        const float* z = Z + (size_t)r * S3_DP;
        float* o = Zc + (size_t)r * S3_DP;
        double acc = 0.0;
        float v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = lane + 64 * t;
            const float zb = d < S3_DREAL ? stat[d] : 0.f;
            v[t] = d < S3_DREAL ? z[d] - zb : 0.f;
            acc += (double)zb * (double)v[t];
        }
        acc = wave_sum_d(acc);
        o[lane] = v[0];
        const int d1 = lane + 64;
        if (d1 < S3_DP) o[d1] = d1 < S3_DREAL ? v[1] : (d1 == S3_DREAL ? (float)(acc + (double)stat[S3_DP]) : (d1 == S3_DREAL + 1 ? 1.f : 0.f));
    }
}

#ifdef S3_DBG_TIMING
__device__ unsigned long long g_s3_dbg[16];
#define S3_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; }
#else
#define S3_T(i)
#endif
struct TSeg { int blk0, jt_lo, jt_hi, old0, lo, hi, fam; };   // others: block blk0 + jt holds old rows old0 + 32 jt + w; valid rows in [lo, hi)
struct TGroup { int own0, nown, own_old0, own_blk0, blk0, nsplit, nseg; TSeg seg[2]; };
struct TArgs {
    int M; const unsigned char* Zb[4]; int perm[4]; int ngroups; TGroup grp[4];   // perm: kernel table m = the caller's table perm[m] (beta, gs, gamma)
    float k0, k1, it0, it1, ka;      // ka = k0 / k1 (the owner rows carry k1)
    const float* beta;
    double* sums;                    // SUM out  [(M+1)][8] (+ slots)
    const double* gs;                // GRAD in  [(M+1)][8]
    float* dZ[4];                    // GRAD out (fp32, old row order), atomic accumulate
    double* gamma;                   // GRAD out [M] (+ slots)
};

// WV: waves per workgroup, each owning 16 owner rows.  8: two waves per SIMD (<= 256 registers), operands requested at the top of a
// sub-step and covered by the partner wave.  4: one wave per SIMD (<= 512 registers), PIPE: operands requested a group of MFMAs ahead.
// MG (gradient sweep): the number of tables whose owner gradient THIS launch accumulates -- the first MG of the M tables; the others only
// contribute their similarities to the joint coefficient.  M = 4 (point + gat + rel + attr) runs as two launches of MG = 2 (the caller's
// tables {0, 1 | 2, 3} and {2, 3 | 0, 1}): four tables of three planes need 176 operand + 224 accumulator registers, two of them fit.
// GAM: accumulate Gamma_m = sum dL/dS_J * S_m (one of the two launches only).
// LITE (forward sums only): the similarities from the h and m planes alone -- products h h + h m + m h (+ the exact K tail), 11 instead of
// 20 MFMAs per sub-step, and the l planes are neither copied to LDS (14 of a block's 20 chunks) nor read.  Each similarity then carries an
// UNBIASED rounding of ~2^-16 (round to nearest at both splits; the dropped m m product is of that size too), i.e. exp(S / tau0) a relative 1e-4 per pair with a 1e-8 bias: over the >= 2^24
// terms the caller requires for this form (ops.BF16X6_SUMS_LITE_MIN_TERMS) a global sum moves by < 1e-7 relative -- below the fp32 rounding
// of its own accumulation.  The gradient sweep always multiplies all six products.
// FOLD (gradient sweep, one wave per SIMD): ONE set of small-product accumulators shared by all tables -- a column tile's five small products
// start from zero in every tile and are added to the tile's gacc by a plain fp32 v_add (round to nearest: unbiased, unlike the MFMA's chop)
// two column-tile groups later, under other MFMAs.  28 registers instead of 28 MG: four tables' owner gradients fit ONE launch (M = 4: every
// similarity formed once, 328 instead of 2 x 244 MFMAs per 16 x 32 pairs), at 28 VALU per table and tile.
template <int M, bool GRAD, int WV, int MG = M, bool GAM = true, bool LITE = false, bool FOLD = false>
__global__ __launch_bounds__(WV * 64, WV == 8 ? 2 : 1) void sweep3_kernel(TArgs a) {
    static_assert(!LITE || !GRAD, "LITE: the forward sums only");
    static_assert(!FOLD || (GRAD && WV == 4), "FOLD: the one-wave-per-SIMD gradient sweep only");
    constexpr int NCT = 7;
    constexpr int WAVES = WV, THREADS = WAVES * 64;
    constexpr int OWN = WV * 16;                                     // owner rows per workgroup
    constexpr int NCH = LITE ? S3_NCH - S3_PLANE / 1024 : S3_NCH;   // DMA chunks of a block that are copied (LITE: not the l plane's six)
    constexpr int KMAX = (NCH + WAVES - 1) / WAVES;                 // DMA chunk slots per table and wave
    constexpr int BUF = M * S3_BLOCK;
    constexpr bool PIPE = WV == 4;
#ifndef S3_OWN_IN_S
#define S3_OWN_IN_S 1
#endif
    constexpr bool OWN_IN_S = GRAD && PIPE && S3_OWN_IN_S;          // the own coefficient part of table m - 1 under table m's S-phase MFMAs
#ifndef S3_SPREAD
#define S3_SPREAD 1
#endif
    // SPREAD (gradient sweep, one wave per SIMD): v_exp_f32 is a quarter-rate instruction -- measured here: the 64 of a tile cost ~15 cycles each
    // (the kernel without them: -13 %) -- and only ONE of them fits under a 16-cycle MFMA.  So: the S sub-steps run in HALF-major order
    // (ss -> other half jh = ss / M, table m = ss % M); sub-step ss carries the own coefficient part of sub-step ss - 1 with one transcendental
    // per MFMA gap (never in a copy's gap); the joint coefficient of half 0 (complete after M sub-steps) rides in the copy-free gaps of the last
    // two sub-steps; the own part of the LAST sub-step runs under the first table's gradient MFMAs.  Left between the phases: the joint
    // coefficient of half 1 and the first table's split.
    constexpr bool SPREAD = OWN_IN_S && S3_SPREAD && !FOLD;      // (FOLD, M = 4: the register file is full)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];      // [2][M][S3_BLOCK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const TGroup& grp = a.grp[g];
    // XCD-aware work order (as sweep16_kernel, contrastive.hip): the group's (split major, owner block minor) work list in 8 contiguous per-XCD chunks
    const int wg_in_grp = (int)blockIdx.x - grp.blk0;
    const int nsplit = grp.nsplit, n_ob = (grp.nown + OWN - 1) / OWN, n_units = n_ob * nsplit;
    const int unit = (wg_in_grp & 7) * ((n_units + 7) >> 3) + (wg_in_grp >> 3);
    if ((wg_in_grp >> 3) >= ((n_units + 7) >> 3) || unit >= n_units) return;
    const int split = unit / n_ob;
    const int own0 = grp.own0 + (unit - split * n_ob) * OWN;
    const int own_end = grp.own0 + grp.nown;
    const int wrow0 = own0 + wave * 16;                               // this wave's first owner row
#ifdef S3_DBG_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif

    // ---- owner rows as the S product's B operand: 3 planes x 3 K = 32 steps + the two tail operands O0 = (h, m, h, h), O1 = (l, 0, m, 0)
    u32x4 opl[M][3][3], otl[M][2];
    float beta[M];
    const bool iv = wrow0 + l15 < own_end;
    {
        const int my_i = wrow0 + l15;
        const int rel = (iv ? my_i : own0) - grp.own_old0;
        const int o = rel & 31, oi = 4 * (o >> 3) + (o & 3), ojh = (o >> 2) & 1;       // row o sits at (half ojh, operand row oi) of its block
        const size_t off = (size_t)(grp.own_blk0 + (rel >> 5)) * S3_BLOCK;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned char* base = a.Zb[m] + off;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int so = ((q * 2 + ojh) * 64 + s3_slot(g4, oi)) * 16;
                    opl[m][p][q] = iv ? *reinterpret_cast<const u32x4*>(base + p * S3_PLANE + so) : u32x4{0, 0, 0, 0};
                }
            // The owner rows carry the factor k1 = log2(e) / tau1 (as sweep16_kernel's): the MFMAs then deliver the exp2 argument of the tau1 terms
            // directly and the tau0 argument is one multiply away -- one VALU less per similarity and temperature in the tile loop.  Done here, once
            // per work unit: x = h + m + l (exact), y = fl(k1 x), y split into three planes again.
            u32x4 th = iv ? *reinterpret_cast<const u32x4*>(base + S3_TAIL + (ojh * 64 + s3_slot(0, oi)) * 16) : u32x4{0, 0, 0, 0};
            u32x4 tm = iv ? *reinterpret_cast<const u32x4*>(base + S3_TAIL + (ojh * 64 + s3_slot(2, oi)) * 16) : u32x4{0, 0, 0, 0};
            u32x4 tl = iv ? *reinterpret_cast<const u32x4*>(base + S3_TAIL + (ojh * 64 + s3_slot(3, oi)) * 16) : u32x4{0, 0, 0, 0};
            const float ksc = a.k1;
            auto rescale = [&](u32x4& ph, u32x4& pm, u32x4& pl) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float x0 = (__builtin_bit_cast(float, ph[d] << 16) + __builtin_bit_cast(float, pm[d] << 16)) + __builtin_bit_cast(float, pl[d] << 16);
                    const float x1 = (__builtin_bit_cast(float, ph[d] & 0xffff0000u) + __builtin_bit_cast(float, pm[d] & 0xffff0000u)) + __builtin_bit_cast(float, pl[d] & 0xffff0000u);
                    unsigned nh, nm, nl;
                    split3_pair(x0 * ksc, x1 * ksc, nh, nm, nl);
                    ph[d] = nh; pm[d] = nm; pl[d] = nl;
                }
            };
#pragma unroll
            for (int q = 0; q < 3; ++q) rescale(opl[m][0][q], opl[m][1][q], opl[m][2][q]);
            rescale(th, tm, tl);
            // against the image's k groups (h, h, m, l):  O0 = (h, m, h, h) -> h h + h m + m h + l h;  O1 = (l, 0, m, 0) -> h l + m m
            otl[m][0] = g4 == 1 ? tm : th;
            otl[m][1] = g4 == 0 ? tl : (g4 == 2 ? tm : u32x4{0, 0, 0, 0});
#pragma unroll
            for (int t = 0; t < 2; ++t)                               // columns 100, 101: the owner holds (1, b_i) against the other's (b_j, 1)
                otl[m][t][2] = (otl[m][t][2] >> 16) | (otl[m][t][2] << 16);
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) beta[m] = a.beta[a.perm[m]];

    // Gradient accumulators, TWO per output: gacc takes the h h products, gsm the five small partial products (<= 2^-8 of them).  Why: the
    // 16-bit MFMAs align their 32 products and C to the largest exponent and CHOP what falls ~7 bits below the result's last place --
    // toward minus infinity whatever the sign (tools/micro/mfma_round_probe.hip: -0.09 ulp per MFMA whose products are 2^-8 .. 2^-16 of C;
    // the fp32 MFMA chops too, but toward zero).  Small products added straight onto the large accumulator would give every gradient entry
    // the same one-sided bias, which the column sums over 10^6 rows (the bias gradients of the layers below) would collect coherently; in
    // their own accumulator nothing is chopped, and the two are added once, in fp32, at the end.
    constexpr int NSM = GRAD ? (FOLD ? 1 : MG) : 1;                  // sets of small-product accumulators
    f32x4 gacc[GRAD ? MG : 1][NCT], gsm[NSM][NCT];
#pragma unroll
    for (int m = 0; m < (GRAD ? MG : 1); ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) gacc[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NSM; ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) gsm[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gam[M];
#pragma unroll
    for (int m = 0; m < M; ++m) gam[m] = 0.f;

    // Tile transport: ONE contiguous 22-KiB copy per table by LDS-DMA (global_load_lds, 1 KiB per wave instruction), slot (table m, k) ->
    // chunk (wave + m) % WAVES + WAVES k: the table index of every DMA is a compile-time constant.  Issued in one burst at the top of a
    // tile for the NEXT tile.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // M0 (the DMA's LDS address) must be provably uniform
    unsigned l16 = threadIdx.x;                                     // (unsigned: base + zext(offset) is what selects the saddr form)
    asm volatile("" : "+v"(l16));
    l16 = (l16 & 63u) * 16u;
    // flat copy slots [f_lo, f_hi) of the M * KMAX (table, k) slots.  The copies of the NEXT tile are spread over this tile's first S
    // sub-steps (a burst of all of them at the top of the tile queues up behind the CU's one texture-address unit -- 16 cycles per 1-KiB
    // instruction, 66 of them per tile from the four waves -- and stalled every wave's in-order stream for ~12 % of the tile,
    // S3_DBG_TIMING); BRANCH-FREE: a slot past the block's last chunk copies the wave's previous chunk again (same bytes to the same
    // place), a tile without successor copies itself into the idle buffer -- a branch here splits the MFMA stream into basic blocks, and
    // hipcc opens every block that follows a join with s_waitcnt lgkmcnt(0) right behind the operand requests it has just issued.
    unsigned long long tb[M];                                                      // scalar base of the tile being copied, per table
    auto set_tile = [&](int blk) {
#pragma unroll
        for (int m = 0; m < M; ++m) tb[m] = sreg64(reinterpret_cast<unsigned long long>(a.Zb[m]) + (unsigned long long)blk * S3_BLOCK);
    };
    auto issue_slots = [&](unsigned char* buf, int f_lo, int f_hi) {
#ifdef S3_DBG_NODMA
        return;
#endif
#pragma unroll
        for (int f = f_lo; f < f_hi; ++f) {
            if (f >= M * KMAX) break;                                               // compile time
            const int m = f / KMAX, k = f % KMAX;
            const int rot = (wave_u + m) & (WAVES - 1);
            int c = rot + k * WAVES;
            if ((k + 1) * WAVES > NCH) c = c >= NCH ? c - WAVES : c;                // only the last k can fall off the block (scalar select)
            if (LITE) c = c >= 2 * (S3_PLANE / 1024) ? c + S3_PLANE / 1024 : c;     // chunks 0..11 = h, m planes; 12..13 -> the tail image (18, 19)
            // scalar base + 32-bit lane offset (the saddr form: no 64-bit VALU address per copy)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(tb[m] + (unsigned)(c * 1024)) + l16),
                                             (__attribute__((address_space(3))) void*)(buf + m * S3_BLOCK + c * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int blk, unsigned char* buf) { set_tile(blk); issue_slots(buf, 0, M * KMAX); };

    // lane-derived LDS offsets: S product (lane-linear up to the swizzle) and the transpose reads of the gradient GEMM's B operand:
    // lane i of a 16-lane group addresses the 8-byte piece (row 8 g4 + 4 rd + (i >> 2), columns 16 ct + 4 (i & 3) ..)
    const int aoff = s3_slot(g4, l15) * 16;
    const int tr_io = 4 * g4 + (l15 >> 2), tr_cs = l15 & 3;
    const int tr_main = s3_slot(tr_cs >> 1, tr_io) * 16 + (tr_cs & 1) * 8;       // + p * PLANE + (ct >> 1) * 2048 + rd * 1024 + (ct & 1) * 512
    const int tr_tail = S3_TAIL + tr_io * 16 + (tr_cs & 1) * 8;                   // h: k group 0; m: + 512 (group 2); l: group 3 (odd: swizzled) = tr_tail_l; + rd * 1024
    const int tr_tail_l = S3_TAIL + (48 + (tr_io ^ 12)) * 16 + (tr_cs & 1) * 8;

#pragma unroll 1
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const TSeg seg = grp.seg[sg];
        float c0[M + 1], c1[M + 1];
#pragma unroll
        for (int m = 0; m <= M; ++m) {
            const int mo = m < M ? a.perm[m] : M;
            c0[m] = GRAD ? (float)(a.gs[mo * 8 + seg.fam * 2 + 0] * (double)a.it0) : 0.f;
            c1[m] = GRAD ? (float)(a.gs[mo * 8 + seg.fam * 2 + 1] * (double)a.it1) : 0.f;
        }
        const float ka = a.ka;                             // the owner rows carry k1: a similarity `sv` arrives as the tau1 exp2 argument, sv * ka is the tau0 one
        double dsum[M + 1][2];
#pragma unroll
        for (int m = 0; m <= M; ++m) { dsum[m][0] = 0.0; dsum[m][1] = 0.0; }
        // LITE: a lane's fp32 partial sums run over FOUR tiles (32 terms) before they go to the fp64 sums -- 2 (M + 1) conversions and fp64 adds
        // (half-rate VALU) per four tiles instead of per tile; the extra fp32 roundings are unbiased and ~3e-7 of a partial each
        float pl0[LITE ? M + 1 : 1], pl1[LITE ? M + 1 : 1];
#pragma unroll
        for (int m = 0; m < (LITE ? M + 1 : 1); ++m) { pl0[m] = 0.f; pl1[m] = 0.f; }

        __syncthreads();
        if (seg.jt_lo + split < seg.jt_hi) issue(seg.blk0 + seg.jt_lo + split, lds3);
        int it = 0;
#pragma unroll 1
        for (int jt = seg.jt_lo + split; jt < seg.jt_hi; jt += nsplit, ++it) {
            unsigned char* buf = lds3 + (it & 1) * BUF;
            const int j0 = seg.old0 + 32 * jt;                 // old row of the tile's first row
            S3_T(5)
            __syncthreads();                                   // tile `it` has landed (every wave waited for its own chunks), buffer it + 1 is free
            S3_T(0)
            const int next_blk = seg.blk0 + (jt + nsplit < seg.jt_hi ? jt + nsplit : jt);      // (the last tile re-copies itself: see issue_slots)
            unsigned char* next_buf = lds3 + ((it + 1) & 1) * BUF;
            set_tile(next_blk);
            S3_T(1)
            if (GRAD) {
                // A segment's first / last tile may hold rows outside [lo, hi) (uniform test).  Zeroing those rows' 16-byte slots in the
                // operand-order image (all planes + tails, all tables) makes their contributions vanish by themselves -- S = 0, c * 0 into the
                // owner gradient, 0 into Gamma -- so the gradient epilogue carries no validity mask.
                const int vlo = max(seg.lo - j0, 0), vhi = min(seg.hi - j0, 32);
                if (vlo > 0 || vhi < 32) {
                    for (int x = tid; x < M * 32 * S3_ROWSLOTS; x += THREADS) {
                        const int pc = x % S3_ROWSLOTS, w = (x / S3_ROWSLOTS) % 32, m = x / (S3_ROWSLOTS * 32);
                        if (w >= vlo && w < vhi) continue;
                        const int wjh = (w >> 2) & 1, wi = 4 * (w >> 3) + (w & 3);
                        unsigned char* base = buf + m * S3_BLOCK;
                        if (pc < 36) {
                            const int pl = pc / 12, q = (pc % 12) >> 2, gq = pc & 3;
                            *reinterpret_cast<u32x4*>(base + pl * S3_PLANE + ((q * 2 + wjh) * 64 + s3_slot(gq, wi)) * 16) = u32x4{0, 0, 0, 0};
                        } else {
                            const int gq = pc - 36;
                            *reinterpret_cast<u32x4*>(base + S3_TAIL + (wjh * 64 + s3_slot(gq, wi)) * 16) = u32x4{0, 0, 0, 0};
                        }
                    }
                    __syncthreads();
                }
            }

            // ---- S^T tiles: sacc[m][jh][r] = S_m[own = lane & 15, other = 8 g4 + 4 jh + r]; one SUB-STEP = (table, other half): 10 A operands
            // from LDS, 20 MFMAs in one accumulator chain, the small partial products first:
            //   tails | l h | m m | m h | h l | h m | h h      (other plane x owner plane).
            // PIPE: the next sub-step's operands are requested once their registers' last readers have issued (tails + l after "l h", m after
            // "m h", h after "h h"): every operand has >= 10 MFMAs to arrive.
            f32x4 sacc[M][2];
            float own[M][2][4];                                // c0 e^{S/tau0} + c1 e^{S/tau1} of the table's own similarities (GRAD)
            float cj[2][4];                                    // joint coefficient dL/dS_J of this lane's 8 pairs (GRAD)
            // A operands of a sub-step: the tail image, l, m, h planes x 3 K steps.  PIPE: two register sets; the whole set of sub-step
            // ss + 1 is requested right behind the FIRST MFMA of sub-step ss -- hipcc waits with s_waitcnt lgkmcnt(0) (never a counted wait:
            // the LDS-DMAs in flight make the counter "out of order" in its model) in front of the first MFMA that reads requested data, so
            // that one wait per sub-step must sit where nothing younger than 19 MFMAs is outstanding.
            constexpr int NSET = PIPE ? 2 : 1;
            u32x4 at[NSET], ap[NSET][3][3];                    // ap[.][0]: h, [1]: m, [2]: l
            auto ld_one = [&](int ss, int idx) {               // idx 0: the tail image; 1 + 3 p' + q: plane l, m, h (p' = 0, 1, 2) K step q
                const int e = ss % NSET;
                const unsigned char* ar = buf + (SPREAD ? ss % M : ss >> 1) * S3_BLOCK + (SPREAD ? ss / M : ss & 1) * 1024 + aoff;
                if (idx < 1) at[e] = *reinterpret_cast<const u32x4*>(ar + S3_TAIL);
                else ap[e][2 - (idx - 1) / 3][(idx - 1) % 3] = *reinterpret_cast<const u32x4*>(ar + (2 - (idx - 1) / 3) * S3_PLANE + ((idx - 1) % 3) * 2048);
            };
            auto ld_set = [&](int ss) {
#pragma unroll
                for (int idx = 0; idx < 10; ++idx) if (!(LITE && idx >= 1 && idx <= 3)) ld_one(ss, idx);        // idx 1..3: the l plane
            };
            if (PIPE) {
                ld_set(0);
                __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0) HERE, not behind the requests of set 1 (see above)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (SPREAD) {
                constexpr int NDS = 2 * M - 2, PER = (M * KMAX + NDS - 1) / NDS;
                constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
                f32x4 accp[2][2];                              // a sub-step's two chains; added up in the NEXT sub-step's first gaps (behind the MFMAs' latency)
                float ot0 = 0.f, ot1 = 0.f, sjt = 0.f, ce0 = 0.f, ce1 = 0.f;
                auto own_step = [&](int mm, int hh, int r, int k) {
                    const float sv = sacc[mm][hh][r];
                    if (k == 0) ot0 = fexp2(sv * ka);
                    else if (k == 1) ot1 = fexp2(sv);
                    else own[mm][hh][r] = fmaf(c0[mm], ot0, c1[mm] * ot1);
                };
                auto cj_step = [&](int hh, int r, int k) {
                    if (k == 0) {
                        sjt = 0.f;
#pragma unroll
                        for (int mm = 0; mm < M; ++mm) sjt = fmaf(beta[mm], sacc[mm][hh][r], sjt);
                    } else if (k == 1) ce0 = fexp2(sjt * ka);
                    else if (k == 2) ce1 = fexp2(sjt);
                    else cj[hh][r] = c0[M] * ce0 + c1[M] * ce1;
                };
#pragma unroll
                for (int ss = 0; ss < 2 * M; ++ss) {
                    const int m = ss % M, jh = ss / M, e = ss % NSET, pa = ss & 1;
                    const int pm = ss > 0 ? (ss - 1) % M : 0, pjh = ss > 0 ? (ss - 1) / M : 0;
                    accp[pa][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accp[pa][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int x = 0; x < 20; ++x) {
                        if (x == 0) accp[pa][0] = mfma_b(at[e], otl[m][1], accp[pa][0]);
                        else if (x == 1) accp[pa][1] = mfma_b(at[e], otl[m][0], accp[pa][1]);
                        else accp[pa][x & 1] = mfma_b(ap[e][PA[(x - 2) / 3]][(x - 2) % 3], opl[m][PB[(x - 2) / 3]][(x - 2) % 3], accp[pa][x & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (ss + 1 < 2 * M && x < 10) ld_one(ss + 1, x);
                        if (ss < NDS && (x % 3) == 1 && (x / 3) < PER) issue_slots(next_buf, ss * PER + (x / 3), ss * PER + (x / 3) + 1);
                        if (ss > 0) {
                            // the previous sub-step's similarities, then its own coefficient part: gaps 3 .. 18 but the copies' (1, 4, 7, 10, 13)
                            if (x == 0) { sacc[pm][pjh][0] = accp[pa ^ 1][0][0] + accp[pa ^ 1][1][0]; sacc[pm][pjh][1] = accp[pa ^ 1][0][1] + accp[pa ^ 1][1][1]; }
                            if (x == 2) { sacc[pm][pjh][2] = accp[pa ^ 1][0][2] + accp[pa ^ 1][1][2]; sacc[pm][pjh][3] = accp[pa ^ 1][0][3] + accp[pa ^ 1][1][3]; }
                            if (pm < MG) {
                                constexpr int G0[4] = {3, 8, 12, 16}, G1[4] = {5, 9, 14, 17}, G2[4] = {6, 11, 15, 18};
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (x == G0[r]) own_step(pm, pjh, r, 0);
                                    if (x == G1[r]) own_step(pm, pjh, r, 1);
                                    if (x == G2[r]) own_step(pm, pjh, r, 2);
                                }
                            }
                        }
                        if (ss >= NDS) {
                            // the joint coefficient of half 0 in the gaps the copies leave free in the last two sub-steps: slot q = 0 .. 11
                            //   q:  0      1      2      3          4      5      6          7      8      9          10     11
                            //       r0k0   r0k1   r0k2   r0k3 r1k0  r1k1   r1k2   r1k3 r2k0  r2k1   r2k2   r2k3 r3k0  r3k1   r3k2       (r3k3 behind the loop)
                            const int q = (x == 19 ? 5 : (x % 3) == 1 && x <= 13 ? x / 3 : -1);
                            if (q >= 0) {
                                const int qq = (ss - NDS) * 6 + q, r = qq / 3, k = qq % 3;
                                if (k == 0) { if (r > 0) cj_step(0, r - 1, 3); cj_step(0, r, 0); }
                                else cj_step(0, r, k);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[M - 1][1][r] = accp[(2 * M - 1) & 1][0][r] + accp[(2 * M - 1) & 1][1][r];
                cj_step(0, 3, 3);
            } else {
#pragma unroll
            for (int ss = 0; ss < 2 * M; ++ss) {
                const int m = ss >> 1, jh = ss & 1, e = ss % NSET;
                if (!PIPE) ld_set(ss);
                // 20 products, the small ones first, dealt alternately to TWO accumulator chains: between MFMAs on different accumulators
                // another instruction costs its issue slot, between two dependent ones it breaks the back-to-back forwarding (+43 cycles,
                // MI355X_MICROARCH.md) -- and a wave that is alone on its SIMD has to place 11 operand requests and up to 5 copies per
                // sub-step: one request per MFMA gap, a copy every third (program order pinned: an LDS-DMA ends a scheduling region, so
                // sched_group_barrier cannot spread across it).
                // The next tile's copies go over the first NDS sub-steps: the compiler puts s_waitcnt vmcnt(0) in front of the first transpose
                // read that follows an LDS-DMA (it cannot tell the two buffers apart), so they must have landed by the gradient phase.
                constexpr int NDS = GRAD ? (2 * M - 2) : 2 * M, PER = (M * KMAX + NDS - 1) / NDS;
                f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                float ownt[2] = {0.f, 0.f};
                // (A plane, owner plane): l h | m m | m h | h l | h m | h h
                constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
                for (int x = 0; x < 20; ++x) {
                    // LITE: h h + h m + m h only (no product with an l plane, no m m); the K tail (which carries the centring's b_i + b_j, the same
                    // for every pair of an owner row) keeps both MFMAs.  (Only the MFMA is skipped: the PIPE form's operand requests and copies below
                    // hang on the same loop index.)
                    const bool skip = LITE && x >= 2 && (PA[(x - 2) / 3] == 2 || PB[(x - 2) / 3] == 2 || (PA[(x - 2) / 3] == 1 && PB[(x - 2) / 3] == 1));
                    if (skip) {}
                    else if (x == 0) acc2[0] = mfma_b(at[e], otl[m][1], acc2[0]);
                    else if (x == 1) acc2[1] = mfma_b(at[e], otl[m][0], acc2[1]);
                    else acc2[x & 1] = mfma_b(ap[e][PA[(x - 2) / 3]][(x - 2) % 3], opl[m][PB[(x - 2) / 3]][(x - 2) % 3], acc2[x & 1]);
                    if (PIPE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (ss + 1 < 2 * M && x < 10 && !(LITE && x >= 1 && x <= 3)) ld_one(ss + 1, x);       // (LITE: not the l plane)
                        if (ss < NDS && (x % 3) == 1 && (x / 3) < PER) issue_slots(next_buf, ss * PER + (x / 3), ss * PER + (x / 3) + 1);
                        // the previous table's own coefficient part under this table's MFMAs (half jh here): three VALU per gap 11 .. 18
                        if (OWN_IN_S && m > 0 && m - 1 < MG && x >= 11 && x < 19) {
                            const int r = (x - 11) >> 1;
                            const float sv = sacc[m - 1][jh][r];
                            if (((x - 11) & 1) == 0) {
                                ownt[0] = fexp2(sv * ka);
                                ownt[1] = fexp2(sv);
                            } else {
                                own[m - 1][jh][r] = fmaf(c0[m - 1], ownt[0], c1[m - 1] * ownt[1]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (x == 1 && ss < NDS) {
                        issue_slots(next_buf, ss * PER, (ss + 1) * PER);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[m][jh][r] = acc2[0][r] + acc2[1][r];       // (element-wise: a vector add becomes v_pk_add_f32, 3 x the price beside MFMAs)
            }
            }

            S3_T(2)
            if (!GRAD) {
                float p0[M + 1], p1[M + 1];
#pragma unroll
                for (int m = 0; m <= M; ++m) { p0[m] = LITE ? pl0[LITE ? m : 0] : 0.f; p1[m] = LITE ? pl1[LITE ? m : 0] : 0.f; }
                // forward sums: exp2(0) = 1 of a padded / foreign row would count, so edge tiles are masked; interior tiles add unmasked
                auto sums_tile = [&](auto masked_c) {
                    constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = j0 + 8 * g4 + 4 * jh + r;
                            const float okf = (!MASKED || (iv && row >= seg.lo && row < seg.hi)) ? 1.f : 0.f;
                            float sj = 0.f;
#pragma unroll
                            for (int m = 0; m < M; ++m) {
                                const float sv = sacc[m][jh][r];
                                sj = fmaf(beta[m], sv, sj);
                                p0[m] = MASKED ? fmaf(okf, fexp2(sv * ka), p0[m]) : p0[m] + fexp2(sv * ka);
                                p1[m] = MASKED ? fmaf(okf, fexp2(sv), p1[m]) : p1[m] + fexp2(sv);
                            }
                            p0[M] = MASKED ? fmaf(okf, fexp2(sj * ka), p0[M]) : p0[M] + fexp2(sj * ka);
                            p1[M] = MASKED ? fmaf(okf, fexp2(sj), p1[M]) : p1[M] + fexp2(sj);
                        }
                };
                if (j0 >= seg.lo && j0 + 32 <= seg.hi && own0 + OWN <= own_end) sums_tile(std::false_type{}); else sums_tile(std::true_type{});   // uniform
                if (!LITE || (it & 3) == 3) {
#pragma unroll
                    for (int m = 0; m <= M; ++m) { dsum[m][0] += (double)p0[m]; dsum[m][1] += (double)p1[m]; }
#pragma unroll
                    for (int m = 0; m < (LITE ? M + 1 : 1); ++m) { pl0[m] = 0.f; pl1[m] = 0.f; }
                } else {
#pragma unroll
                    for (int m = 0; m < (LITE ? M + 1 : 1); ++m) { pl0[m] = p0[m]; pl1[m] = p1[m]; }
                }
            } else {
                // Gradient GEMM B operands (8 consecutive other rows 8 g4 .. + 7 of column 16 ct + c) by LDS transpose reads of the row planes,
                // one STEP = (table, column tile): 6 transpose reads, 6 MFMAs: five small partial products into gsm, h h into gacc
                //   l h | h l | m m | m h | h m || h h      (coefficient plane x row plane).
                // PIPE (one wave per SIMD: nobody else fills the issue slots a VALU instruction leaves, ~5 cycles each when a wave runs them
                // back to back, 2.6 beside MFMAs): the operands of step k + 1 are requested before the MFMAs of step k, and the coefficient
                // planes of table m + 1 are computed UNDER the MFMAs of table m (two VALU per MFMA, sched_group_barrier).
                constexpr int BD = PIPE ? 4 : 1;
                u32x4 bp[BD][3];
                auto ld_b = [&](int k) {
                    const int m = k / NCT, ct = k % NCT, par = k % BD;
                    const unsigned char* pb = buf + m * S3_BLOCK;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned char* ph = ct < 6 ? pb + p * S3_PLANE + tr_main + (ct >> 1) * 2048 + (ct & 1) * 512
                                                         : pb + (p == 2 ? tr_tail_l : tr_tail + (p == 1 ? 512 : 0));
                        const u32x2 x0 = tr_read16(ph), x1 = tr_read16(ph + 1024);
                        bp[par][p] = u32x4{x0[0], x0[1], x1[0], x1[1]};
                    }
                };
                // own coefficient parts not computed under the S phase's MFMAs: the last table's (PIPE), all of them otherwise
                auto own_last = [&]() {                         // SPREAD: the last sub-step's own part (table M - 1, half 1) -- under the first table's gradient MFMAs
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = sacc[M - 1][1][r];
                        own[M - 1][1][r] = fmaf(c0[M - 1], fexp2(sv * ka), c1[M - 1] * fexp2(sv));
                    }
                };
#pragma unroll
                for (int m = SPREAD ? M : OWN_IN_S ? M - 1 : 0; m < MG; ++m)
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sv = sacc[m][jh][r];
#ifdef S3_DBG_NOEPI
                            own[m][jh][r] = sv;
#else
                            own[m][jh][r] = fmaf(c0[m], fexp2(sv * ka), c1[m] * fexp2(sv));
#endif
                        }
                // joint coefficient dL/dS_J for this lane's 8 pairs (plain VALU: beside MFMAs a v_pk_*_f32 costs 3 x a v_fma, tools/micro/valu_issue.hip)
#pragma unroll
                for (int jh = SPREAD ? 1 : 0; jh < 2; ++jh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sj = 0.f;
#pragma unroll
                        for (int m = 0; m < M; ++m) sj = fmaf(beta[m], sacc[m][jh][r], sj);
#ifdef S3_DBG_NOEPI
                        cj[jh][r] = sj;
#else
                        cj[jh][r] = c0[M] * fexp2(sj * ka) + c1[M] * fexp2(sj);
#endif
                    }
                auto gamma_acc = [&]() {                        // Gamma_m = sum dL/dS_J * S_m (used from the anchor-owner sweeps only: each pair once;
#pragma unroll                                                 //  no branch here -- it would cut the gradient phase's scheduling region in two)
                    for (int m = 0; m < M; ++m) {
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) gam[m] = fmaf(cj[jh][r], sacc[m][jh][r], gam[m]);
                    }
                };
                if ((!PIPE || FOLD) && GAM) gamma_acc();        // (FOLD: here, so that the similarities' registers are free in the gradient phase)
                S3_T(3)
                // c_m for this lane's 8 consecutive other rows (k slot j = 4 jh + r), split into three bf16 planes: the A operand
                u32x4 ch[2], cm[2], cl[2];                     // double-buffered by table parity
                auto planes = [&](int m) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int jh = p >> 1, r = (p & 1) * 2;
#ifdef S3_DBG_NOEPI
                        const float v0 = own[m][jh][r] + cj[jh][r], v1 = own[m][jh][r + 1] + cj[jh][r + 1];
#else
                        const float v0 = fmaf(beta[m], cj[jh][r], own[m][jh][r]), v1 = fmaf(beta[m], cj[jh][r + 1], own[m][jh][r + 1]);
#endif
                        unsigned h_, m_, l_;
                        split3_pair(v0, v1, h_, m_, l_);
                        ch[m & 1][p] = h_; cm[m & 1][p] = m_; cl[m & 1][p] = l_;
                    }
                };
                planes(0);
                // (coefficient plane, row plane) of the five small products, in accumulation order
                auto step_small = [&](int m, int ct, int par, int i) {
                    const int e = m & 1;
                    const u32x4 ca = i == 0 ? cl[e] : (i == 1 || i == 4) ? ch[e] : cm[e];
                    const int pb_ = i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 1 : i == 3 ? 0 : 1;
                    const int sm = (GRAD && !FOLD) ? m : 0;
                    gsm[sm][ct] = mfma_b(ca, bp[par][pb_], (FOLD && i == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : gsm[sm][ct]);
                };
                // FOLD: the small products of column-tile group `gq` (of the table it belongs to) onto their gacc, element-wise v_add
                auto fold_group = [&](int gq) {
                    const int m = gq >> 2, ct0 = (gq & 3) * 2, n = (gq & 3) == 3 ? 1 : 2;
#pragma unroll
                    for (int j = 0; j < n; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gacc[GRAD ? m : 0][ct0 + j][r] += gsm[0][ct0 + j][r];
                };
                if (!PIPE) {
#pragma unroll
                    for (int m = 0; m < MG; ++m) {
                        if (m > 0) planes(m);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            ld_b(NCT * m + ct);
#pragma unroll
                            for (int i = 0; i < 5; ++i) step_small(m, ct, 0, i);
                            gacc[GRAD ? m : 0][ct] = mfma_b(ch[m & 1], bp[0][0], gacc[GRAD ? m : 0][ct]);
                        }
                    }
                } else {
                    // column tiles in PAIRS (0,1) (2,3) (4,5) (6): the MFMAs of a pair alternate between its two tiles, so that consecutive
                    // MFMAs never share an accumulator and the fillers between them -- the next group's 6 or 12 transpose reads, two VALU of
                    // the next table's coefficients per MFMA -- cost their issue slots only
                    __builtin_amdgcn_sched_barrier(0);
                    ld_b(0); ld_b(1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g4i = 0; g4i < 4 * MG; ++g4i) {
                        const int m = g4i >> 2, ct0 = (g4i & 3) * 2, n = (g4i & 3) == 3 ? 1 : 2, k0_ = NCT * m + ct0;
                        if (SPREAD && g4i == 0 && M - 1 < MG && M > 1) own_last();
                        if ((g4i & 3) == 0 && m + 1 < MG) planes(m + 1);       // source order only: spread under this table's MFMAs below
                        if (GAM && !FOLD && g4i == (MG > 1 ? 4 : 0)) gamma_acc();   // ... and Gamma under the second table's (the first carries planes(1))
                        if (FOLD && g4i >= 2) fold_group(g4i - 2);             // (its MFMAs finished a group ago)
                        const int kn = k0_ + n;                               // the next group's first step
                        const int nn = kn >= NCT * MG ? 0 : ((kn % NCT) == 6 ? 1 : 2);
#pragma unroll
                        for (int j = 0; j < nn; ++j) ld_b(kn + j);
                        if (n == 2) {
#pragma unroll
                            for (int i = 0; i < 5; ++i) { step_small(m, ct0, k0_ % BD, i); step_small(m, ct0 + 1, (k0_ + 1) % BD, i); }
                            gacc[GRAD ? m : 0][ct0] = mfma_b(ch[m & 1], bp[k0_ % BD][0], gacc[GRAD ? m : 0][ct0]);
                            gacc[GRAD ? m : 0][ct0 + 1] = mfma_b(ch[m & 1], bp[(k0_ + 1) % BD][0], gacc[GRAD ? m : 0][ct0 + 1]);
                        } else {
                            step_small(m, ct0, k0_ % BD, 0);
                            gacc[GRAD ? m : 0][ct0] = mfma_b(ch[m & 1], bp[k0_ % BD][0], gacc[GRAD ? m : 0][ct0]);
#pragma unroll
                            for (int i = 1; i < 5; ++i) step_small(m, ct0, k0_ % BD, i);
                        }
                        // directives: per MFMA one transpose read (while there are any) and two VALU
                        if (FOLD) {
                            // (the same directives by template recursion: in the MG = 4 instantiation hipcc leaves the loop below rolled -- a run-time
                            //  loop around nothing -- and the directives with it)
                            if (n == 2) { if (nn == 2) sgb_seq<0, 12, 12, true>(); else if (nn == 1) sgb_seq<0, 12, 6, true>(); else sgb_seq<0, 12, 0, true>(); }
                            else { if (nn == 2) sgb_seq<0, 6, 12, false>(); else if (nn == 1) sgb_seq<0, 6, 6, false>(); else sgb_seq<0, 6, 0, false>(); }
                        } else {
#pragma unroll
                        for (int x = 0; x < 6 * n; ++x) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (x < 6 * nn) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            if (n == 2 || x < 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        }
                        }
                        if ((g4i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (FOLD) { fold_group(4 * MG - 2); fold_group(4 * MG - 1); }
                }
                S3_T(4)
            }
        }
        if (!GRAD) {
#pragma unroll
            for (int m = 0; m <= M; ++m)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if (LITE) dsum[m][tt] += (double)(tt ? pl1[LITE ? m : 0] : pl0[LITE ? m : 0]);         // what the last (< 4) tiles left
                    const double v = wave_sum_d(dsum[m][tt]);
                    if (lane == 0 && v != 0.0) atomicAdd(a.sums + (M + 1) * 8 * (1 + my_slot()) + m * 8 + seg.fam * 2 + tt, v);
                }
        }
    }
#ifdef S3_DBG_TIMING
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_s3_dbg[(GRAD ? 8 : 0) + i], tacc[i]);
#endif
    if (GRAD) {
#pragma unroll
        for (int m = 0; m < MG; ++m) {
            float* dz = a.dZ[m];
            // dZ[i, 0..99] += sum_j c_ij z'_j and dZ[i, 101] += rowsum_i = sum_j c_ij (output column 101 of the last column tile): the true
            // gradient is the first + rowsum x zbar, but it is never formed -- sga_loss_scatter_tangent takes the two apart (see there)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) if (!FOLD) gacc[GRAD ? m : 0][ct] += gsm[(GRAD && !FOLD) ? m : 0][ct];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int d = ct * 16 + l15;
                if (d < S3_DREAL || d == S3_DREAL + 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = wrow0 + 4 * g4 + r;
                        if (i < own_end) atomicAdd(dz + (size_t)i * S3_DP + d, gacc[GRAD ? m : 0][ct][r]);
                    }
                }
            }
        }
        if (GAM && g < 2) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(gam[m]) / a.k1;             // Gamma was accumulated on the pre-scaled similarities
                if (lane == 0 && v != 0.f) atomicAdd(a.gamma + M * (1 + my_slot()) + a.perm[m], (double)v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The stash products of the anchors x anchors backward on the same three exact bf16 planes: out[own] += sum_oth C[own, oth] Z[oth], C a block
// of the fp32 coefficient stash the A x A kernel wrote (replaces the four fp32-MFMA GEMMs of sga_loss_stash_grad_symx; the autograd of
// losses.py:6,50-57,81-94 through S = X1 X2^T).  It IS the gradient phase of sweep3_kernel with the coefficient tile loaded instead of
// computed: 4 waves x 32 owner rows, the "other" rows' planes as 32-row tiles in LDS (one table per launch: 2 x 20 KB), the fp32
// coefficients of a wave's 16 x 32 tile split into three planes in registers (the A operand), six MFMAs per column tile, the small partial
// products in their own accumulator.  Because the planes are the sweeps' (centred where the table asks for it, column 101 = 1), the result
// arrives in the same two parts: dZ[r, 0..100) += sum c (z - zbar), dZ[r, 101] += sum c.
// One launch = up to four products (P.n), each {stash S [.][ld], transposed?, owner segment / first row / count, other segment / first row /
// count}: C[own o, oth t] = tn ? S[t ld + o] : S[o ld + t].
// ---------------------------------------------------------------------------------------------------------------------------------------
struct SProd { const float* S; int ld, tn, own_seg, own0, nown, oth_seg, oth0, noth, blk0, nsplit; };
struct SArgs { const unsigned char* Zb; float* dZ; const float* zero; int A, nbA, n; SProd p[4]; };   // zero: a readable word that holds 0.f

__global__ __launch_bounds__(256, 2) void stash3_kernel(SArgs a) {
    // 4 waves x 32 owner rows: a wave applies every B operand it reads from LDS (the other rows' planes, transposed) to TWO 16-row coefficient
    // sets.  With 16 rows per wave and 16 waves per CU the transpose reads alone were 344 KB per CU and tile round -- 2 700 cycles of the LDS's
    // 128 B per cycle, as long as the round's MFMAs (round 6: the kernel ran at 2.6 TB/s of stash reads with neither HBM nor the matrix pipe full).
    constexpr int NCT = 7, WAVES = 4, RW = 2, OWN = WAVES * RW * 16;
    constexpr int KMAX = (S3_NCH + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];      // [2][S3_BLOCK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.n && (int)blockIdx.x >= a.p[i].blk0) pi = i;
    // A REFERENCE into the kernel arguments: a copy of the struct by value keeps every field in SGPRs over the whole kernel (11 spilled SGPRs
    // with the two row sets' branches); the fields the tile loop reads (load_c) are copied once -- a scalar load inside the loop would share
    // lgkmcnt with the LDS reads.
    const SProd& P = a.p[pi];
    const float* const PS = P.S;
    const float* const zero_word = a.zero;
    const int Pld = P.ld, Ptn = P.tn, Pnown = P.nown, Poth0 = P.oth0, Pnoth = P.noth;
    const int w_in = (int)blockIdx.x - P.blk0;
    const int n_ob = (P.nown + OWN - 1) / OWN;
    if (w_in >= n_ob * P.nsplit) return;
    const int split = w_in / n_ob, ob = w_in - split * n_ob;
    const int orow0 = ob * OWN + wave * (RW * 16) + l15;              // this lane's owner rows orow0, orow0 + 16 (as the coefficient tile's rows), relative to own0
    // others: tiles of 32 rows of the segment, [t_lo, t_hi) cut into nsplit runs
    const int t_lo = P.oth0 >> 5, t_hi = (P.oth0 + P.noth + 31) >> 5;
    const int per = (t_hi - t_lo + P.nsplit - 1) / P.nsplit;
    const int jt0 = t_lo + split * per, jt1 = min(t_hi, jt0 + per);
    if (jt0 >= jt1) return;
    const int oblk0 = P.oth_seg ? a.nbA : 0;

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned l16 = threadIdx.x;
    asm volatile("" : "+v"(l16));
    l16 = (l16 & 63u) * 16u;
    auto issue = [&](int blk, unsigned char* buf) {
        const unsigned long long tb = sreg64(reinterpret_cast<unsigned long long>(a.Zb) + (unsigned long long)blk * S3_BLOCK);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            int c = wave_u + k * WAVES;
            if ((k + 1) * WAVES > S3_NCH) c = c >= S3_NCH ? c - WAVES : c;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(tb + (unsigned)(c * 1024)) + l16),
                                             (__attribute__((address_space(3))) void*)(buf + c * 1024), 16, 0, 0);
        }
    };
    // this lane's 8 coefficients of tile jt for its row set rs: owner row orow0 + 16 rs, others 32 jt + 8 g4 + k (segment rows); zero outside the product's ranges
    auto load_c = [&](int jt, int rs, float (&c)[8]) {
        const int orow = orow0 + 16 * rs;
        const bool ov = orow < Pnown;
        const int r0 = 32 * jt + 8 * g4 - Poth0;                     // stash index of k = 0
        if (!Ptn && (Pld & 3) == 0 && (Poth0 & 3) == 0 && ov && r0 >= 0 && r0 + 8 <= Pnoth) {
            const f32x4* q = reinterpret_cast<const f32x4*>(PS + (size_t)orow * Pld + r0);
            const f32x4 u = q[0], v = q[1];
            c[0] = u[0]; c[1] = u[1]; c[2] = u[2]; c[3] = u[3]; c[4] = v[0]; c[5] = v[1]; c[6] = v[2]; c[7] = v[3];
        } else {
            // (no exec-masked loads and no masks kept until the values arrive -- eight of them per call cost this kernel its scalar registers:
            //  an element outside the product's ranges is read from a word that holds zero, the tables' slack block)
            const float* base = PS + (Ptn ? (size_t)orow : (size_t)orow * Pld);
            const int stride = Ptn ? Pld : 1;
            const unsigned lim = ov ? (unsigned)Pnoth : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = r0 + k;
                const bool ok = (unsigned)t < lim;                    // one compare per element: 0 <= t < noth and the row is valid
                const float* q = ok ? base + (size_t)t * stride : zero_word;
                c[k] = *q;
            }
        }
    };

    f32x4 gacc[RW][NCT], gsm[RW][NCT];
#pragma unroll
    for (int rs = 0; rs < RW; ++rs)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { gacc[rs][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; gsm[rs][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int tr_io = 4 * g4 + (l15 >> 2), tr_cs = l15 & 3;
    const int tr_main = s3_slot(tr_cs >> 1, tr_io) * 16 + (tr_cs & 1) * 8;
    const int tr_tail = S3_TAIL + tr_io * 16 + (tr_cs & 1) * 8;
    const int tr_tail_l = S3_TAIL + (48 + (tr_io ^ 12)) * 16 + (tr_cs & 1) * 8;

    // The coefficient tiles come straight from HBM (the stash is read once per product and never again): TWO tiles in flight per wave -- a tile's
    // values are split into their planes first thing in its step and the load of the tile after next goes into the same registers, while the
    // next tile's load (issued one step earlier) is still travelling.
    float c0[RW][8], c1[RW][8];
#pragma unroll
    for (int rs = 0; rs < RW; ++rs) {
        load_c(jt0, rs, c0[rs]);
        if (jt0 + 1 < jt1) load_c(jt0 + 1, rs, c1[rs]);
    }
    issue(oblk0 + jt0, lds3);
    int it = 0;
    auto step = [&](int jt, float (&cc)[RW][8]) {
        unsigned char* buf = lds3 + (it & 1) * S3_BLOCK;
        __syncthreads();                                   // tile `it` has landed, the other buffer is free
        issue(oblk0 + (jt + 1 < jt1 ? jt + 1 : jt), lds3 + ((it + 1) & 1) * S3_BLOCK);
        u32x4 ch[RW], cm[RW], cl[RW];
#pragma unroll
        for (int rs = 0; rs < RW; ++rs)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned h_, m_, l_;
                split3_pair(cc[rs][2 * p], cc[rs][2 * p + 1], h_, m_, l_);
                ch[rs][p] = h_; cm[rs][p] = m_; cl[rs][p] = l_;
            }
        if (jt + 2 < jt1) {                                // into the registers just consumed
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) load_c(jt + 2, rs, cc[rs]);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            u32x4 bp[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned char* ph = ct < 6 ? buf + p * S3_PLANE + tr_main + (ct >> 1) * 2048 + (ct & 1) * 512
                                                 : buf + (p == 2 ? tr_tail_l : tr_tail + (p == 1 ? 512 : 0));
                const u32x2 x0 = tr_read16(ph), x1 = tr_read16(ph + 1024);
                bp[p] = u32x4{x0[0], x0[1], x1[0], x1[1]};
            }
            // (coefficient plane x row plane) l h | h l | m m | m h | h m into gsm, h h into gacc; the two row sets alternate
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) gsm[rs][ct] = mfma_b(cl[rs], bp[0], gsm[rs][ct]);
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) gsm[rs][ct] = mfma_b(ch[rs], bp[2], gsm[rs][ct]);
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) gsm[rs][ct] = mfma_b(cm[rs], bp[1], gsm[rs][ct]);
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) gsm[rs][ct] = mfma_b(cm[rs], bp[0], gsm[rs][ct]);
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) gsm[rs][ct] = mfma_b(ch[rs], bp[1], gsm[rs][ct]);
#pragma unroll
            for (int rs = 0; rs < RW; ++rs) gacc[rs][ct] = mfma_b(ch[rs], bp[0], gacc[rs][ct]);
        }
        ++it;
    };
#pragma unroll 1
    for (int jt = jt0; jt < jt1; jt += 2) {
        step(jt, c0);
        if (jt + 1 < jt1) step(jt + 1, c1);
    }
    // out rows: accumulator layout row = 4 g4 + r of the row set's 16, column 16 ct + l15
    float* dz = a.dZ + (size_t)(P.own_seg ? a.A : 0) * S3_DP;
#pragma unroll
    for (int rs = 0; rs < RW; ++rs)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int d = ct * 16 + l15;
            if (d < S3_DREAL || d == S3_DREAL + 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ob * OWN + wave * (RW * 16) + 16 * rs + 4 * g4 + r;
                    if (o < P.nown) atomicAdd(dz + (size_t)(P.own0 + o) * S3_DP + d, gacc[rs][ct][r] + gsm[rs][ct][r]);
                }
            }
        }
}

int fill_t(TArgs& a, const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1, bool grad,
           int a_lo, int a_hi, int own_rows, const char* who) {
    if (M < 2 || M > 4) { sga_set_error("%s: M=%d (the three-plane bf16 sweeps are built for 2, 3 or 4 modality tables)", who, M); return SGA_ERR_ARG; }
    if (a_lo < 0 || a_hi > A || a_lo > a_hi) { sga_set_error("%s: anchor shard [%d,%d) outside [0,%d]", who, a_lo, a_hi, A); return SGA_ERR_ARG; }
    a.M = M;
    const TLayout L = make_tlayout(A, J1, J2);
    for (int m = 0; m < M; ++m) {
        if (!Zb[m]) { sga_set_error("%s: null table", who); return SGA_ERR_ARG; }
        a.Zb[m] = static_cast<const unsigned char*>(Zb[m]);
        a.perm[m] = m;
    }
    a.beta = beta; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1; a.ka = a.k0 / a.k1;
    const int ns = a_hi - a_lo;
    const int bx1 = 0, bx2 = L.nbA, bn1 = 2 * L.nbA, bn2 = 2 * L.nbA + L.nb1;
    const int ox1 = 0, ox2 = A, on1 = 2 * A, on2 = 2 * A + J1;
    const TSeg N1a{bn1, 0, L.nb1, on1, on1, on1 + J1, 0}, N2a{bn2, 0, L.nb2, on2, on2, on2 + J2, 1};
    const TSeg N2b{bn2, 0, L.nb2, on2, on2, on2 + J2, 2}, N1b{bn1, 0, L.nb1, on1, on1, on1 + J1, 3};
    int g = 0;
    auto add = [&](int own0, int nown, int own_old0, int own_blk0, TSeg s0, TSeg s1) {
        if (nown <= 0) return;
        TGroup& G = a.grp[g++];
        G.own0 = own0; G.nown = nown; G.own_old0 = own_old0; G.own_blk0 = own_blk0; G.nseg = 2; G.seg[0] = s0; G.seg[1] = s1; G.nsplit = 1; G.blk0 = 0;
    };
    add(ox1 + a_lo, ns, ox1, bx1, N1a, N2a);                       // s11, s12
    add(ox2 + a_lo, ns, ox2, bx2, N2b, N1b);                       // s22, s21
    if (grad) {
        const int jl = a_lo / 32, jh = (a_hi + 31) / 32;
        const TSeg X1f0{bx1, jl, jh, ox1, ox1 + a_lo, ox1 + a_hi, 0}, X2f3{bx2, jl, jh, ox2, ox2 + a_lo, ox2 + a_hi, 3};
        const TSeg X1f1{bx1, jl, jh, ox1, ox1 + a_lo, ox1 + a_hi, 1}, X2f2{bx2, jl, jh, ox2, ox2 + a_lo, ox2 + a_hi, 2};
        add(on1, J1, on1, bn1, X1f0, X2f3);
        add(on2, J2, on2, bn2, X1f1, X2f2);
    }
    a.ngroups = g;
    // uniform work units; see sweep16_kernel (contrastive.hip) for the XCD argument
    int nwg = 0;
    for (int i = 0; i < g; ++i) {
        TGroup& G = a.grp[i];
        int steps = 0;
        for (int sg = 0; sg < G.nseg; ++sg) steps += G.seg[sg].jt_hi - G.seg[sg].jt_lo;
        // Work units of 160 .. 640 tile steps: the longer a unit, the fewer owner-operand loads and gradient flushes per tile (configs[2]: 2.80 ->
        // 2.74 s from 160 to 640, flat beyond; SHORTER units, whose tile stream would fit an XCD's L2 whatever the workgroups' phases, only
        // cost: 2.88 s at 56) -- as long as every CU still gets >= ~12 units of the group.  (SGA_SWEEP3_UNIT overrides, for experiments.)
        static const int unit_env = [] { const char* e = std::getenv("SGA_SWEEP3_UNIT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
        const long n_ob_g = (G.nown + own_rows - 1) / own_rows;
        const long want = (long)steps * n_ob_g / (12L * sga_num_cus());
        const int unit_steps = unit_env ? unit_env : want < 160 ? 160 : want > 640 ? 640 : (int)want;
        int nsp = (steps + unit_steps - 1) / unit_steps;
        if (nsp > steps) nsp = steps;
        if (nsp < 1) nsp = 1;
        G.nsplit = nsp;
        G.blk0 = nwg;
        nwg += (((G.nown + own_rows - 1) / own_rows) * nsp + 7) / 8 * 8;
    }
    return -nwg;                                                    // negative: number of workgroups (0 is a valid "nothing to do")
}

#ifndef S3_WV_SUMS
#define S3_WV_SUMS 8
#endif
#ifndef S3_WV_GRAD3
#define S3_WV_GRAD3 4
#endif
#ifndef S3_WV_GRAD2
#define S3_WV_GRAD2 4
#endif
// (M = 4: four tables' operands -- 176 registers -- leave no room for a second wave on the SIMD in either sweep)
template <int M, bool GRAD> constexpr int s3_wv() { return M == 4 ? 4 : GRAD ? (M == 3 ? S3_WV_GRAD3 : S3_WV_GRAD2) : S3_WV_SUMS; }

template <int M, bool GRAD, int MG = M, bool GAM = true, bool LITE = false, bool FOLD = false>
void launch_t(const TArgs& a, int nwg, hipStream_t s) {
    constexpr int WV = s3_wv<M, GRAD>();
    const size_t lds = (size_t)2 * M * S3_BLOCK;
    auto k = sweep3_kernel<M, GRAD, WV, MG, GAM, LITE, FOLD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(WV * 64), lds, s, a);
}

}  // namespace

#ifdef S3_DBG_TIMING
extern "C" int sga_dbg_sweep3(unsigned long long* host16) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_s3_dbg), sizeof(g_s3_dbg)) != hipSuccess) return 1;
    static unsigned long long z[16];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_s3_dbg), z, sizeof(z)) != hipSuccess;
}
#endif

extern "C" size_t sga_loss_split3_bytes(int A, int J1, int J2) {
    const TLayout L = make_tlayout(A, J1, J2);
    // blocks + one block of slack + the table's statistics + the column-sum partials
    return (size_t)(2 * L.nbA + L.nb1 + L.nb2 + 1) * S3_BLOCK + S3_STAT_BYTES + (size_t)S3_CS_BLOCKS * S3_DP * sizeof(double);
}

extern "C" int sga_loss_split3_tables(const float* Z, int A, int J1, int J2, void* Zb, float* Zc, void* stream) {
    SGA_CHECK_ARG(A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_split3_tables: bad sizes");
    const TLayout L = make_tlayout(A, J1, J2);
    const int nb = 2 * L.nbA + L.nb1 + L.nb2;
    if (nb == 0) return SGA_OK;
    SGA_CHECK_ARG(Z && Zb, "sga_loss_split3_tables: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned char* tail = static_cast<unsigned char*>(Zb) + (size_t)nb * S3_BLOCK;
    if (hipMemsetAsync(tail, 0, S3_BLOCK + S3_STAT_BYTES, s) != hipSuccess) { sga_set_error("sga_loss_split3_tables: memset failed"); return SGA_ERR_HIP; }
    float* stat = reinterpret_cast<float*>(tail + S3_BLOCK);
    double* part = reinterpret_cast<double*>(tail + S3_BLOCK + S3_STAT_BYTES);
    const int R = 2 * A + J1 + J2;
    const int ncs = R < 64 * S3_CS_BLOCKS ? (R + 63) / 64 : S3_CS_BLOCKS;
    hipLaunchKernelGGL(split3_colsum_kernel, dim3(ncs), dim3(128), 0, s, Z, R, part);
    hipLaunchKernelGGL(split3_stats_kernel, dim3(1), dim3(128), 0, s, part, ncs, R, stat);
    hipLaunchKernelGGL(split3_tables_kernel, dim3(nb), dim3(256), 0, s, Z, A, J1, J2, static_cast<unsigned char*>(Zb), stat, Zc);
    SGA_CHECK_LAUNCH("sga_loss_split3_tables");
    return SGA_OK;
}

extern "C" size_t sga_loss_centre_bytes(void) { return S3_STAT_BYTES + (size_t)S3_CS_BLOCKS * S3_DP * sizeof(double); }

extern "C" int sga_loss_centre_tables(const float* Z, int A, int J1, int J2, float* Zc, void* stat_ws, void* stream) {
    SGA_CHECK_ARG(A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_centre_tables: bad sizes");
    const int R = 2 * A + J1 + J2;
    SGA_CHECK_ARG(stat_ws, "sga_loss_centre_tables: null statistics workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(stat_ws, 0, S3_STAT_BYTES, s) != hipSuccess) { sga_set_error("sga_loss_centre_tables: memset failed"); return SGA_ERR_HIP; }
    if (R == 0) return SGA_OK;
    SGA_CHECK_ARG(Z && Zc, "sga_loss_centre_tables: null pointer");
    float* stat = static_cast<float*>(stat_ws);
    double* part = reinterpret_cast<double*>(static_cast<unsigned char*>(stat_ws) + S3_STAT_BYTES);
    const int ncs = R < 64 * S3_CS_BLOCKS ? (R + 63) / 64 : S3_CS_BLOCKS;
    hipLaunchKernelGGL(split3_colsum_kernel, dim3(ncs), dim3(128), 0, s, Z, R, part);
    hipLaunchKernelGGL(split3_stats_kernel, dim3(1), dim3(128), 0, s, part, ncs, R, stat);
    int grid = (R + 3) / 4;
    if (grid > 8 * sga_num_cus()) grid = 8 * sga_num_cus();
    hipLaunchKernelGGL(centre_tables_kernel, dim3(grid), dim3(256), 0, s, Z, R, stat, Zc);
    SGA_CHECK_LAUNCH("sga_loss_centre_tables");
    return SGA_OK;
}

// dE[idx[r], :] += J_normalize^T dZ[r, :] for a gradient that arrives in two parts: G = dZ[r, 0..D) = sum_j c_rj (z_j - zbar) and
// rho = dZ[r, 101] = sum_j c_rj (the true gradient of the unit row z_r is G + rho zbar).  The normalisation's Jacobian projects the
// component along z_r out: P_r (G + rho zbar) = P_r (G - rho (z_r - zbar)) because P_r z_r = 0 -- and THAT is what is evaluated.  For a table
// of nearly parallel rows (the only kind that is centred: meta_embedding_rel's bag-of-words rows) the true gradient is almost radial,
// |P_r dZ| ~ 1e-3 .. 1e-4 |dZ|: formed in fp32 first, the tangential remainder is what is left of dZ's rounding (3e-4 of the table
// gradient's maximum, 1.5-4.7 % of d meta_embedding_rel.{weight, bias} at 1024-4096 pairs against fp64 -- with fp32 MFMA sweeps and with
// exact planes alike); G and rho (z_r - zbar) are both small, nothing cancels.  An un-centred table (zbar = 0: flag at stat[105]) holds
// the true gradient in dZ[r, 0..D) and takes the plain path.
__global__ void scatter_tangent_kernel(const float* __restrict__ dZ, const float* __restrict__ Z, const float* __restrict__ nrm,
                                       const int* __restrict__ idx, int R, int D, const float* __restrict__ stat, float* __restrict__ dE) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const bool centred = stat[S3_DP + 1] != 0.f;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
        const float* g = dZ + (size_t)r * S3_DP;
        const float* z = Z + (size_t)r * S3_DP;
        const float rho = centred ? g[S3_DREAL + 1] : 0.f;
        float gv[2], zv[2];
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = lane + 64 * t;
            zv[t] = d < D ? z[d] : 0.f;
            gv[t] = d < D ? (centred ? fmaf(-rho, zv[t] - stat[d], g[d]) : g[d]) : 0.f;
            dot = fmaf(gv[t], zv[t], dot);
        }
        dot = wave_sum(dot);
        const float n = nrm[r];
        const bool clamped = n < 1e-12f;
        const float inv = 1.f / fmaxf(n, 1e-12f);
        if (clamped) dot = 0.f;
        float* o = dE + (size_t)idx[r] * D;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = lane + 64 * t;
            if (d < D) atomicAdd(o + d, (gv[t] - zv[t] * dot) * inv);
        }
    }
}

static int scatter_tangent_impl(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int R, int D, const float* stat, float* dE,
                                hipStream_t s) {
    int grid = (R + 3) / 4;
    if (grid > 8 * sga_num_cus()) grid = 8 * sga_num_cus();
    hipLaunchKernelGGL(scatter_tangent_kernel, dim3(grid), dim3(256), 0, s, dZ, Z, nrm, idx, R, D, stat, dE);
    SGA_CHECK_LAUNCH("sga_loss_scatter_tangent");
    return SGA_OK;
}

extern "C" int sga_loss_scatter_tangent(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int A, int J1, int J2, int D,
                                        const void* Zb, float* dE, void* stream) {
    SGA_CHECK_ARG(D >= 1 && D <= S3_DREAL && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_scatter_tangent: bad argument (emb_dim <= 100)");
    const int R = 2 * A + J1 + J2;
    if (R == 0) return SGA_OK;
    SGA_CHECK_ARG(dZ && Z && nrm && idx && dE && Zb, "sga_loss_scatter_tangent: null pointer");
    const TLayout L = make_tlayout(A, J1, J2);
    const float* stat = reinterpret_cast<const float*>(static_cast<const unsigned char*>(Zb) + (size_t)(2 * L.nbA + L.nb1 + L.nb2 + 1) * S3_BLOCK);
    return scatter_tangent_impl(dZ, Z, nrm, idx, R, D, stat, dE, static_cast<hipStream_t>(stream));
}

// the same with the table's statistics block given directly (the workspace of sga_loss_centre_tables)
extern "C" int sga_loss_scatter_tangent_stat(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int R, int D,
                                             const void* stat_ws, float* dE, void* stream) {
    SGA_CHECK_ARG(D >= 1 && D <= S3_DREAL && R >= 0, "sga_loss_scatter_tangent_stat: bad argument (emb_dim <= 100)");
    if (R == 0) return SGA_OK;
    SGA_CHECK_ARG(dZ && Z && nrm && idx && dE && stat_ws, "sga_loss_scatter_tangent_stat: null pointer");
    return scatter_tangent_impl(dZ, Z, nrm, idx, R, D, static_cast<const float*>(stat_ws), dE, static_cast<hipStream_t>(stream));
}

extern "C" int sga_loss_multi_sums_bf16x6(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                          double* sums, int a_lo, int a_hi, int lite, void* stream) {
    SGA_CHECK_ARG(Zb && beta && sums && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_sums_bf16x6: bad argument");
    SGA_CHECK_ARG(M >= 2 && M <= 4, "sga_loss_multi_sums_bf16x6: M=%d (2, 3 or 4 modality tables)", M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(sums, (M + 1) * 8, s, "sga_loss_multi_sums_bf16x6")) return rc0;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    TArgs a{};
    const int own_rows = 16 * (M == 4 ? s3_wv<4, false>() : M == 3 ? s3_wv<3, false>() : s3_wv<2, false>());
    const int r = fill_t(a, Zb, M, beta, A, J1, J2, tau0, tau1, false, a_lo, a_hi, own_rows, "sga_loss_multi_sums_bf16x6");
    if (r > 0) return r;
    a.sums = sums;
    if (lite) {           // h and m planes only (see sweep3_kernel, LITE): the caller vouches for >= 2^24 terms per global sum
        if (M == 2) launch_t<2, false, 2, true, true>(a, -r, s); else if (M == 3) launch_t<3, false, 3, true, true>(a, -r, s); else launch_t<4, false, 4, true, true>(a, -r, s);
    } else {
        if (M == 2) launch_t<2, false>(a, -r, s); else if (M == 3) launch_t<3, false>(a, -r, s); else launch_t<4, false>(a, -r, s);
    }
    fold_slots(sums, (M + 1) * 8, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_sums_bf16x6");
    return SGA_OK;
}

extern "C" int sga_loss_multi_grad_bf16x6(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                          const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Zb && beta && gs && dZ && gamma && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_grad_bf16x6: bad argument");
    SGA_CHECK_ARG(M >= 2 && M <= 4, "sga_loss_multi_grad_bf16x6: M=%d (2, 3 or 4 modality tables)", M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rcz = zero_slots(gamma, M, s, "sga_loss_multi_grad_bf16x6")) return rcz;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    TArgs a{};
    const int own_rows = 16 * (M == 4 ? s3_wv<4, true>() : M == 3 ? s3_wv<3, true>() : s3_wv<2, true>());
    const int r = fill_t(a, Zb, M, beta, A, J1, J2, tau0, tau1, true, a_lo, a_hi, own_rows, "sga_loss_multi_grad_bf16x6");
    if (r > 0) return r;
    a.gs = gs; a.gamma = gamma;
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(dZ[m], "sga_loss_multi_grad_bf16x6: null dZ"); a.dZ[m] = dZ[m]; }
    if (M == 2) launch_t<2, true>(a, -r, s);
    else if (M == 3) launch_t<3, true>(a, -r, s);
    else {
#ifdef S3_M4_TWO_LAUNCHES
        // four tables: two launches, each accumulating the owner gradients of two tables (all four similarities are formed in both)
        launch_t<4, true, 2, true>(a, -r, s);
        TArgs b = a;
        for (int m = 0; m < 4; ++m) { const int o = (m + 2) & 3; b.Zb[m] = a.Zb[o]; b.dZ[m] = a.dZ[o]; b.perm[m] = o; }
        launch_t<4, true, 2, false>(b, -r, s);
#else
        // four tables in ONE launch: every similarity formed once, the small-product accumulators shared between the tables (FOLD)
        launch_t<4, true, 4, true, false, true>(a, -r, s);
#endif
    }
    fold_slots(gamma, M, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_grad_bf16x6");
    return SGA_OK;
}

extern "C" int sga_loss_stash_grad_symx_bf16x6(const float* M1, const float* M2, const void* Zb, int A, int J1, int J2, float* dZ,
                                               int a_lo, int a_hi, int j_lo, int j_hi, int mir, void* stream) {
    SGA_CHECK_ARG(M1 && Zb && dZ && A >= 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi && j_lo >= 0 && j_lo <= j_hi && j_hi <= A &&
                  mir >= j_lo && (M2 || mir >= j_hi), "sga_loss_stash_grad_symx_bf16x6: bad argument");
    const int ns = a_hi - a_lo, c1 = j_hi - j_lo, c2 = mir < j_hi ? j_hi - mir : 0;
    if (A == 0 || ns == 0 || c1 == 0) return SGA_OK;
    const TLayout L = make_tlayout(A, J1, J2);
    SArgs a{};
    a.Zb = static_cast<const unsigned char*>(Zb); a.dZ = dZ; a.A = A; a.nbA = L.nbA;
    a.zero = reinterpret_cast<const float*>(a.Zb + (size_t)(2 * L.nbA + L.nb1 + L.nb2) * S3_BLOCK);      // the slack block (zeroed by sga_loss_split3_tables)
    // d1[a_lo + i] += sum_j M1[j][i] X2[j_lo + j];  d2[j_lo + j] += sum_i M1[j][i] X1[a_lo + i];
    // d1[mir + j]  += sum_i M2[j][i] X2[a_lo + i];  d2[a_lo + i] += sum_j M2[j][i] X1[mir + j]      (segments: 0 = X1, 1 = X2)
    int n = 0;
    a.p[n++] = SProd{M1, ns, 1, 0, a_lo, ns, 1, j_lo, c1, 0, 1};
    a.p[n++] = SProd{M1, ns, 0, 1, j_lo, c1, 0, a_lo, ns, 0, 1};
    if (c2 > 0) {
        a.p[n++] = SProd{M2, ns, 0, 0, mir, c2, 1, a_lo, ns, 0, 1};
        a.p[n++] = SProd{M2, ns, 1, 1, a_lo, ns, 0, mir, c2, 0, 1};
    }
    a.n = n;
    // work units of ~128 owner rows x ~64 other tiles: enough workgroups to fill the chip several times over whatever the block's shape
    int nwg = 0;
    for (int i = 0; i < n; ++i) {
        SProd& P = a.p[i];
        const int n_ob = (P.nown + 127) / 128, tiles = ((P.oth0 + P.noth + 31) >> 5) - (P.oth0 >> 5);
        int nsp = (tiles + 63) / 64;
        const int want = (4 * sga_num_cus() + n_ob - 1) / n_ob;        // at least ~4 workgroups per CU per product
        if (nsp < want) nsp = want;
        if (nsp > tiles) nsp = tiles;
        if (nsp < 1) nsp = 1;
        P.nsplit = nsp; P.blk0 = nwg;
        nwg += n_ob * nsp;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)2 * S3_BLOCK;
    hipFuncSetAttribute(reinterpret_cast<const void*>(stash3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stash3_kernel, dim3(nwg), dim3(256), lds, s, a);
    SGA_CHECK_LAUNCH("sga_loss_stash_grad_symx_bf16x6");
    return SGA_OK;
}
