// Opt-in split-bf16 x3 form of the anchors x negatives loss sweeps (sga_set_mfma_mode(1); the default stays the exact-fp32
// sweep16_kernel in contrastive.hip).  Same mathematics and the same two-owner-sweep structure:
//     pass 1 (sums)   s_fam,temp[table] = sum exp(S / tau)                 S = X_own . X_other^T per modality table,
//     backward (grad) dZ[own] += C . Z[other],  C = dL/dS_m + beta_m dL/dS_J    S_J = sum_m beta_m S_m (joint table derived)
// (reference src/aligner/losses.py:5-15 and its autograd), but every fp32 operand enters the matrix cores as bf16 hi + lo
// (v = hi + lo, 16 significand bits) and a product is three bf16 MFMAs into the same fp32 accumulator: hi*hi + hi*lo + lo*hi.
// fp32 MFMA runs at 1/16 of the bf16 rate, so the MFMA time of a tile drops ~4.7x and the (unchanged, fp32) exp2/coefficient
// epilogue becomes the larger part.
//
// Data layout.  sga_loss_split_tables turns a packed fp32 table Z [X1 | X2 | N1 | N2] into 32-row BLOCKS, each segment padded
// to whole blocks; a block holds two bf16 planes (hi | lo) of 6 656 B, 13 312 B contiguous, stored in the S product's MFMA OPERAND
// ORDER, so that its LDS reads are lane-linear (lane L reads 16 B at base + 16 L: ds_read_b128 conflict-free by construction):
//     [K step q (3)][half jh (2)][lane = 16 g4 + i][8 bf16] = columns 32 q + 8 g4 .. + 7 of row 8 (i>>2) + 4 jh + (i&3),
//     then the K = 16 tail [jh][g4 < 2][i][4 bf16] = columns 96 + 4 g4 .. + 3.
// A tile of "other" rows is ONE contiguous 13 KiB copy per table (13 global_load_lds DMA chunks; the LDS-DMA path sustains only
// ~18 B/clk per CU, which is why there is no second, transposed copy of the tile), double-buffered in LDS (3 tables: 78 KiB).
// The gradient GEMM's B operand -- 8 consecutive "other" rows of one column per lane, i.e. the TRANSPOSE of how a row is stored --
// comes from the same planes through gfx950's LDS transpose read: ds_read_b64_tr_b16 hands lane c of a 16-lane group element
// (c & 3) of the 8-byte pieces addressed by lanes 4 j + (c >> 2), j = 0..3 (tools/micro/tr16_probe.hip), so with lane i pointing at
// (row r0 + (i >> 2), columns col0 + 4 (i & 3) ..) lane c receives rows r0 .. r0 + 3 of column col0 + c: two such reads per plane
// are one 16x16x32 B operand.
// MFMA bookkeeping (v_mfma_f32_16x16x32_bf16; A: lane&15 = row, B: lane&15 = column, lane>>4 = k group of 8 slots):
//   S^T tile: A = other rows from LDS, B = owner rows (registers).  Half jh of a 32-row tile uses A row i <-> other row
//   8 (i>>2) + 4 jh + (i&3), so that a lane's 8 accumulator values (2 halves x 4) are the 8 CONSECUTIVE other rows 8 g4 .. 8 g4+7:
//   exactly the k slots of the gradient MFMA  dZ[own] += C[own, other] Z[other, cols]  whose A operand is therefore the
//   coefficient registers (split into bf16 hi/lo, 6 VALU per pair) and whose B operand comes from the transpose reads above.
#include <type_traits>

#include "loss_math.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef SB_WAVES_N
#define SB_WAVES_N 8
#endif
constexpr int SB_WAVES = SB_WAVES_N, SB_THREADS = SB_WAVES * 64, SB_OWN = SB_WAVES * 16;
constexpr int SB_ROT = SB_WAVES == 8 ? 3 : 1;    // per-table rotation of the DMA chunk -> wave assignment (spreads the waves that get one chunk more)
constexpr int sb_min_chunks(int M, int nch) {    // fewest DMA chunks any wave issues per tile under that assignment
    int best = 1 << 30;
    for (int w = 0; w < SB_WAVES; ++w) {
        int n = 0;
        for (int m = 0; m < M; ++m)
            for (int c = (w + SB_ROT * m) & (SB_WAVES - 1); c < nch; c += SB_WAVES) ++n;
        best = n < best ? n : best;
    }
    return best;
}
#ifndef SB_NBUF
#define SB_NBUF 2
#endif
constexpr int SB_DP = 104;
constexpr int SB_ROWB = SB_DP * 2;               // bytes of a row in an R plane
constexpr int SB_PLANE = 32 * SB_ROWB;           // 6656 B
constexpr int SB_BLOCK = 2 * SB_PLANE;           // 13312 B: hi | lo

__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    const bf16x2 h = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const float b0 = __builtin_bit_cast(float, hi << 16), b1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0 - b0, v1 - b1}, bf16x2));
}
__device__ __forceinline__ f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {     // ds_read_b64_tr_b16
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p)));
}
__device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4, a), __builtin_bit_cast(bf16x4, b), c, 0, 0, 0);
}

struct BLayout { int nbA, nb1, nb2; };
__host__ __device__ inline BLayout make_layout(int A, int J1, int J2) { return BLayout{(A + 31) / 32, (J1 + 31) / 32, (J2 + 31) / 32}; }

// fp32 packed table -> blocked bf16 hi/lo planes (one workgroup per 32-row block)
__global__ __launch_bounds__(256) void split_tables_kernel(const float* __restrict__ Z, int A, int J1, int J2, unsigned char* __restrict__ Zb) {
    __shared__ float tile[32 * SB_DP];
    const BLayout L = make_layout(A, J1, J2);
    int b = blockIdx.x, old0, len;
    if (b < L.nbA) { old0 = 0; len = A; }
    else if (b < 2 * L.nbA) { b -= L.nbA; old0 = A; len = A; }
    else if (b < 2 * L.nbA + L.nb1) { b -= 2 * L.nbA; old0 = 2 * A; len = J1; }
    else { b -= 2 * L.nbA + L.nb1; old0 = 2 * A + J1; len = J2; }
    const int nvalid = min(32, len - 32 * b);
    const float* src = Z + (size_t)(old0 + 32 * b) * SB_DP;
    for (int e = threadIdx.x; e < 32 * SB_DP; e += 256) tile[e] = (e / SB_DP) < nvalid ? src[e] : 0.f;
    __syncthreads();
    unsigned* out = reinterpret_cast<unsigned*>(Zb + (size_t)blockIdx.x * SB_BLOCK);
    constexpr int PD = SB_PLANE / 4;                  // dwords per plane
    // main part: dword (q, jh, g4, i, pair p of 4): columns 32 q + 8 g4 + 2 p, +1 of row 8 (i>>2) + 4 jh + (i&3)
    for (int e = threadIdx.x; e < 3 * 2 * 64 * 4; e += 256) {
        const int p = e & 3, ln = (e >> 2) & 63, jh = (e >> 8) & 1, q = e >> 9;
        const int i = ln & 15, g4 = ln >> 4;
        const int row = 8 * (i >> 2) + 4 * jh + (i & 3), col = 32 * q + 8 * g4 + 2 * p;
        unsigned hi, lo;
        split_pair(tile[row * SB_DP + col], tile[row * SB_DP + col + 1], hi, lo);
        out[e] = hi; out[PD + e] = lo;
    }
    // K = 16 tail: dword (jh, g4 < 2, i, pair p of 2): columns 96 + 4 g4 + 2 p, +1
    for (int e = threadIdx.x; e < 2 * 2 * 16 * 2; e += 256) {
        const int p = e & 1, i = (e >> 1) & 15, g4 = (e >> 5) & 1, jh = e >> 6;
        const int row = 8 * (i >> 2) + 4 * jh + (i & 3), col = 96 + 4 * g4 + 2 * p;
        unsigned hi, lo;
        split_pair(tile[row * SB_DP + col], tile[row * SB_DP + col + 1], hi, lo);
        out[1536 + e] = hi; out[PD + 1536 + e] = lo;
    }
}

struct BSeg { int blk0, jt_lo, jt_hi, old0, lo, hi, fam; };   // others: block blk0 + jt holds old rows old0 + 32 jt + w; valid rows in [lo, hi)
struct BGroup { int own0, nown, own_old0, own_blk0, blk0, nsplit, nseg; BSeg seg[2]; };
struct BArgs {
    int M; const unsigned char* Zb[4]; int ngroups; BGroup grp[4];
    float k0, k1, it0, it1;
    const float* beta;
    double* sums;                    // SUM out  [(M+1)][8] (+ slots)
    const double* gs;                // GRAD in  [(M+1)][8]
    float* dZ[4];                    // GRAD out (fp32, old row order), atomic accumulate
    double* gamma;                   // GRAD out [M] (+ slots)
};

template <int M, bool GRAD>
__global__ __launch_bounds__(SB_THREADS, SB_WAVES == 4 ? 2 : 1) void sweepb_kernel(BArgs a) {
    constexpr int NCT = 7;
    constexpr int TBYTES = SB_BLOCK;
    constexpr int BUF = M * TBYTES, NCH = TBYTES / 1024;
    constexpr int NBUF = SB_NBUF;                                             // ring depth: tiles it+1 .. it+NBUF-1 are in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];      // [NBUF][M][TBYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const BGroup& grp = a.grp[g];
    // XCD-aware work order, as in contrastive.hip's sweep16_kernel: the group's (split major, owner block minor) work list in 8
    // contiguous per-XCD chunks; groups are padded to a multiple of 8 workgroups, the padding exits here.
    const int wg_in_grp = (int)blockIdx.x - grp.blk0;
    const int nsplit = grp.nsplit, n_ob = (grp.nown + SB_OWN - 1) / SB_OWN, n_units = n_ob * nsplit;
    const int unit = (wg_in_grp & 7) * ((n_units + 7) >> 3) + (wg_in_grp >> 3);
    if ((wg_in_grp >> 3) >= ((n_units + 7) >> 3) || unit >= n_units) return;
    const int split = unit / n_ob;
    const int own0 = grp.own0 + (unit - split * n_ob) * SB_OWN;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + wave * 16 + l15;
    const bool iv = my_i < own_end;

    // ---- owner rows as the S product's B operand: 3 K = 32 steps (8 columns per lane) + the K = 16 tail (columns 96 + 4 g4 .. +3)
    // OLDS: the gradient build for three tables needs 84 (owner hi/lo operands) + 84 (accumulators) + 24 (S tiles) registers before any
    // temporary -- 39 spilled, with scratch reloads (and their vmcnt waits) inside the tile loop.  There the owner rows' LO planes live in
    // the 80 KiB of LDS the two-deep ring leaves free instead: [table][K step][thread] 16-byte slots, written and read by the same
    // lane only (lane-linear ds_read_b128: conflict free, no barrier), 9 more LDS reads per half tile.
    constexpr bool OLDS = GRAD && M == 3;
    u32x4 ohi[M][3], olo[OLDS ? 1 : M][3];
    u32x2 othi[M], otlo[M];
    float beta[M];
    u32x4* const olds = reinterpret_cast<u32x4*>(ldsb + NBUF * BUF) + tid;       // + (m * 3 + q) * SB_THREADS
    {
        const int rel = (iv ? my_i : own0) - grp.own_old0;
        const int o = rel & 31, oi = 4 * (o >> 3) + (o & 3), ojh = (o >> 2) & 1;       // row o sits at (half ojh, operand row oi) of its block
        const size_t off = (size_t)(grp.own_blk0 + (rel >> 5)) * SB_BLOCK;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned char* base = a.Zb[m] + off;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ohi[m][q] = *reinterpret_cast<const u32x4*>(base + ((q * 2 + ojh) * 64 + g4 * 16 + oi) * 16);
                u32x4 lo4 = *reinterpret_cast<const u32x4*>(base + SB_PLANE + ((q * 2 + ojh) * 64 + g4 * 16 + oi) * 16);
                if (!iv) { ohi[m][q] = u32x4{0, 0, 0, 0}; lo4 = u32x4{0, 0, 0, 0}; }
                if (OLDS) olds[(m * 3 + q) * SB_THREADS] = lo4; else olo[OLDS ? 0 : m][q] = lo4;
            }
            const bool tv = iv && g4 < 2;                      // columns 96..103 only: the k slots of lanes 32..63 multiply zeros
            othi[m] = tv ? *reinterpret_cast<const u32x2*>(base + 6144 + ((ojh * 2 + (g4 & 1)) * 16 + oi) * 8) : u32x2{0, 0};
            otlo[m] = tv ? *reinterpret_cast<const u32x2*>(base + SB_PLANE + 6144 + ((ojh * 2 + (g4 & 1)) * 16 + oi) * 8) : u32x2{0, 0};
            beta[m] = a.beta[m];
        }
    }
    f32x4 gacc[GRAD ? M : 1][NCT];
#pragma unroll
    for (int m = 0; m < (GRAD ? M : 1); ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) gacc[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gam[M];
#pragma unroll
    for (int m = 0; m < M; ++m) gam[m] = 0.f;

    // transposed-read piece of this lane (see the header): operand row i_o = 4 g4 + (l15 >> 2), column sub-piece l15 & 3
    // (per tile below: tr_main = (cs >> 1) * 256 + io * 16 + (cs & 1) * 8  [+ (ct >> 1) * 2048 + (ct & 1) * 512 + rd * 1024] and
    //  tr_tail = 6144 + (cs & 1) * 128 + io * 8  [+ rd * 256; columns 96..103, sub-pieces 2, 3 alias 0, 1], io = 4 g4 + (l15 >> 2), cs = l15 & 3)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // M0 (the DMA's LDS address) must be provably uniform
    // Wave w fetches chunks (w + SB_ROT m) % WAVES + WAVES k of table m (as contrastive.hip's sweep16_kernel: the table index of every DMA
    // is a compile-time constant, so no kernel-argument reload + s_waitcnt sits in front of it); sb_min_chunks() is what every wave
    // issues at least per tile -- the counted vmcnt wait below relies on it.
    auto issue = [&](int blk, unsigned char* buf) {
        int l16 = threadIdx.x;
        asm volatile("" : "+v"(l16));         // lane offset recomputed here, not kept (and spilled) as part of per-lane 64-bit bases
        l16 = (l16 & 63) * 16;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int rot = (wave_u + SB_ROT * m) & (SB_WAVES - 1);
            const unsigned char* src = a.Zb[m] + ((size_t)blk * SB_BLOCK + rot * 1024) + l16;
            unsigned char* dst = buf + m * TBYTES + rot * 1024;
#pragma unroll
            for (int k = 0; k * SB_WAVES < NCH; ++k) {
                if ((k + 1) * SB_WAVES > NCH && rot + k * SB_WAVES >= NCH) break;       // uniform; only the last k can fall off the table
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * SB_WAVES * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + k * SB_WAVES * 1024), 16, 0, 0);
            }
        }
    };

#pragma unroll 1
    for (int sg = 0; sg < 2; ++sg) {
        if (sg >= grp.nseg) break;
        const BSeg seg = grp.seg[sg];
        float c0[M + 1], c1[M + 1];
#pragma unroll
        for (int m = 0; m <= M; ++m) {
            c0[m] = GRAD ? (float)(a.gs[m * 8 + seg.fam * 2 + 0] * (double)a.it0) : 0.f;
            c1[m] = GRAD ? (float)(a.gs[m * 8 + seg.fam * 2 + 1] * (double)a.it1) : 0.f;
        }
        double dsum[M + 1][2];
#pragma unroll
        for (int m = 0; m <= M; ++m) { dsum[m][0] = 0.0; dsum[m][1] = 0.0; }

        __syncthreads();
#pragma unroll
        for (int d = 0; d < NBUF - 1; ++d)
            if (seg.jt_lo + split + d * nsplit < seg.jt_hi) issue(seg.blk0 + seg.jt_lo + split + d * nsplit, ldsb + d * BUF);
        int it = 0;
#pragma unroll 1
        for (int jt = seg.jt_lo + split; jt < seg.jt_hi; jt += nsplit, ++it) {
            unsigned char* buf = ldsb + (it % NBUF) * BUF;
            const int j0 = seg.old0 + 32 * jt;                 // old row of the tile's first row
            // Three-table gradient build: the lane-derived LDS offsets are recomputed per tile from an opaque copy of the thread id (~10
            // VALU).  As loop invariants they are the first thing the allocator spills at 256 registers, and every scratch reload in
            // this loop brings a vmcnt wait that also stalls on the DMA ring.
            int tid_t = tid;
            if (OLDS) asm volatile("" : "+v"(tid_t));
            const int lane_t = tid_t & 63, g4_t = lane_t >> 4, l15_t = lane_t & 15;
            const int tr_io_t = 4 * g4_t + (l15_t >> 2), tr_cs_t = l15_t & 3;
            const int tr_main_t = (tr_cs_t >> 1) * 256 + tr_io_t * 16 + (tr_cs_t & 1) * 8;
            const int tr_tail_t = 6144 + (tr_cs_t & 1) * 128 + tr_io_t * 8;
            u32x4* const olds_t = reinterpret_cast<u32x4*>(ldsb + NBUF * BUF) + tid_t;
            // tile `it` must have landed: all but the NBUF-2 most recently issued tiles of this wave are complete (a wave's DMA
            // chunks return in order), then the barrier publishes every wave's chunks; it also frees buffer (it-1) % NBUF
            if (NBUF == 2) {
                __syncthreads();
            } else {
                constexpr int keep = sb_min_chunks(M, NCH) * (NBUF - 2);    // every wave issues at least that many chunks per tile
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(keep) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (jt + (NBUF - 1) * nsplit < seg.jt_hi) issue(seg.blk0 + jt + (NBUF - 1) * nsplit, ldsb + ((it + NBUF - 1) % NBUF) * BUF);
            if (GRAD) {
                // A segment's first / last tile may hold rows outside [lo, hi) (uniform test).  Zeroing those rows' pieces in the operand-order
                // image (both planes, all tables; layout as in split_tables_kernel) makes their contributions vanish by themselves -- S = 0,
                // c * 0 into the owner gradient, 0 into Gamma -- so the gradient epilogue carries no validity mask (8 registers, 32 multiplies).
                const int vlo = max(seg.lo - j0, 0), vhi = min(seg.hi - j0, 32);
                if (vlo > 0 || vhi < 32) {
                    for (int x = tid; x < M * 2 * 32 * 14; x += SB_THREADS) {
                        const int pc = x % 14, w = (x / 14) % 32, pl = (x / (14 * 32)) % 2, m = x / (14 * 32 * 2);
                        if (w >= vlo && w < vhi) continue;
                        const int wjh = (w >> 2) & 1, wi = 4 * (w >> 3) + (w & 3);
                        unsigned char* base = buf + m * TBYTES + pl * SB_PLANE;
                        if (pc < 12) *reinterpret_cast<u32x4*>(base + (((pc >> 2) * 2 + wjh) * 64 + (pc & 3) * 16 + wi) * 16) = u32x4{0, 0, 0, 0};
                        else *reinterpret_cast<u32x2*>(base + 6144 + ((wjh * 2 + (pc - 12)) * 16 + wi) * 8) = u32x2{0, 0};
                    }
                    __syncthreads();
                }
            }

            // ---- S^T tiles: sacc[m][jh][r] = S_m[own = lane&15, other = 8 g4 + 4 jh + r].  Per half, the M main chains and the M
            // tail chains are interleaved (a dependent MFMA is M issues away) and a K step's operands are read for all tables first.
            f32x4 sacc[M][2];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const unsigned char* ar = buf + (jh * 64 + lane_t) * 16;        // lane-linear: [q][jh][lane][16 B]
                f32x4 acc[M], tacc[M];
#pragma unroll
                for (int m = 0; m < M; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; tacc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                // tail K = 16: only lanes g4 < 2 carry columns 96..103; lanes 32..63 multiply the owner's zeros and re-read the same
                // (finite) data.  It runs in its OWN accumulator chain: chaining v_mfma_*_16x16x16 onto an accumulator that a
                // v_mfma_*_16x16x32 has just written gave schedule-dependent wrong sums on gfx950 (mixed-type SrcC forwarding).
                const unsigned char* at = buf + 6144 + ((jh * 2 + (g4_t & 1)) * 16 + l15_t) * 8;
                u32x2 th[M], tl[M];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    th[m] = *reinterpret_cast<const u32x2*>(at + m * TBYTES);
                    tl[m] = *reinterpret_cast<const u32x2*>(at + m * TBYTES + SB_PLANE);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x4 ah[M], al[M];
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        ah[m] = *reinterpret_cast<const u32x4*>(ar + m * TBYTES + q * 2048);
                        al[m] = *reinterpret_cast<const u32x4*>(ar + m * TBYTES + SB_PLANE + q * 2048);
                    }
#pragma unroll
                    for (int m = 0; m < M; ++m) acc[m] = mfma32(ah[m], ohi[m][q], acc[m]);
#pragma unroll
                    for (int m = 0; m < M; ++m) tacc[m] = q == 0 ? mfma16(th[m], othi[m], tacc[m]) : (q == 1 ? mfma16(th[m], otlo[m], tacc[m]) : mfma16(tl[m], othi[m], tacc[m]));
#pragma unroll
                    for (int m = 0; m < M; ++m) acc[m] = mfma32(ah[m], OLDS ? olds_t[(m * 3 + q) * SB_THREADS] : olo[OLDS ? 0 : m][q], acc[m]);
#pragma unroll
                    for (int m = 0; m < M; ++m) acc[m] = mfma32(al[m], ohi[m][q], acc[m]);
                }
#pragma unroll
                for (int m = 0; m < M; ++m) sacc[m][jh] = acc[m] + tacc[m];
            }

            if (!GRAD) {
                float p0[M + 1], p1[M + 1];
#pragma unroll
                for (int m = 0; m <= M; ++m) { p0[m] = 0.f; p1[m] = 0.f; }
                // forward sums: exp2(0) = 1 of a padded / foreign row would count, so edge tiles are masked; interior tiles (every owner row of
                // the block valid, all 32 other rows inside [lo, hi)) add unmasked
                auto sums_tile = [&](auto masked_c) {
                    constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = j0 + 8 * g4_t + 4 * jh + r;
                            const float okf = (!MASKED || (iv && row >= seg.lo && row < seg.hi)) ? 1.f : 0.f;
                            float sj = 0.f;
#pragma unroll
                            for (int m = 0; m < M; ++m) {
                                const float sv = sacc[m][jh][r];
                                sj = fmaf(beta[m], sv, sj);
                                p0[m] = MASKED ? fmaf(okf, fexp2(sv * a.k0), p0[m]) : p0[m] + fexp2(sv * a.k0);
                                p1[m] = MASKED ? fmaf(okf, fexp2(sv * a.k1), p1[m]) : p1[m] + fexp2(sv * a.k1);
                            }
                            p0[M] = MASKED ? fmaf(okf, fexp2(sj * a.k0), p0[M]) : p0[M] + fexp2(sj * a.k0);
                            p1[M] = MASKED ? fmaf(okf, fexp2(sj * a.k1), p1[M]) : p1[M] + fexp2(sj * a.k1);
                        }
                };
                if (j0 >= seg.lo && j0 + 32 <= seg.hi && own0 + SB_OWN <= own_end) sums_tile(std::false_type{}); else sums_tile(std::true_type{});   // uniform
#pragma unroll
                for (int m = 0; m <= M; ++m) { dsum[m][0] += (double)p0[m]; dsum[m][1] += (double)p1[m]; }
            } else {
                float cj[2][4];
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sj = 0.f;
#pragma unroll
                        for (int m = 0; m < M; ++m) sj = fmaf(beta[m], sacc[m][jh][r], sj);
                        cj[jh][r] = c0[M] * fexp2(sj * a.k0) + c1[M] * fexp2(sj * a.k1);
                    }
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    // c_m for this lane's 8 consecutive other rows (k slot j = 4 jh + r), split into bf16 hi / lo: the A operand
                    u32x4 chi, clo;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int jh = p >> 1, r = (p & 1) * 2;
                        const float sv0 = sacc[m][jh][r], sv1 = sacc[m][jh][r + 1];
                        const float v0 = fmaf(beta[m], cj[jh][r], c0[m] * fexp2(sv0 * a.k0) + c1[m] * fexp2(sv0 * a.k1));
                        const float v1 = fmaf(beta[m], cj[jh][r + 1], c0[m] * fexp2(sv1 * a.k0) + c1[m] * fexp2(sv1 * a.k1));
                        unsigned hi, lo;
                        split_pair(v0, v1, hi, lo);
                        chi[p] = hi; clo[p] = lo;
                    }
                    // B operand (8 consecutive other rows 8 g4 .. + 7 of column 16 ct + c) by LDS transpose reads of the row planes: lane i
                    // of a 16-lane group addresses the 8-byte piece (row 8 g4 + 4 rd + (i >> 2), columns 16 ct + 4 (i & 3) ..)
                    const unsigned char* tb = buf + m * TBYTES + tr_main_t;
                    const unsigned char* tb6 = buf + m * TBYTES + tr_tail_t;
                    constexpr int CU = OLDS ? 2 : 4;             // column tiles in flight (their hi/lo B operands are 8 registers each)
#pragma unroll
                    for (int c0t = 0; c0t < NCT; c0t += CU) {
                        u32x4 bh[CU], bl[CU];
#pragma unroll
                        for (int u = 0; u < CU; ++u) {
                            const int ct = c0t + u;
                            if (ct < NCT) {
                                const unsigned char* p0 = ct < 6 ? tb + (ct >> 1) * 2048 + (ct & 1) * 512 : tb6;
                                const int rstep = ct < 6 ? 1024 : 256;                      // rd = 1: the other half (jh) of the operand-order image
                                const u32x2 h0 = tr_read(p0), h1 = tr_read(p0 + rstep);
                                const u32x2 l0 = tr_read(p0 + SB_PLANE), l1 = tr_read(p0 + SB_PLANE + rstep);
                                bh[u] = u32x4{h0[0], h0[1], h1[0], h1[1]};
                                bl[u] = u32x4{l0[0], l0[1], l1[0], l1[1]};
                            }
                        }
#pragma unroll
                        for (int u = 0; u < CU; ++u) if (c0t + u < NCT) gacc[GRAD ? m : 0][c0t + u] = mfma32(chi, bh[u], gacc[GRAD ? m : 0][c0t + u]);
#pragma unroll
                        for (int u = 0; u < CU; ++u) if (c0t + u < NCT) gacc[GRAD ? m : 0][c0t + u] = mfma32(chi, bl[u], gacc[GRAD ? m : 0][c0t + u]);
#pragma unroll
                        for (int u = 0; u < CU; ++u) if (c0t + u < NCT) gacc[GRAD ? m : 0][c0t + u] = mfma32(clo, bh[u], gacc[GRAD ? m : 0][c0t + u]);
                    }
                }
                if (g < 2) {                                   // Gamma_m = sum dL/dS_J * S_m, each pair once (anchor-owner sweep)
#pragma unroll
                    for (int m = 0; m < M; ++m)
#pragma unroll
                        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) gam[m] = fmaf(cj[jh][r], sacc[m][jh][r], gam[m]);
                }
#pragma unroll
                for (int m = 0; m < M; ++m) asm volatile("" : "+v"(gam[m]));
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
                if (d < SB_DP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = own0 + wave * 16 + 4 * g4 + r;
                        if (i < own_end) atomicAdd(dz + (size_t)i * SB_DP + d, gacc[GRAD ? m : 0][ct][r]);
                    }
                }
            }
        }
        if (g < 2) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(gam[m]);
                if (lane == 0 && v != 0.f) atomicAdd(a.gamma + M * (1 + my_slot()) + m, (double)v);
            }
        }
    }
}

int fill_b(BArgs& a, const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1, bool grad,
           int a_lo, int a_hi, const char* who) {
    if (M < 2 || M > 3) { sga_set_error("%s: M=%d (the bf16x3 sweeps are built for 2 or 3 modality tables)", who, M); return SGA_ERR_ARG; }
    if (a_lo < 0 || a_hi > A || a_lo > a_hi) { sga_set_error("%s: anchor shard [%d,%d) outside [0,%d]", who, a_lo, a_hi, A); return SGA_ERR_ARG; }
    a.M = M;
    for (int m = 0; m < M; ++m) { if (!Zb[m]) { sga_set_error("%s: null table", who); return SGA_ERR_ARG; } a.Zb[m] = static_cast<const unsigned char*>(Zb[m]); }
    a.beta = beta; a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1;
    const BLayout L = make_layout(A, J1, J2);
    const int ns = a_hi - a_lo;
    const int bx1 = 0, bx2 = L.nbA, bn1 = 2 * L.nbA, bn2 = 2 * L.nbA + L.nb1;
    const int ox1 = 0, ox2 = A, on1 = 2 * A, on2 = 2 * A + J1;
    const BSeg N1a{bn1, 0, L.nb1, on1, on1, on1 + J1, 0}, N2a{bn2, 0, L.nb2, on2, on2, on2 + J2, 1};
    const BSeg N2b{bn2, 0, L.nb2, on2, on2, on2 + J2, 2}, N1b{bn1, 0, L.nb1, on1, on1, on1 + J1, 3};
    int g = 0;
    auto add = [&](int own0, int nown, int own_old0, int own_blk0, BSeg s0, BSeg s1) {
        if (nown <= 0) return;
        BGroup& G = a.grp[g++];
        G.own0 = own0; G.nown = nown; G.own_old0 = own_old0; G.own_blk0 = own_blk0; G.nseg = 2; G.seg[0] = s0; G.seg[1] = s1; G.nsplit = 1; G.blk0 = 0;
    };
    add(ox1 + a_lo, ns, ox1, bx1, N1a, N2a);                       // s11, s12
    add(ox2 + a_lo, ns, ox2, bx2, N2b, N1b);                       // s22, s21
    if (grad) {
        const int jl = a_lo / 32, jh = (a_hi + 31) / 32;
        const BSeg X1f0{bx1, jl, jh, ox1, ox1 + a_lo, ox1 + a_hi, 0}, X2f3{bx2, jl, jh, ox2, ox2 + a_lo, ox2 + a_hi, 3};
        const BSeg X1f1{bx1, jl, jh, ox1, ox1 + a_lo, ox1 + a_hi, 1}, X2f2{bx2, jl, jh, ox2, ox2 + a_lo, ox2 + a_hi, 2};
        add(on1, J1, on1, bn1, X1f0, X2f3);
        add(on2, J2, on2, bn2, X1f1, X2f2);
    }
    a.ngroups = g;
    // uniform ~target-step work units (one 8-wave workgroup per CU at a time)
    int nwg = 0;
    for (int i = 0; i < g; ++i) {
        BGroup& G = a.grp[i];
        int steps = 0;
        for (int sg = 0; sg < G.nseg; ++sg) steps += G.seg[sg].jt_hi - G.seg[sg].jt_lo;
        // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2); the kernel maps workgroup -> (split, owner block)
        // so that the 32 co-resident workgroups of an XCD share `split`, i.e. they walk the SAME other tiles at the same time for
        // different owner blocks: a tile comes from HBM/MALL once per XCD and 31 more times from its L2.
        int nsp = (steps + 159) / 160;
        if (nsp > steps) nsp = steps;
        if (nsp < 1) nsp = 1;
        G.nsplit = nsp;
        G.blk0 = nwg;
        nwg += (((G.nown + SB_OWN - 1) / SB_OWN) * nsp + 7) / 8 * 8;
    }
    return -nwg;                                                    // negative: number of workgroups (0 is a valid "nothing to do")
}

template <int M, bool GRAD>
void launch_b(const BArgs& a, int nwg, hipStream_t s) {
    const size_t lds = (size_t)SB_NBUF * M * SB_BLOCK + ((GRAD && M == 3) ? (size_t)M * 3 * SB_THREADS * 16 : 0);   // + the owner rows' lo planes
    auto k = sweepb_kernel<M, GRAD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(SB_THREADS), lds, s, a);
}

}  // namespace

extern "C" size_t sga_loss_split_bytes(int A, int J1, int J2) {
    const BLayout L = make_layout(A, J1, J2);
    return (size_t)(2 * L.nbA + L.nb1 + L.nb2 + 1) * SB_BLOCK;      // + one block of slack: the K tail reads a few bytes past a row
}

extern "C" int sga_loss_split_tables(const float* Z, int A, int J1, int J2, void* Zb, void* stream) {
    SGA_CHECK_ARG(A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_split_tables: bad sizes");
    const BLayout L = make_layout(A, J1, J2);
    const int nb = 2 * L.nbA + L.nb1 + L.nb2;
    if (nb == 0) return SGA_OK;
    SGA_CHECK_ARG(Z && Zb, "sga_loss_split_tables: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(static_cast<unsigned char*>(Zb) + (size_t)nb * SB_BLOCK, 0, SB_BLOCK, s) != hipSuccess) { sga_set_error("sga_loss_split_tables: memset failed"); return SGA_ERR_HIP; }
    hipLaunchKernelGGL(split_tables_kernel, dim3(nb), dim3(256), 0, s, Z, A, J1, J2, static_cast<unsigned char*>(Zb));
    SGA_CHECK_LAUNCH("sga_loss_split_tables");
    return SGA_OK;
}

extern "C" int sga_loss_multi_sums_bf16x3(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                          double* sums, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Zb && beta && sums && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_sums_bf16x3: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc0 = zero_slots(sums, (M + 1) * 8, s, "sga_loss_multi_sums_bf16x3")) return rc0;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    BArgs a{};
    const int r = fill_b(a, Zb, M, beta, A, J1, J2, tau0, tau1, false, a_lo, a_hi, "sga_loss_multi_sums_bf16x3");
    if (r > 0) return r;
    a.sums = sums;
    if (M == 2) launch_b<2, false>(a, -r, s); else launch_b<3, false>(a, -r, s);
    fold_slots(sums, (M + 1) * 8, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_sums_bf16x3");
    return SGA_OK;
}

extern "C" int sga_loss_multi_grad_bf16x3(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                          const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Zb && beta && gs && dZ && gamma && A >= 0 && J1 >= 0 && J2 >= 0, "sga_loss_multi_grad_bf16x3: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rcz = zero_slots(gamma, M > 0 ? M : 1, s, "sga_loss_multi_grad_bf16x3")) return rcz;
    if (A == 0 || a_hi <= a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    BArgs a{};
    const int r = fill_b(a, Zb, M, beta, A, J1, J2, tau0, tau1, true, a_lo, a_hi, "sga_loss_multi_grad_bf16x3");
    if (r > 0) return r;
    a.gs = gs; a.gamma = gamma;
    for (int m = 0; m < M; ++m) { SGA_CHECK_ARG(dZ[m], "sga_loss_multi_grad_bf16x3: null dZ"); a.dZ[m] = dZ[m]; }
    if (M == 2) launch_b<2, true>(a, -r, s); else launch_b<3, true>(a, -r, s);
    fold_slots(gamma, M, s);
    SGA_CHECK_LAUNCH("sga_loss_multi_grad_bf16x3");
    return SGA_OK;
}
