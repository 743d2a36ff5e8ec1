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
// to whole blocks; a block holds four bf16 planes, 26 624 B contiguous:
//     R hi | R lo   [32 rows][104 cols]    row-major: the S product's operands (8 consecutive k per lane = one 16-byte read)
//     T hi | T lo   [104 cols][32 rows]    transposed: the gradient GEMM's B operand (8 consecutive "other" rows of a column)
// so a tile of "other" rows is ONE contiguous 26 KiB copy per table (26 global_load_lds DMA chunks), double-buffered in LDS
// (3 tables: 2 x 78 KiB = 156 KiB: one 8-wave workgroup per CU, 128 owner rows).
// MFMA bookkeeping (v_mfma_f32_16x16x32_bf16; A: lane&15 = row, B: lane&15 = column, lane>>4 = k group of 8 slots):
//   S^T tile: A = other rows from LDS, B = owner rows (registers).  Half jh of a 32-row tile uses A row i <-> other row
//   8 (i>>2) + 4 jh + (i&3), so that a lane's 8 accumulator values (2 halves x 4) are the 8 CONSECUTIVE other rows 8 g4 .. 8 g4+7:
//   exactly the k slots of the gradient MFMA  dZ[own] += C[own, other] Z[other, cols]  whose A operand is therefore the
//   coefficient registers (split into bf16 hi/lo, 6 VALU per pair) and whose B operand is one ds_read_b128 of the T plane.
#include "loss_math.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int SB_WAVES = 8, SB_THREADS = SB_WAVES * 64, SB_OWN = SB_WAVES * 16;
constexpr int SB_DP = 104;
constexpr int SB_ROWB = SB_DP * 2;               // bytes of a row in an R plane
constexpr int SB_PLANE = 32 * SB_ROWB;           // 6656 B
constexpr int SB_BLOCK = 4 * SB_PLANE;           // 26624 B: R hi | R lo | T hi | T lo

__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    const bf16x2 h = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const float b0 = __builtin_bit_cast(float, hi << 16), b1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0 - b0, v1 - b1}, bf16x2));
}
__device__ __forceinline__ f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
    for (int e = threadIdx.x; e < 32 * 52; e += 256) {             // R planes: dword (row w, column pair p)
        const int w = e / 52, p = e - w * 52;
        unsigned hi, lo;
        split_pair(tile[w * SB_DP + 2 * p], tile[w * SB_DP + 2 * p + 1], hi, lo);
        out[e] = hi; out[PD + e] = lo;
    }
    for (int e = threadIdx.x; e < SB_DP * 16; e += 256) {          // T planes: dword (column c, row pair q)
        const int c = e >> 4, q = e & 15;
        unsigned hi, lo;
        split_pair(tile[(2 * q) * SB_DP + c], tile[(2 * q + 1) * SB_DP + c], hi, lo);
        out[2 * PD + e] = hi; out[3 * PD + e] = lo;
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
__global__ __launch_bounds__(SB_THREADS, 1) void sweepb_kernel(BArgs a) {
    constexpr int NCT = 7;
    constexpr int TBYTES = GRAD ? SB_BLOCK : 2 * SB_PLANE;          // the sums pass only needs the R planes
    constexpr int BUF = M * TBYTES, NCH = TBYTES / 1024, NCHUNK = M * NCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];      // [2][M][TBYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < a.ngroups && (int)blockIdx.x >= a.grp[i].blk0) g = i;
    const BGroup& grp = a.grp[g];
    const int wg_in_grp = (int)blockIdx.x - grp.blk0;
    const int nsplit = grp.nsplit, split = wg_in_grp % nsplit;
    const int own0 = grp.own0 + (wg_in_grp / nsplit) * SB_OWN;
    const int own_end = grp.own0 + grp.nown;
    const int my_i = own0 + wave * 16 + l15;
    const bool iv = my_i < own_end;

    // ---- owner rows as the S product's B operand: 3 K = 32 steps (8 columns per lane) + the K = 16 tail (columns 96 + 4 g4 .. +3)
    u32x4 ohi[M][3], olo[M][3];
    u32x2 othi[M], otlo[M];
    float beta[M];
    {
        const int rel = (iv ? my_i : own0) - grp.own_old0;
        const size_t off = (size_t)(grp.own_blk0 + (rel >> 5)) * SB_BLOCK + (size_t)(rel & 31) * SB_ROWB;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned char* base = a.Zb[m] + off;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ohi[m][q] = *reinterpret_cast<const u32x4*>(base + (32 * q + 8 * g4) * 2);
                olo[m][q] = *reinterpret_cast<const u32x4*>(base + SB_PLANE + (32 * q + 8 * g4) * 2);
                if (!iv) { ohi[m][q] = u32x4{0, 0, 0, 0}; olo[m][q] = u32x4{0, 0, 0, 0}; }
            }
            const bool tv = iv && g4 < 2;                      // columns 96..103 only: the k slots of lanes 32..63 multiply zeros
            othi[m] = tv ? *reinterpret_cast<const u32x2*>(base + (96 + 4 * g4) * 2) : u32x2{0, 0};
            otlo[m] = tv ? *reinterpret_cast<const u32x2*>(base + SB_PLANE + (96 + 4 * g4) * 2) : u32x2{0, 0};
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

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // M0 (the DMA's LDS address) must be provably uniform
    auto issue = [&](int blk, unsigned char* buf) {
#pragma unroll
        for (int c0 = 0; c0 < NCHUNK; c0 += SB_WAVES) {
            const int c = c0 + wave_u;
            if (c >= NCHUNK) break;
            const int m = c / NCH, cc = c - m * NCH;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(a.Zb[m] + (size_t)blk * SB_BLOCK + cc * 1024 + lane * 16),
                (__attribute__((address_space(3))) void*)(buf + m * TBYTES + cc * 1024), 16, 0, 0);
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
        if (seg.jt_lo + split < seg.jt_hi) issue(seg.blk0 + seg.jt_lo + split, ldsb);
        int it = 0;
#pragma unroll 1
        for (int jt = seg.jt_lo + split; jt < seg.jt_hi; jt += nsplit, ++it) {
            unsigned char* buf = ldsb + (it & 1) * BUF;
            const int j0 = seg.old0 + 32 * jt;                 // old row of the tile's first row
            __syncthreads();                                   // tile `it` landed / other buffer free
            if (jt + nsplit < seg.jt_hi) issue(seg.blk0 + jt + nsplit, ldsb + ((it + 1) & 1) * BUF);

            // ---- S^T tiles: sacc[m][jh][r] = S_m[own = lane&15, other = 8 g4 + 4 jh + r]
            f32x4 sacc[M][2];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const int rowi = 8 * (l15 >> 2) + 4 * jh + (l15 & 3);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const unsigned char* ar = buf + m * TBYTES + rowi * SB_ROWB;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const u32x4 ah = *reinterpret_cast<const u32x4*>(ar + (32 * q + 8 * g4) * 2);
                        const u32x4 al = *reinterpret_cast<const u32x4*>(ar + SB_PLANE + (32 * q + 8 * g4) * 2);
                        acc = mfma32(ah, ohi[m][q], acc);
                        acc = mfma32(ah, olo[m][q], acc);
                        acc = mfma32(al, ohi[m][q], acc);
                    }
                    const u32x2 th = *reinterpret_cast<const u32x2*>(ar + (96 + 4 * g4) * 2);
                    const u32x2 tl = *reinterpret_cast<const u32x2*>(ar + SB_PLANE + (96 + 4 * g4) * 2);
                    acc = mfma16(th, othi[m], acc);
                    acc = mfma16(th, otlo[m], acc);
                    acc = mfma16(tl, othi[m], acc);
                    sacc[m][jh] = acc;
                }
            }

            float okf[2][4];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = j0 + 8 * g4 + 4 * jh + r;
                    okf[jh][r] = (iv && row >= seg.lo && row < seg.hi) ? 1.f : 0.f;
                }
            if (!GRAD) {
                float p0[M + 1], p1[M + 1];
#pragma unroll
                for (int m = 0; m <= M; ++m) { p0[m] = 0.f; p1[m] = 0.f; }
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sj = 0.f;
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            const float sv = sacc[m][jh][r];
                            sj = fmaf(beta[m], sv, sj);
                            p0[m] = fmaf(okf[jh][r], fexp2(sv * a.k0), p0[m]);
                            p1[m] = fmaf(okf[jh][r], fexp2(sv * a.k1), p1[m]);
                        }
                        p0[M] = fmaf(okf[jh][r], fexp2(sj * a.k0), p0[M]);
                        p1[M] = fmaf(okf[jh][r], fexp2(sj * a.k1), p1[M]);
                    }
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
                        cj[jh][r] = okf[jh][r] * (c0[M] * fexp2(sj * a.k0) + c1[M] * fexp2(sj * a.k1));
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
                for (int m = 0; m < M; ++m) {
                    // c_m for this lane's 8 consecutive other rows (k slot j = 4 jh + r), split into bf16 hi / lo: the A operand
                    u32x4 chi, clo;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int jh = p >> 1, r = (p & 1) * 2;
                        const float sv0 = sacc[m][jh][r], sv1 = sacc[m][jh][r + 1];
                        const float v0 = okf[jh][r] * fmaf(beta[m], cj[jh][r], c0[m] * fexp2(sv0 * a.k0) + c1[m] * fexp2(sv0 * a.k1));
                        const float v1 = okf[jh][r + 1] * fmaf(beta[m], cj[jh][r + 1], c0[m] * fexp2(sv1 * a.k0) + c1[m] * fexp2(sv1 * a.k1));
                        unsigned hi, lo;
                        split_pair(v0, v1, hi, lo);
                        chi[p] = hi; clo[p] = lo;
                    }
                    const unsigned char* tb = buf + m * TBYTES + 2 * SB_PLANE + l15 * 64 + g4 * 16;   // T plane: [column][32 others]
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const u32x4 bh = *reinterpret_cast<const u32x4*>(tb + ct * 16 * 64);
                        const u32x4 bl = *reinterpret_cast<const u32x4*>(tb + SB_PLANE + ct * 16 * 64);
                        f32x4 acc = gacc[GRAD ? m : 0][ct];
                        acc = mfma32(chi, bh, acc);
                        acc = mfma32(chi, bl, acc);
                        acc = mfma32(clo, bh, acc);
                        gacc[GRAD ? m : 0][ct] = acc;
                    }
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
        int nsp = (steps + 159) / 160;
        if (nsp < 1) nsp = 1;
        G.nsplit = nsp;
        G.blk0 = nwg;
        nwg += ((G.nown + SB_OWN - 1) / SB_OWN) * nsp;
    }
    return -nwg;                                                    // negative: number of workgroups (0 is a valid "nothing to do")
}

template <int M, bool GRAD>
void launch_b(const BArgs& a, int nwg, hipStream_t s) {
    const size_t lds = (size_t)2 * M * (GRAD ? SB_BLOCK : 2 * SB_PLANE);
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
