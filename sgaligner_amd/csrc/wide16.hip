// fp16-input / fp32-accumulate path of the batch-global loss on WIDE tables (BASELINE.json configs[4]: 1024-d embeddings, "MFMA
// similarity GEMM at fp16").  Opt-in (ops.set_mfma_mode('f16')); exact fp32 stays the default everywhere.
//
// Replaces, for packed tables wider than 128 columns, the fp32 sweeps of contrastive.hip:
//   sga_loss_neg_sums      -> sga_loss_neg_sums_f16       the 4x2 global sums  sum_ij exp(S_ij / tau)              (losses.py:6-11)
//   sga_loss_neg_grad_wide -> sga_loss_neg_grad_f16       dL/dZ through the anchors x negatives similarities       (autograd of :6-11)
// with S = Z Z^T and both gradient GEMMs on v_mfma_f32_32x32x16_f16.  One tiled NT-GEMM core (out[m][n] = sum_k A[m][k] B[n][k],
// both operands fp16 with k contiguous, 256 x 256 tile per workgroup of 8 waves, K chunks of 64 double-buffered in LDS by LDS-DMA -- see
// wide16_body) carries three epilogues:
//   SUMS  exp2 of the S tile, masked, reduced to the fp64 per-wave slots;
//   COEF  c_ij = (g0/tau0 exp(S/tau0) + g1/tau1 exp(S/tau1)) / alpha  written as fp16 (alpha = the coefficient's bound, so |c/alpha| <= 1:
//         inside fp16's range whatever the loss scale; values below 6e-8 of the bound flush to zero);
//   GEMM  out += alpha * acc  (fp32 atomics; split over K for occupancy);
//   STORE out  = acc          (fp32; the anchors x anchors similarity blocks of contrastive.hip's epilogue-only kernels, wide16_api.h).
// Operand layouts (sga_wide16_prepare): Zh [R][Dp] = fp16 copy of the packed normalised table (S operand), ZhT [Dp][ldt] = its
// transpose with every segment (X1 | X2 | N1 | N2) starting at a multiple of 8 columns (the gradient GEMMs' B operand: k = packed row).
// The gradient GEMMs need the coefficient tile in both orientations (anchor-major C for dZ[anchors] = C Z[neg], negative-major C^T for
// dZ[neg] = C^T Z[anchors]); ONE S pass writes both: in the 32 x 32 accumulator layout a lane's registers 4 t .. 4 t + 3 are four consecutive
// rows, i.e. one 8-byte store into C^T (until round 6 S was formed twice, once per orientation).
#include "loss_math.h"
#include "wide16_api.h"

namespace {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int W_THREADS = 512, W_T = 256, W_KC = 64;
constexpr int W_OP = W_T * W_KC * 2;        // bytes of one operand's K chunk in LDS: 256 rows x 128 B (32 KiB)
constexpr int W_STAGE = 2 * W_OP;           // A rows | B rows
constexpr int W_LDS = 2 * W_STAGE;          // two stages: 128 KiB, one workgroup (8 waves, two per SIMD) per CU
constexpr int W_SUMS = 0, W_COEF = 1, W_GEMM = 2, W_STORE = 3;

struct W16Args {
    const f16* A; long lda; int M;          // A operand rows [M][K]  (MFMA rows: r / lane>>5)
    const f16* B; long ldb; int N;          // B operand rows [N][K]  (MFMA columns: lane & 31)
    int K, kper;                            // contraction length, K range per blockIdx.z (multiple of W_KC)
    float k0, k1, it0, it1;                 // log2(e)/tau, 1/tau
    const double* gs; int fam;              // dL/d(sums) [8] (COEF / GEMM: the block's coefficient scale), sum family 0..3
    double* sums;                           // SUMS out (slots)
    f16* cout; long ldc;                    // COEF out  C [M][ldc]; columns N .. ldc - 1 are written as zeros (the GEMMs read whole 8-half groups)
    f16* cout2; long ldc2;                  // COEF out  C^T [N][ldc2] (ldc2 = M padded to 8), same zero padding
    float* out; long ldo;                   // GEMM out [M][ldo], atomically accumulated (STORE: plainly stored)
    const float* alpha_dev;                 // GEMM: out += *alpha_dev * acc when set (a scale another kernel left in device memory), else the bound from gs
};

// 16 readable zero bytes: the source of every 8-half group that lies past the K range
__device__ __attribute__((aligned(16))) const unsigned g_w16_zero[4] = {0u, 0u, 0u, 0u};

// One LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_addr + 16 lane] (lds_addr wave-uniform, through M0).  Inline asm
// on purpose: hipcc does not count these loads, so IT never drains them (beside the builtin it waits vmcnt(0) in front of the next LDS
// read, i.e. for the chunk that was requested a moment ago); the one wait per K chunk is written out below.
__device__ __forceinline__ void glds16(const void* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory");
}

// the coefficient bound of a sum family: |g0|/tau0 * 2^k0 + |g1|/tau1 * 2^k1  (S <= 1 for normalised rows; +2 % for fp16 rounding of S)
__device__ __forceinline__ float coef_bound(const double* gs, int fam, float k0, float k1, float it0, float it1) {
    const float c0 = fabsf((float)(gs[fam * 2 + 0] * (double)it0)), c1 = fabsf((float)(gs[fam * 2 + 1] * (double)it1));
    return fmaxf(1.02f * (c0 * fexp2(k0) + c1 * fexp2(k1)), 1e-30f);
}

// The tile core: out[m][n] = sum_k A[m][k] B[n][k] on a 256 x 256 tile, 8 waves as 2 (m) x 4 (n), a wave owns 128 x 64 = 4 x 2 accumulator
// tiles of v_mfma_f32_32x32x16_f16 (128 registers; 6 ds_read_b128 per 8 MFMAs).  K in chunks of 64: both operands' 256 rows x 128 B go
// global -> LDS by LDS-DMA (no staging registers, no ds_write pass), double-buffered: the chunk after this one is requested right behind
// the barrier that frees its buffer and lands under this chunk's 32 MFMAs per wave; one wait + one barrier per chunk.
// LDS image of an operand chunk: row r at 128 r, its eight 16-byte k groups XOR-swizzled, group g at slot g ^ ((r >> 1) & 7): the 16 lanes
// ds_read_b128 serves per cycle (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a 32-row operand tile, one k group) then hit 16 different
// 16-byte slots of the 256-byte bank row.  The DMA writes lane-linearly, so the swizzle is in the SOURCE address: lane l of copy i
// (rows 8 i .. 8 i + 7) fetches row 8 i + (l >> 3), group (l & 7) ^ swizzle(row).
// Edges: rows past M / N re-read the last row (their results are masked in the epilogues); 8-half groups that START past the K range read
// the zero block; a group that straddles K (GEMM mode: K = a packed row count) reads the operands' zero padding up to the next multiple of 8.
// Tile order (XCD-aware): a launch's tile grid (gx column tiles x gy row tiles per z slice) is a 1-D grid of T8 = gx gy rounded up to 8
// workgroups; the hardware deals consecutive workgroup ids to the 8 XCDs in turn, so XCD c = id % 8 gets the CONTIGUOUS range
// [c T8 / 8, (c + 1) T8 / 8) of tile positions, and positions walk 8 row tiles down before they step to the next column tile: the ~32
// workgroups an XCD runs at a time form an 8 x 4 patch that shares 12 operand panels through its L2 (a 256 x 256 x 64 tile step has 128
// FLOP per operand byte: without that reuse 1 PFLOP/s would ask HBM for 8 TB/s).
template <int MODE>
__device__ __forceinline__ void wide16_body(const W16Args& a, int bz, int gx, int gy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;
    const int pos = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    const int grp = pos / (8 * gx), q = pos - grp * 8 * gx;
    const int gm = min(8, gy - 8 * grp);                     // row tiles of this group (the last group may be shorter)
    if (pos >= gx * gy) return;
    const int tile_m = 8 * grp + q % gm, tile_n = q / gm;
    const int m0 = tile_m * W_T, n0 = tile_n * W_T;
    if (m0 >= a.M || n0 >= a.N) return;                      // (batched launches: the grid is the largest family's)
    const int kb = bz * a.kper, ke = min(a.K, kb + a.kper);
    if (kb >= ke) return;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: copy i = 8 wave + q of a chunk's 64 (0..31: A rows 8 i .., 32..63: B rows); waves 0..3 carry A, 4..7 carry B
    const f16* rp[8];
    int kg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = wave * 8 + q, ii = i & 31, row = 8 * ii + (lane >> 3);
        const int g = (lane & 7) ^ ((row >> 1) & 7);
        kg[q] = 8 * g;
        if (i < 32) rp[q] = a.A + (size_t)min(m0 + row, a.M - 1) * a.lda + 8 * g;
        else rp[q] = a.B + (size_t)min(n0 + row, a.N - 1) * a.ldb + 8 * g;
    }
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds16) + (unsigned)wave * 8192u;
    const f16* zero = reinterpret_cast<const f16*>(g_w16_zero);
    auto stage = [&](int buf, int k, int q_lo, int q_hi) {
#pragma unroll
        for (int q = q_lo; q < q_hi; ++q) glds16(k + kg[q] < ke ? rp[q] + k : zero, lbase + (unsigned)(buf * W_STAGE + q * 1024));
    };
    const int swz = (l31 >> 1) & 7;
    stage(0, kb, 0, 8);
    int it = 0;
    for (int k = kb; k < ke; k += W_KC, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's copies of chunk `it` have landed ...
        __syncthreads();                                     // ... everybody's have, and everybody has read chunk it - 1: its buffer is free
        const bool more = k + W_KC < ke;
        const unsigned char* sa = lds16 + (it & 1) * W_STAGE + (wm * 128 + l31) * 128;
        const unsigned char* sb = lds16 + (it & 1) * W_STAGE + W_OP + (wn * 64 + l31) * 128;
#pragma unroll
        for (int kk = 0; kk < W_KC / 16; ++kk) {
            const int co = ((2 * kk + h) ^ swz) << 4;
            f16x8 av[4], bv[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const f16x8*>(sa + i * 4096 + co);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const f16x8*>(sb + j * 4096 + co);
            // the next chunk's copies ride in the shadow of the first two steps' LDS reads, four each (all eight in a burst behind the
            // barrier queue up at the CU's one address unit -- 16 cycles per 1-KiB copy, 64 per chunk -- before any wave reaches an MFMA)
            if (kk < 2 && more) {
                __builtin_amdgcn_sched_barrier(0);
                stage((it + 1) & 1, k + W_KC, 4 * kk, 4 * kk + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: element (i, j, r) of this lane <-> A row m = m0 + 128 wm + 32 i + mfma32_row(r, h), B row n = n0 + 64 wn + 32 j + l31
    const int mw = m0 + wm * 128, nw = n0 + wn * 64 + l31;
    if (MODE == W_SUMS) {
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw + i * 32 + mfma32_row(r, h), n = nw + j * 32;
                    const float ok = (m < a.M && n < a.N) ? 1.f : 0.f;
                    p0 = fmaf(ok, fexp2(acc[i][j][r] * a.k0), p0);
                    p1 = fmaf(ok, fexp2(acc[i][j][r] * a.k1), p1);
                }
        const double v0 = wave_sum_d((double)p0), v1 = wave_sum_d((double)p1);
        if (lane == 0) {
            const int slot = (int)(((unsigned)blockIdx.x * 8u + (unsigned)wave) % SGA_SLOTS);
            atomicAdd(a.sums + 8 * (1 + slot) + a.fam * 2 + 0, v0);
            atomicAdd(a.sums + 8 * (1 + slot) + a.fam * 2 + 1, v1);
        }
    } else if (MODE == W_COEF) {
        // Both orientations from the one S tile: C [m][n] (lanes along n: 64-byte runs of 2-byte stores) and C^T [n][m] -- a lane's four
        // registers r = 4 t .. 4 t + 3 are four CONSECUTIVE rows m, i.e. one 8-byte store into row n of C^T.
        const float inv = 1.f / coef_bound(a.gs, a.fam, a.k0, a.k1, a.it0, a.it1);
        const float c0 = (float)(a.gs[a.fam * 2 + 0] * (double)a.it0) * inv, c1 = (float)(a.gs[a.fam * 2 + 1] * (double)a.it1) * inv;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nw + j * 32;
            const float keep = n < a.N ? 1.f : 0.f;             // a row's padding up to ldc: zeros
            f16* const cb = a.cout + (size_t)(mw + 4 * h) * a.ldc + n;
            f16* const ctb = a.cout2 + (size_t)n * a.ldc2 + mw + 4 * h;
            const int lc = (int)a.ldc;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int mq = mw + i * 32 + 8 * t + 4 * h;  // = mfma32_row(4 t, h): rows mq .. mq + 3
                    f16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[i][j][4 * t + e];
                        v[e] = (f16)((mq + e < a.M ? keep : 0.f) * (c0 * fexp2(x * a.k0) + c1 * fexp2(x * a.k1)));
                    }
                    // (one 64-bit pointer per (i, j) block and lane, rows by 32-bit offsets: a 64-bit multiply per element costs more than its store)
                    if (n < a.ldc) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (mq + e < a.M) cb[(i * 32 + 8 * t + e) * lc] = v[e];
                    }
                    if (n < a.N && mq < a.ldc2) *reinterpret_cast<f16x4*>(ctb + i * 32 + 8 * t) = v;
                }
        }
    } else if (MODE == W_STORE) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nw + j * 32;
            if (n >= a.N) continue;
            float* const ob = a.out + (size_t)(mw + 4 * h) * a.ldo + n;
            const int lo_ = (int)a.ldo;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (mw + 4 * h + ro < a.M) ob[ro * lo_] = acc[i][j][r];
                }
        }
    } else {
        const float alpha = a.alpha_dev ? *a.alpha_dev : coef_bound(a.gs, a.fam, a.k0, a.k1, a.it0, a.it1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nw + j * 32;
            if (n >= a.N) continue;
            float* const ob = a.out + (size_t)(mw + 4 * h) * a.ldo + n;
            const int lo_ = (int)a.ldo;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
#ifdef W16_DBG_NOATOMIC
                    if (mw + 4 * h + ro < a.M) ob[ro * lo_] = alpha * acc[i][j][r];       // (timing experiment: wrong results)
#else
                    if (mw + 4 * h + ro < a.M) atomicAdd(ob + ro * lo_, alpha * acc[i][j][r]);
#endif
                }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(W_THREADS, 1) void wide16_kernel(W16Args a, int gx, int gy) { wide16_body<MODE>(a, blockIdx.z, gx, gy); }

// Up to eight products -- the four sum families of a table -- in ONE launch (blockIdx.z = product * zper + K split): 16 launches of ~85 us per table and backward at
// configs[4] -- grids of a few hundred tiles each -- become 4.
struct W16Batch { W16Args a[8]; int n, zper, gx, gy; };
template <int MODE>
__global__ __launch_bounds__(W_THREADS, 1) void wide16_batch_kernel(W16Batch b) {
    const int q = blockIdx.z / b.zper;
    if (q >= b.n) return;
    wide16_body<MODE>(b.a[q], blockIdx.z - q * b.zper, b.gx, b.gy);
}

// Z fp32 [R][Dp] -> Zh fp16 [R][Dp] and ZhT fp16 [Dp][ldt] (column of packed row r: r + shift of its segment)
struct PrepArgs { const float* Z; int R, Dp; f16* Zh; f16* ZhT; long ldt; int seg_end[4]; int shift[4]; };
__global__ __launch_bounds__(256) void wide16_prepare_kernel(PrepArgs a) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, d0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, d = d0 + tx;
        float v = 0.f;
        if (r < a.R && d < a.Dp) {
            v = a.Z[(size_t)r * a.Dp + d];
            a.Zh[(size_t)r * a.Dp + d] = (f16)v;
        }
        tile[ty * 16 + i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = d0 + ty * 16 + i, r = r0 + tx;
        if (r < a.R && d < a.Dp) {
            const int sg = (r >= a.seg_end[0]) + (r >= a.seg_end[1]) + (r >= a.seg_end[2]);
            a.ZhT[(size_t)d * a.ldt + r + a.shift[sg]] = (f16)tile[tx][ty * 16 + i];
        }
    }
}

inline int pad8(int n) { return (n + 7) / 8 * 8; }
// padded column of the first row of each segment of ZhT
struct SegCols { int x1, x2, n1, n2, ldt; };
inline SegCols seg_cols(int A, int J1, int J2) {
    SegCols s;
    s.x1 = 0; s.x2 = pad8(A); s.n1 = s.x2 + pad8(A); s.n2 = s.n1 + pad8(J1); s.ldt = s.n2 + pad8(J2);
    if (s.ldt < 8) s.ldt = 8;
    return s;
}

// K splits of a gradient GEMM whose launch holds `tiles` output tiles (all products of the launch).  One 128-KiB workgroup runs per CU at a
// time, so the launch proceeds in rounds of `ncu` workgroups, and a workgroup costs its K chunks plus a constant: prologue, and the fp32
// atomics of its 256 x 256 tile, which the memory side executes as read-modify-writes of whole lines -- measured ~27 us per workgroup, the
// time of ~20 K chunks (tools/bench_wide16.py with -DW16_DBG_NOATOMIC).  Take the split count that minimises rounds x (chunks + 26).
inline int choose_ksplit(int tiles, int K, int ncu, int& kper) {
    const int kmax = max(1, min(16, (K + 511) / 512));
    int best = 1;
    long best_cost = -1;
    for (int ks = 1; ks <= kmax; ++ks) {
        const long wgs = (long)tiles * ks;
        const long rounds = (wgs + ncu - 1) / ncu;
        const long chunks = ((K + ks - 1) / ks + W_KC - 1) / W_KC;
        const long cost = rounds * (chunks + 26);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ks; }
    }
    kper = ((K + best - 1) / best + W_KC - 1) / W_KC * W_KC;
    return (K + kper - 1) / kper;
}

inline int tiles8(int gx, int gy) { return (gx * gy + 7) / 8 * 8; }
// the epilogues address a 128-row block of C / out by 32-bit element offsets from one row pointer: pitch x 128 must stay below 2^31
inline bool pitches_ok(const W16Args& a) { return a.ldc < (1L << 24) && a.ldc2 < (1L << 24) && a.ldo < (1L << 24); }
template <int MODE>
void launch(const W16Args& a, int ksplit, hipStream_t s) {
    const int gx = (a.N + W_T - 1) / W_T, gy = (a.M + W_T - 1) / W_T;
    if (!pitches_ok(a)) { sga_set_error("wide16: a row pitch of 2^24 elements or more"); return; }
    hipFuncSetAttribute(reinterpret_cast<const void*>(wide16_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    hipLaunchKernelGGL(wide16_kernel<MODE>, dim3(tiles8(gx, gy), 1, ksplit), dim3(W_THREADS), W_LDS, s, a, gx, gy);
}
template <int MODE>
void launch_batch(W16Batch& b, const int* ksplit, hipStream_t s) {
    int gx = 1, gy = 1, kz = 1;
    for (int q = 0; q < b.n; ++q) {
        gx = max(gx, (b.a[q].N + W_T - 1) / W_T); gy = max(gy, (b.a[q].M + W_T - 1) / W_T); kz = max(kz, ksplit[q]);
    }
    for (int q = 0; q < b.n; ++q) if (!pitches_ok(b.a[q])) { sga_set_error("wide16: a row pitch of 2^24 elements or more"); return; }
    b.zper = kz; b.gx = gx; b.gy = gy;
    hipFuncSetAttribute(reinterpret_cast<const void*>(wide16_batch_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    hipLaunchKernelGGL(wide16_batch_kernel<MODE>, dim3(tiles8(gx, gy), 1, kz * b.n), dim3(W_THREADS), W_LDS, s, b);
}

}  // namespace

extern "C" long sga_wide16_ldt(int A, int J1, int J2) { return seg_cols(A, J1, J2).ldt; }

extern "C" int sga_wide16_prepare(const float* Z, int Dp, int A, int J1, int J2, void* Zh, void* ZhT, void* stream) {
    SGA_CHECK_ARG(Dp > 0 && Dp % 8 == 0 && A >= 0 && J1 >= 0 && J2 >= 0, "sga_wide16_prepare: bad sizes (Dp must be a multiple of 8)");
    const int R = 2 * A + J1 + J2;
    if (R == 0) return SGA_OK;
    SGA_CHECK_ARG(Z && Zh && ZhT, "sga_wide16_prepare: null pointer");
    const SegCols sc = seg_cols(A, J1, J2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(ZhT, 0, (size_t)Dp * sc.ldt * sizeof(f16), s) != hipSuccess) { sga_set_error("sga_wide16_prepare: memset failed"); return SGA_ERR_HIP; }
    PrepArgs p{};
    p.Z = Z; p.R = R; p.Dp = Dp; p.Zh = static_cast<f16*>(Zh); p.ZhT = static_cast<f16*>(ZhT); p.ldt = sc.ldt;
    p.seg_end[0] = A; p.seg_end[1] = 2 * A; p.seg_end[2] = 2 * A + J1; p.seg_end[3] = R;
    p.shift[0] = sc.x1 - 0; p.shift[1] = sc.x2 - A; p.shift[2] = sc.n1 - 2 * A; p.shift[3] = sc.n2 - (2 * A + J1);
    hipLaunchKernelGGL(wide16_prepare_kernel, dim3((R + 63) / 64, (Dp + 63) / 64), dim3(256), 0, s, p);
    SGA_CHECK_LAUNCH("sga_wide16_prepare");
    return SGA_OK;
}

namespace {
struct Blk { int own_row, n_own, oth_row, n_oth, fam; int own_col, oth_col; };     // packed rows of Zh / padded columns of ZhT
inline void four_blocks(Blk (&b)[4], int A, int J1, int J2) {
    const SegCols sc = seg_cols(A, J1, J2);
    const int x1 = 0, x2 = A, n1 = 2 * A, n2 = 2 * A + J1;
    b[0] = Blk{x1, A, n1, J1, 0, sc.x1, sc.n1};       // s11   (losses.py:7 with e1i, e1j)
    b[1] = Blk{x1, A, n2, J2, 1, sc.x1, sc.n2};       // s12   (:8   e1i, e2j)
    b[2] = Blk{x2, A, n2, J2, 2, sc.x2, sc.n2};       // s22   (second call of calculate_prob_dist: e2i, e2j)
    b[3] = Blk{x2, A, n1, J1, 3, sc.x2, sc.n1};       // s21
}
}  // namespace

extern "C" int sga_loss_neg_sums_f16(const void* Zh, int Dp, int A, int J1, int J2, float tau0, float tau1, double* sums8, int a_lo, int a_hi,
                                     void* stream) {
    SGA_CHECK_ARG(Zh && sums8 && Dp % 8 == 0 && A >= 0 && J1 >= 0 && J2 >= 0 && tau0 > 0 && tau1 > 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi,
                  "sga_loss_neg_sums_f16: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = zero_slots(sums8, 8, s, "sga_loss_neg_sums_f16")) return rc;
    if (a_hi == a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    Blk b[4];
    four_blocks(b, A, J1, J2);
    const f16* Z = static_cast<const f16*>(Zh);
    W16Batch bs{};
    int ones[4] = {1, 1, 1, 1};
    for (int q = 0; q < 4; ++q) {
        if (b[q].n_own == 0 || b[q].n_oth == 0) continue;
        W16Args a{};
        a.A = Z + (size_t)(b[q].own_row + a_lo) * Dp; a.lda = Dp; a.M = a_hi - a_lo;      // this process's anchor shard against all negatives
        a.B = Z + (size_t)b[q].oth_row * Dp; a.ldb = Dp; a.N = b[q].n_oth;
        a.K = Dp; a.kper = (Dp + W_KC - 1) / W_KC * W_KC;
        a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.fam = b[q].fam; a.sums = sums8;
        bs.a[bs.n++] = a;
    }
    if (bs.n) launch_batch<W_SUMS>(bs, ones, s);               // the four families in one launch
    fold_slots(sums8, 8, s);
    SGA_CHECK_LAUNCH("sga_loss_neg_sums_f16");
    return SGA_OK;
}

// bytes of stash for anchor-row blocks of `rows` anchors (one (X, N) block at a time: C [rows][pad8(J)] + C^T [J][pad8(rows)], fp16)
extern "C" size_t sga_loss_neg_grad_f16_bytes(int A, int J1, int J2) {
    const int J = J1 > J2 ? J1 : J2;
    return 4 * (sizeof(f16) * ((size_t)A * pad8(J) + (size_t)J * pad8(A)) + 256);     // four stash pairs: the four families share their launches
}

extern "C" int sga_loss_neg_grad_f16(const void* Zh, const void* ZhT, int Dp, int A, int J1, int J2, float tau0, float tau1,
                                     const double* gs8, float* dZ, void* stash, size_t stash_bytes, int a_lo, int a_hi, void* stream) {
    SGA_CHECK_ARG(Zh && ZhT && gs8 && dZ && stash && Dp % 8 == 0 && A >= 0 && J1 >= 0 && J2 >= 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi &&
                  (a_lo % 8 == 0 || a_hi == a_lo), "sga_loss_neg_grad_f16: bad argument (a_lo must be a multiple of 8)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a_hi == a_lo || (J1 == 0 && J2 == 0)) return SGA_OK;
    const int Jmax = J1 > J2 ? J1 : J2;
    // anchor rows per pass so that C + C^T of one block fit the workspace
    auto need = [&](size_t r) { return sizeof(f16) * (r * pad8(Jmax) + (size_t)Jmax * pad8((int)r)) + 256; };
    size_t rows = a_hi - a_lo;
    // prefer the largest block of which FOUR stash pairs fit (the four sum families then share every launch); only if not even 128 rows do,
    // fall back to the largest block with one pair (family-by-family launches)
    while (rows > 128 && 4 * need(rows) > stash_bytes) rows = (rows / 2 + 127) / 128 * 128;
    if (4 * need(rows) > stash_bytes) {
        rows = a_hi - a_lo;
        while (rows > 128 && need(rows) > stash_bytes) rows = (rows / 2 + 127) / 128 * 128;
    }
    SGA_CHECK_ARG(need(rows) <= stash_bytes, "sga_loss_neg_grad_f16: workspace of %zu bytes holds fewer than %zu anchor rows for J = %d",
                  stash_bytes, rows, Jmax);
    const SegCols sc = seg_cols(A, J1, J2);
    const f16* Z = static_cast<const f16*>(Zh);
    const f16* ZT = static_cast<const f16*>(ZhT);
    Blk b[4];
    four_blocks(b, A, J1, J2);
    const int ncu = sga_num_cus();
    auto ksplit_for = [&](int M, int N, int K, int nfam, int& kper) {
        return choose_ksplit(((M + W_T - 1) / W_T) * ((N + W_T - 1) / W_T) * nfam, K, ncu, kper);
    };
    // one stash pair per family if the workspace holds four (then the four families share every launch), else family by family
    const bool batched = 4 * need(rows) <= stash_bytes;
    int nfam = 0;
    for (int q = 0; q < 4; ++q) nfam += b[q].n_oth > 0;
    if (!batched || nfam < 1) nfam = 1;
    for (int lo = a_lo; lo < a_hi; lo += (int)rows) {                 // (blocks of a multiple of 128 rows from a_lo: every `lo` stays a multiple of 8)
        const int ns = (lo + (int)rows < a_hi ? (int)rows : a_hi - lo);
        W16Batch bc{}, bg1{}, bg2{};
        int k1s[4], k2s[4], ones[4] = {1, 1, 1, 1};
        for (int q = 0; q < 4; ++q) {
            const int J = b[q].n_oth;
            if (J == 0) continue;
            f16* C = reinterpret_cast<f16*>(static_cast<unsigned char*>(stash) + (batched ? (size_t)q * (need(rows) / 16 * 16) : 0));   // [ns][pad8(J)]   anchor-major
            f16* CT = C + ((size_t)ns * pad8(J) + 7) / 8 * 8;                // [J][pad8(ns)]   negative-major
            W16Args a{};
            a.K = Dp; a.kper = (Dp + W_KC - 1) / W_KC * W_KC;
            a.k0 = LOG2E / tau0; a.k1 = LOG2E / tau1; a.it0 = 1.f / tau0; a.it1 = 1.f / tau1; a.gs = gs8; a.fam = b[q].fam;
            // ONE S pass writes both orientations: C [anchors of the block][pad8(J)] and C^T [J][pad8(ns)]
            a.A = Z + (size_t)(b[q].own_row + lo) * Dp; a.lda = Dp; a.M = ns;
            a.B = Z + (size_t)b[q].oth_row * Dp; a.ldb = Dp; a.N = J;
            a.cout = C; a.ldc = pad8(J);
            a.cout2 = CT; a.ldc2 = pad8(ns);
            if (batched) bc.a[bc.n++] = a; else launch<W_COEF>(a, 1, s);
            // dZ[anchors] += alpha C Z[negatives]          (B operand: ZhT rows = columns d, k = negative)
            W16Args g{};
            g.k0 = a.k0; g.k1 = a.k1; g.it0 = a.it0; g.it1 = a.it1; g.gs = gs8; g.fam = b[q].fam;
            g.A = C; g.lda = pad8(J); g.M = ns;
            g.B = ZT + b[q].oth_col; g.ldb = sc.ldt; g.N = Dp;
            g.K = J;
            g.out = dZ + (size_t)(b[q].own_row + lo) * Dp; g.ldo = Dp;
            int ks = ksplit_for(g.M, g.N, g.K, nfam, g.kper);
            if (batched) { k1s[bg1.n] = ks; bg1.a[bg1.n++] = g; } else launch<W_GEMM>(g, ks, s);
            // dZ[negatives] += alpha C^T Z[anchors of the block]
            g.A = CT; g.lda = pad8(ns); g.M = J;
            g.B = ZT + b[q].own_col + lo; g.N = Dp;
            g.K = ns;
            g.out = dZ + (size_t)b[q].oth_row * Dp;
            ks = ksplit_for(g.M, g.N, g.K, nfam, g.kper);
            if (batched) { k2s[bg2.n] = ks; bg2.a[bg2.n++] = g; } else launch<W_GEMM>(g, ks, s);
        }
        if (batched && bc.n) {
            launch_batch<W_COEF>(bc, ones, s);
            launch_batch<W_GEMM>(bg1, k1s, s);
            launch_batch<W_GEMM>(bg2, k2s, s);
        }
    }
    SGA_CHECK_LAUNCH("sga_loss_neg_grad_f16");
    return SGA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The tile core as a service (wide16_api.h): up to eight fp32 similarity blocks out = A B^T in one launch.
// ---------------------------------------------------------------------------------------------------------------------------------------
int sga_wide16_store_batch(const SgaW16Store* e, int n, hipStream_t s) {
    if (n < 0 || n > 8) { sga_set_error("sga_wide16_store_batch: %d products (at most 8)", n); return SGA_ERR_ARG; }
    W16Batch b{};
    int ones[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int q = 0; q < n; ++q) {
        if (e[q].M <= 0 || e[q].N <= 0 || e[q].K <= 0) continue;
        if (!e[q].A || !e[q].B || !e[q].out || e[q].K % 8 || e[q].lda % 8 || e[q].ldb % 8) {
            sga_set_error("sga_wide16_store_batch: product %d: null operand or K / row pitch not a multiple of 8", q);
            return SGA_ERR_ARG;
        }
        W16Args a{};
        a.A = static_cast<const f16*>(e[q].A); a.lda = e[q].lda; a.M = e[q].M;
        a.B = static_cast<const f16*>(e[q].B); a.ldb = e[q].ldb; a.N = e[q].N;
        a.K = e[q].K; a.kper = (e[q].K + W_KC - 1) / W_KC * W_KC;
        a.out = e[q].out; a.ldo = e[q].ldo;
        b.a[b.n++] = a;
    }
    if (b.n) launch_batch<W_STORE>(b, ones, s);
    SGA_CHECK_LAUNCH("sga_wide16_store_batch");
    return SGA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The two stash products of the anchors x anchors backward (sga_loss_stash_grad) on the same core, for a WIDE table in mode 'f16':
//   dX2[j, :] += sum_i M1[j, i] X1[a_lo + i, :]        dX1[a_lo + i, :] += sum_j M1[j, i] X2[j, :]        M1 [A][ns] fp32 = (dL/dS)^T
// (the autograd of losses.py:6,50-57,81-94 through S = X1 X2^T).  The coefficients enter the matrix cores as fp16 of 2^e M1, 2^e the power of
// two that puts the block's largest magnitude into [2^13, 2^14) (found by a max pass; exact scaling, undone by the GEMM's alpha; what falls
// below 2^-38 of the largest coefficient flushes to zero), in both orientations (M1h [A][pad8 ns], M1Th [ns][pad8 A], written by one
// transposing pass); the B operands are the table's ZhT rows of sga_wide16_prepare.  fp32 accumulate, fp32 atomics into dZ.
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void stash16_absmax_kernel(const float* __restrict__ M1, size_t n, unsigned* __restrict__ amax) {
    __shared__ float wm[4];
    float m = 0.f;
    const size_t n4 = n / 4;                                 // (M1 is 16-byte aligned)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(M1)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(M1[n4 * 4 + threadIdx.x]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(amax, __float_as_uint(m));    // non-negative floats order like their bit patterns
    }
}
// hdr: [0] max |M1| (bits), [1] alpha = 2^-e (for the GEMMs).  64 x 64 tiles of M1 (rows j, columns i).
__global__ __launch_bounds__(256) void stash16_convert_kernel(const float* __restrict__ M1, int A, int ns, float* __restrict__ hdr,
                                                              f16* __restrict__ Mh, int ldh, f16* __restrict__ MTh, int ldt2) {
    __shared__ float tile[64][65];
    const float amax = hdr[0];
    int ex = 0;
    if (amax > 0.f) (void)frexpf(amax, &ex);                      // amax = f 2^ex, f in [0.5, 1)
    const float scale = amax > 0.f ? ldexpf(1.f, 14 - ex) : 1.f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hdr[1] = amax > 0.f ? ldexpf(1.f, ex - 14) : 1.f;
    const int j0 = blockIdx.y * 64, i0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = j0 + ty * 16 + q, i = i0 + tx;
        const float v = (j < A && i < ns) ? M1[(size_t)j * ns + i] * scale : 0.f;
        if (j < A && i < ldh) Mh[(size_t)j * ldh + i] = (f16)v;
        tile[ty * 16 + q][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = i0 + ty * 16 + q, j = j0 + tx;
        if (i < ns && j < ldt2) MTh[(size_t)i * ldt2 + j] = (f16)tile[tx][ty * 16 + q];
    }
}
}  // namespace

extern "C" size_t sga_loss_stash_grad_f16_bytes(int A, int ns) {
    return 256 + sizeof(f16) * ((size_t)A * pad8(ns) + (size_t)ns * pad8(A)) + 256;
}

extern "C" int sga_loss_stash_grad_f16(const float* M1, const void* ZhT, int Dp, int A, int J1, int J2, float* dZ, int a_lo, int a_hi,
                                       void* ws, size_t ws_bytes, void* stream) {
    SGA_CHECK_ARG(A >= 0 && J1 >= 0 && J2 >= 0 && Dp >= 8 && Dp % 8 == 0 && a_lo >= 0 && a_hi <= A && a_lo <= a_hi && (a_lo % 8 == 0 || a_hi == a_lo),
                  "sga_loss_stash_grad_f16: bad sizes (Dp and a_lo must be multiples of 8)");
    const int ns = a_hi - a_lo;
    if (A == 0 || ns == 0) return SGA_OK;
    SGA_CHECK_ARG(M1 && ZhT && dZ && ws && reinterpret_cast<uintptr_t>(M1) % 16 == 0, "sga_loss_stash_grad_f16: null or misaligned pointer");
    SGA_CHECK_ARG(ws_bytes >= sga_loss_stash_grad_f16_bytes(A, ns), "sga_loss_stash_grad_f16: workspace of %zu bytes, %zu needed", ws_bytes,
                  sga_loss_stash_grad_f16_bytes(A, ns));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* hdr = static_cast<float*>(ws);
    f16* Mh = reinterpret_cast<f16*>(static_cast<unsigned char*>(ws) + 256);
    const int ldh = pad8(ns), ldt2 = pad8(A);
    f16* MTh = Mh + ((size_t)A * ldh + 127) / 128 * 128;
    if (hipMemsetAsync(ws, 0, 16, s) != hipSuccess) { sga_set_error("sga_loss_stash_grad_f16: memset failed"); return SGA_ERR_HIP; }
    const size_t n = (size_t)A * ns;
    const int nb = (int)((n + 256 * 32 - 1) / (256 * 32));
    hipLaunchKernelGGL(stash16_absmax_kernel, dim3(nb < 1024 ? (nb > 0 ? nb : 1) : 1024), dim3(256), 0, s, M1, n, reinterpret_cast<unsigned*>(ws));
    hipLaunchKernelGGL(stash16_convert_kernel, dim3((ldh + 63) / 64, (ldt2 + 63) / 64), dim3(256), 0, s, M1, A, ns, hdr, Mh, ldh, MTh, ldt2);
    const SegCols sc = seg_cols(A, J1, J2);
    const f16* ZT = static_cast<const f16*>(ZhT);
    const int ncu = sga_num_cus();
    auto ksplit_for = [&](int M, int N, int K, int& kper) {
        return choose_ksplit(((M + W_T - 1) / W_T) * ((N + W_T - 1) / W_T), K, ncu, kper);      // (the two products differ in shape: each on its own)
    };
    W16Batch b{};
    int ks[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    W16Args g{};
    g.alpha_dev = hdr + 1;
    // dX2[j] += sum_i M1[j, i] X1[a_lo + i]          (B operand: ZhT rows = columns d, k = anchor a_lo + i of segment X1)
    g.A = Mh; g.lda = ldh; g.M = A;
    g.B = ZT + sc.x1 + a_lo; g.ldb = sc.ldt; g.N = Dp;
    g.K = ns;
    g.out = dZ + (size_t)A * Dp; g.ldo = Dp;
    ks[0] = ksplit_for(g.M, g.N, g.K, g.kper);
    b.a[b.n++] = g;
    // dX1[a_lo + i] += sum_j M1[j, i] X2[j]
    g.A = MTh; g.lda = ldt2; g.M = ns;
    g.B = ZT + sc.x2; g.N = Dp;
    g.K = A;
    g.out = dZ + (size_t)a_lo * Dp;
    ks[1] = ksplit_for(g.M, g.N, g.K, g.kper);
    b.a[b.n++] = g;
    launch_batch<W_GEMM>(b, ks, s);
    SGA_CHECK_LAUNCH("sga_loss_stash_grad_f16");
    return SGA_OK;
}
