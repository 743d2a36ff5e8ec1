// Generic exact-fp32 MFMA GEMM used by the small dense layers of the path and by the loss backward.
//
//   C[M,N] (+)= opA(A)[M,K] * opB(B)[K,N] (+ bias[N])
//
// Replaces, on the reference path: nn.Linear forward/backward for object_embedding /
// structure_embedding / meta_embedding_rel / meta_embedding_attr (src/aligner/sg_aligner.py:112,116,
// 119,122; the .float() casts of the f64 bag-of-words inputs at :73-74 are folded into the A loader),
// and the dS * E products of the contrastive-loss backward (autograd of src/aligner/losses.py:6-8).
//
// v_mfma_f32_32x32x2_f32, 128x128 block tile, 4 waves; a wave owns 32 output columns (so C stores are
// 128-byte coalesced) and walks four 32-row tiles; K is staged through LDS in chunks of 32 with
// conflict-free ds_read_b128 operand reads (mfma_tiles.h).  Optional split-K (atomic fp32 adds) for
// the weight-gradient shape (M,N small, K = number of objects).
#include "mfma_tiles.h"

namespace {

constexpr int GM_THREADS = 256;

// Stage a [128][SGA_KC] operand tile: element (row, k) of op(X).
//   kmajor == 0 : X is [rows][K]   (k contiguous)     -> X[row*ld + k]
//   kmajor == 1 : X is [K][rows]   (row contiguous)   -> X[k*ld + row]
template <typename TIn>
__device__ __forceinline__ void stage_tile(float* __restrict__ tile, const TIn* __restrict__ X, long ld, int kmajor,
                                           int row0, int nrows, int k0, int kend, int tid, bool vec_ok) {
    if (!kmajor) {
        if (vec_ok && sizeof(TIn) == 4) {
            constexpr int V = SGA_KC / 4;
            for (int e = tid; e < 128 * V; e += GM_THREADS) {
                const int r = e / V, c = (e % V) * 4;
                const int gr = row0 + r, gk = k0 + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gr < nrows) {
                    if (gk + 3 < kend) {
                        v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + (size_t)gr * ld + gk);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (gk + j < kend) v[j] = (float)X[(size_t)gr * ld + gk + j];
                    }
                }
                *reinterpret_cast<f32x4*>(tile + r * SGA_LDS_STRIDE + c) = v;
            }
        } else {
            for (int e = tid; e < 128 * SGA_KC; e += GM_THREADS) {
                const int r = e / SGA_KC, c = e % SGA_KC;
                const int gr = row0 + r, gk = k0 + c;
                tile[r * SGA_LDS_STRIDE + c] = (gr < nrows && gk < kend) ? (float)X[(size_t)gr * ld + gk] : 0.f;
            }
        }
    } else {
        for (int e = tid; e < 128 * SGA_KC; e += GM_THREADS) {
            const int r = e % 128, c = e / 128;
            const int gr = row0 + r, gk = k0 + c;
            tile[r * SGA_LDS_STRIDE + c] = (gr < nrows && gk < kend) ? (float)X[(size_t)gk * ld + gr] : 0.f;
        }
    }
}

template <typename TA>
__global__ __launch_bounds__(GM_THREADS) void gemm_kernel(const TA* __restrict__ A, long lda, int transA,
                                                          const float* __restrict__ B, long ldb, int transB,
                                                          float* __restrict__ C, long ldc,
                                                          const float* __restrict__ bias, int M, int N, int K,
                                                          int accumulate, int k_per_split, int use_atomic,
                                                          int a_vec_ok, int b_vec_ok, int act,
                                                          const float* __restrict__ resid, long ldr) {
    __shared__ __attribute__((aligned(16))) float As[128 * SGA_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[128 * SGA_LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    f32x16 acc[4];
    zero_acc<4>(acc);
    for (int k0 = kbeg; k0 < kend; k0 += SGA_KC) {
        __syncthreads();
        stage_tile<TA>(As, A, lda, transA, m0, M, k0, kend, tid, a_vec_ok);
        stage_tile<float>(Bs, B, ldb, !transB, n0, N, k0, kend, tid, b_vec_ok);
        __syncthreads();
        // MFMA "A" rows = output rows m (4 tiles), MFMA "B" row = this wave's output column n
        mfma_chunk<4>(acc, As, Bs + (wave * 32 + (lane & 31)) * SGA_LDS_STRIDE, lane);
    }
    const int n = n0 + wave * 32 + (lane & 31);
    if (n >= N) return;
    const float bv = (bias && blockIdx.z == 0) ? bias[n] : 0.f;
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + t * 32 + mfma32_row(r, h);
            if (m < M) {
                float* p = C + (size_t)m * ldc + n;
                float v = acc[t][r] + bv;
                if (use_atomic) { atomicAdd(p, v); continue; }
                if (accumulate) v += *p;
                if (act == 1) v = fmaxf(v, 0.f);                       // ReLU
                else if (act == 2) v = v > 0.f ? v : 0.2f * v;        // LeakyReLU(0.2)
                if (resid) v += resid[(size_t)m * ldr + n];           // residual AFTER the activation (x + act(...))
                *p = v;
            }
        }
    }
}

// Small outputs with layouts the fast kernels do not take (K = 3 / 41: not a multiple of 4, float64 bag-of-words inputs, transposed
// operands at K = a few hundred objects): the generic kernel above gives such a problem to one or two workgroups that walk K in
// barrier-separated chunks (30-45 us for a 320 x 100 x 164 product: 7 launches and 0.24 ms of a 1.6 ms step at the reference's batch
// sizes).  Here a workgroup owns ONE 32 x 32 output tile and its 4 waves split K between them (wave w takes k = 8 i + 2 w, + 1);
// operands come straight from global memory / L2 by element (any stride, any K, fp64 A converted on load), the four partial tiles
// are added through LDS.  No K chunking, no staging barriers.
template <typename TA>
__global__ __launch_bounds__(GM_THREADS) void gemm_small_kernel(const TA* __restrict__ A, long lda, int transA, const float* __restrict__ B,
                                                                long ldb, int transB, float* __restrict__ C, long ldc,
                                                                const float* __restrict__ bias, int M, int N, int K, int accumulate,
                                                                int act, const float* __restrict__ resid, long ldr) {
    __shared__ float red[3][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int m = m0 + l31, n = n0 + l31;
    const bool mv = m < M, nv = n < N;
    const long am = transA ? (long)(mv ? m : 0) : (long)(mv ? m : 0) * lda;     // + k * (transA ? lda : 1)
    const long ak = transA ? lda : 1;
    const long bn = transB ? (long)(nv ? n : 0) * ldb : (long)(nv ? n : 0);     // + k * (transB ? 1 : ldb)
    const long bk = transB ? 1 : ldb;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int k0 = 2 * wave; k0 < K; k0 += 8) {
        const int k = k0 + h;
        const bool kv = k < K;
        const float av = (mv && kv) ? (float)A[am + (long)k * ak] : 0.f;
        const float bv = (nv && kv) ? B[bn + (long)k * bk] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    // acc[r] = partial C[m0 + row(r,h)][n0 + l31]
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][mfma32_row(r, h) * 33 + l31] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && nv) {
        const float bvv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, h), mm = m0 + row;
            if (mm < M) {
                float v = acc[r] + red[0][row * 33 + l31] + red[1][row * 33 + l31] + red[2][row * 33 + l31] + bvv;
                float* p = C + (size_t)mm * ldc + n;
                if (accumulate) v += *p;
                if (act == 1) v = fmaxf(v, 0.f);
                else if (act == 2) v = v > 0.f ? v : 0.2f * v;
                if (resid) v += resid[(size_t)mm * ldr + n];
                *p = v;
            }
        }
    }
}

// Epilogue of one 32 x 32 accumulator tile: C[m_base + row(r, h)][n] = act(acc + bias (+ C)) (+ resid), column statistics on the side.
// Addressing: ONE 64-bit row pointer per tile and lane, the sixteen rows of a lane (row(r, h) = (r & 3) + 8 (r >> 2) + 4 h) by 32-bit
// offsets that are sums of a few multiples of ldc -- `C + (size_t)m * ldc + n` per element was two v_mul_lo_u32, a v_mad_u64_u32 and two more
// 64-bit adds for every output (a K = 128 layer over 163 840 rows spent 38 of its 77 us there whatever its K: tools/bench_gemm.py with the
// epilogue compiled out).  The optional C / residual loads of a tile are issued together; interior tiles carry no row masks.
__device__ __forceinline__ void store_tile32(const f32x16& a, float bv, float* __restrict__ C, long ldc, int m_base, int h, int M, int n, bool nv,
                                             int accumulate, int act, const float* __restrict__ resid, long ldr, float& cs, float& cq) {
    if (!nv) return;
    const bool full = m_base + 32 <= M;                   // uniform
    float* cp = C + (size_t)(m_base + 4 * h) * ldc + n;
    const int l1 = (int)ldc;                              // (a 32-row tile of a matrix whose row pitch fits 31 bits: checked by the launcher)
    const float* rp = resid ? resid + (size_t)(m_base + 4 * h) * ldr + n : nullptr;
    const int r1 = (int)ldr;
    const float slope = act == 2 ? 0.2f : 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                         // four consecutive rows at a time (registers: the fp32 kernel keeps three workgroups per CU)
        float old[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (full || m_base + 4 * h + 8 * q + e < M) old[e] = cp[(8 * q + e) * l1];
        }
        if (resid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (full || m_base + 4 * h + 8 * q + e < M) rs[e] = rp[(8 * q + e) * r1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (a[4 * q + e] + bv) + old[e];
            const float neg = act == 1 ? 0.f : slope * v;
            v = v > 0.f ? v : neg;
            v += rs[e];
            if (full || m_base + 4 * h + 8 * q + e < M) {
                cp[(8 * q + e) * l1] = v;
                cs += v; cq = fmaf(v, v, cq);
            }
        }
    }
}

// C = act(A W^T + bias) (+ resid) for the common layout (A [M,K] and W [N,K] both k-contiguous, 16-byte aligned rows,
// K % 4 == 0, no split-K): the next K chunk travels global -> registers while the MFMAs of the current one run, so a
// workgroup does not alternate between "everybody loads" and "everybody multiplies" (the generic kernel above: 59
// TFLOP/s on the PCT per-point layers).
template <int MT>
__global__ __launch_bounds__(GM_THREADS) void gemm_nt_kernel(const float* __restrict__ A, long lda,
                                                             const float* __restrict__ B, long ldb,
                                                             float* __restrict__ C, long ldc,
                                                             const float* __restrict__ bias, int M, int N, int K,
                                                             int accumulate, int act, const float* __restrict__ resid,
                                                             long ldr, double* __restrict__ colstats) {
    // MT = rows of A per workgroup (128, or 64 when the 128-row grid would leave most of the chip idle in its last round: gemm_launch)
    constexpr int NTA = MT / 32;
    __shared__ __attribute__((aligned(16))) float As[MT * SGA_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[128 * SGA_LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * MT, n0 = blockIdx.y * 128;
    constexpr int V = SGA_KC / 4;                       // quads per tile row
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid, r = e / V, c = (e % V) * 4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const bool kin = k0 + c < K;                // K % 4 == 0: a quad is inside or outside as a whole
            if (i < NTA) ra[i] = (m0 + r < M && kin) ? *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * lda + k0 + c) : z;
            rb[i] = (n0 + r < N && kin) ? *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + r) * ldb + k0 + c) : z;
        }
    };
    f32x16 acc[NTA];
    zero_acc<NTA>(acc);
    gload(0);
    for (int k0 = 0; k0 < K; k0 += SGA_KC) {
        __syncthreads();                                // the previous chunk's operand reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid, r = e / V, c = (e % V) * 4;
            if (i < NTA) *reinterpret_cast<f32x4*>(As + r * SGA_LDS_STRIDE + c) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + r * SGA_LDS_STRIDE + c) = rb[i];
        }
        __syncthreads();
        if (k0 + SGA_KC < K) gload(k0 + SGA_KC);       // in flight under the MFMAs below
        mfma_chunk<NTA>(acc, As, Bs + (wave * 32 + (lane & 31)) * SGA_LDS_STRIDE, lane);
    }
    const int n = n0 + wave * 32 + (lane & 31);
    const bool nv = n < N;
    if (!nv && !colstats) return;
    const float bv = (bias && nv) ? bias[n] : 0.f;
    const int h = lane >> 5;
    float cs = 0.f, cq = 0.f;                           // column sum / sum of squares of what is written (colstats != null)
#pragma unroll
    for (int t = 0; t < NTA; ++t) store_tile32(acc[t], bv, C, ldc, m0 + t * 32, h, M, n, nv, accumulate, act, resid, ldr, cs, cq);
    if (colstats) {
        // BatchNorm batch statistics of the output in the epilogue that produces it (pct.py: every conv is followed by a BatchNorm):
        // the two lane halves of a column fold, then one fp64 atomic pair per column and workgroup -- sums[0..N) = sum, [N..2N) = sum of squares
        cs += __shfl_xor(cs, 32, 64); cq += __shfl_xor(cq, 32, 64);
        if (h == 0 && nv) { atomicAdd(colstats + n, (double)cs); atomicAdd(colstats + N + n, (double)cq); }
    }
}

// The same product with every fp32 operand split EXACTLY into three bf16 terms (x = h + m + l, 8 + 8 + 8 significand bits, fp32's exponent range:
// no pre-scale, no range condition -- the arithmetic of the loss sweeps, sweep3.hip): a product is the six partial products
// h h' | h m' + m h' | m m' + h l' + l h' on v_mfma_f32_32x32x16_bf16 into fp32 accumulators (the dropped three are <= 2^-23 of the
// product: below the rounding of the fp32 accumulation that both this kernel and the fp32 MFMA perform), 6/16 of the fp32 MFMA's matrix
// time.  The split happens ONCE per staged element, on its way from the prefetch registers to LDS (11 VALU per pair of values; an operand
// element is then read by every wave / every column tile); LDS holds three bf16 planes per operand, [plane][row][32 k] with 80-byte rows
// (64 + 16: the 16 rows a ds_read_b128 lane group touches land on 16 different 16-byte slots of the 256-byte bank row).
// TWO accumulators per output: the h h' products and the five small ones apart -- the 16-bit MFMA aligns its products to the largest exponent
// and chops what falls below toward minus infinity whatever the sign (tools/micro/mfma_round_probe.hip), and a BatchNorm batch sum over 10^5
// rows of the output would collect that one-sided bias coherently; in their own accumulator nothing of the small products is chopped.
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int g_u32x2 __attribute__((ext_vector_type(2)));
constexpr int N3_RS = 80;                                // bytes per LDS row of a plane
__device__ __forceinline__ void g_split3_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    const g_bf16x2 H = __builtin_convertvector(g_f32x2{v0, v1}, g_bf16x2);
    const unsigned hu = __builtin_bit_cast(unsigned, H);
    const float r0 = v0 - __builtin_bit_cast(float, hu << 16);
    const float r1 = v1 - __builtin_bit_cast(float, hu & 0xffff0000u);
    const g_bf16x2 Mi = __builtin_convertvector(g_f32x2{r0, r1}, g_bf16x2);
    const unsigned mu = __builtin_bit_cast(unsigned, Mi);
    const float s0 = r0 - __builtin_bit_cast(float, mu << 16);
    const float s1 = r1 - __builtin_bit_cast(float, mu & 0xffff0000u);
    const g_bf16x2 Lo = __builtin_convertvector(g_f32x2{s0, s1}, g_bf16x2);
    h = hu; m = mu; l = __builtin_bit_cast(unsigned, Lo);
}
template <int MT>
__global__ __launch_bounds__(GM_THREADS, 2) void gemm_nt3_kernel(const float* __restrict__ A, long lda,
                                                                 const float* __restrict__ B, long ldb,
                                                                 float* __restrict__ C, long ldc,
                                                                 const float* __restrict__ bias, int M, int N, int K,
                                                                 int accumulate, int act, const float* __restrict__ resid,
                                                                 long ldr, double* __restrict__ colstats) {
    constexpr int NTA = MT / 32;
    constexpr int APL = MT * N3_RS, BPL = 128 * N3_RS;  // bytes of one plane
    __shared__ __attribute__((aligned(16))) unsigned char As3[3 * APL];
    __shared__ __attribute__((aligned(16))) unsigned char Bs3[3 * BPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * MT, n0 = blockIdx.y * 128;
    constexpr int V = SGA_KC / 4;                       // quads per tile row
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid, r = e / V, c = (e % V) * 4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const bool kin = k0 + c < K;                // K % 4 == 0: a quad is inside or outside as a whole
            if (i < NTA) ra[i] = (m0 + r < M && kin) ? *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * lda + k0 + c) : z;
            rb[i] = (n0 + r < N && kin) ? *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + r) * ldb + k0 + c) : z;
        }
    };
    auto put3 = [&](unsigned char* base, int plane_bytes, int r, int c, const f32x4& v) {
        unsigned h0, m0_, l0, h1, m1, l1;
        g_split3_pair(v[0], v[1], h0, m0_, l0);
        g_split3_pair(v[2], v[3], h1, m1, l1);
        unsigned char* p = base + r * N3_RS + c * 2;
        *reinterpret_cast<g_u32x2*>(p) = g_u32x2{h0, h1};
        *reinterpret_cast<g_u32x2*>(p + plane_bytes) = g_u32x2{m0_, m1};
        *reinterpret_cast<g_u32x2*>(p + 2 * plane_bytes) = g_u32x2{l0, l1};
    };
    f32x16 acc[NTA], accs[NTA];
    zero_acc<NTA>(acc);
    zero_acc<NTA>(accs);
    gload(0);
    const unsigned char* bp = Bs3 + (wave * 32 + l31) * N3_RS + h * 16;
    const unsigned char* ap = As3 + l31 * N3_RS + h * 16;
    for (int k0 = 0; k0 < K; k0 += SGA_KC) {
        __syncthreads();                                // the previous chunk's operand reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid, r = e / V, c = (e % V) * 4;
            if (i < NTA) put3(As3, APL, r, c, ra[i]);
            put3(Bs3, BPL, r, c, rb[i]);
        }
        __syncthreads();
        if (k0 + SGA_KC < K) gload(k0 + SGA_KC);       // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < SGA_KC / 16; ++kk) {
            const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(bp + kk * 32);
            const g_bf16x8 bm = *reinterpret_cast<const g_bf16x8*>(bp + BPL + kk * 32);
            const g_bf16x8 bl = *reinterpret_cast<const g_bf16x8*>(bp + 2 * BPL + kk * 32);
#pragma unroll
            for (int t = 0; t < NTA; ++t) {
                const unsigned char* at = ap + t * 32 * N3_RS + kk * 32;
                const g_bf16x8 ah = *reinterpret_cast<const g_bf16x8*>(at);
                const g_bf16x8 am = *reinterpret_cast<const g_bf16x8*>(at + APL);
                const g_bf16x8 al = *reinterpret_cast<const g_bf16x8*>(at + 2 * APL);
                accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accs[t], 0, 0, 0);
                accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accs[t], 0, 0, 0);
                accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, accs[t], 0, 0, 0);
                accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, accs[t], 0, 0, 0);
                accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, accs[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    }
    const int n = n0 + wave * 32 + l31;
    const bool nv = n < N;
    if (!nv && !colstats) return;
    const float bv = (bias && nv) ? bias[n] : 0.f;
    float cs = 0.f, cq = 0.f;                           // column sum / sum of squares of what is written (colstats != null)
#pragma unroll
    for (int t = 0; t < NTA; ++t) {
        f32x16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = acc[t][r] + accs[t][r];
        store_tile32(sum, bv, C, ldc, m0 + t * 32, h, M, n, nv, accumulate, act, resid, ldr, cs, cq);
    }
    if (colstats) {
        cs += __shfl_xor(cs, 32, 64); cq += __shfl_xor(cq, 32, 64);
        if (h == 0 && nv) { atomicAdd(colstats + n, (double)cs); atomicAdd(colstats + N + n, (double)cq); }
    }
}

// C[M,N] (+)= A^T B for A [K,M], B [K,N] with a NARROW B (N <= 8: the weight gradient of a layer that reads raw coordinates -- the first
// Conv1d(3, .) of the PCT / PointNet encoders over 10^5 point rows: the MFMA kernels want N % 4 == 0, and the generic kernel walked it in
// 620 K splits at 0.1 TB/s).  Memory-bound by A: a lane owns an output row m (A's column: coalesced 256-byte reads per k row), the N
// values of B's row k are the same for every lane (scalar loads), the four waves of a workgroup take every fourth k row of the chunk, fold
// through LDS, and one set of fp32 atomics per workgroup goes to C (zeroed by the host unless the caller accumulates).
__global__ __launch_bounds__(GM_THREADS) void gemm_tn_narrow_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                                    float* __restrict__ C, long ldc, int M, int N, int K, int kchunk) {
    __shared__ float red[3][8][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 64 + lane;
    const int k0 = blockIdx.y * kchunk, k1 = min(K, k0 + kchunk);
    const bool mv = m < M;
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.f;
    for (int k = k0 + wave; k < k1; k += 4) {
        const float a = mv ? A[(size_t)k * lda + m] : 0.f;
        const float* br = B + (size_t)k * ldb;
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < N) acc[n] = fmaf(a, br[n], acc[n]);
    }
    if (wave > 0) {
#pragma unroll
        for (int n = 0; n < 8; ++n) red[wave - 1][n][lane] = acc[n];
    }
    __syncthreads();
    if (wave == 0 && mv) {
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < N) {
                const float v = acc[n] + red[0][n][lane] + red[1][n][lane] + red[2][n][lane];
                if (v != 0.f) atomicAdd(C + (size_t)m * ldc + n, v);
            }
    }
}

// C[M,N] += A^T B for A [K,M], B [K,N] both row-major (the weight-gradient shape dW = dY^T X with K = rows of the
// batch): the row-major [32 k][128] chunks go to LDS as they are (f32x4 in, f32x4 out) and the MFMA operands are read
// as ds_read_b32 with lane = output index (conflict free), K pair (2s, 2s+1) per step -- no transposed staging.
// Split over K with atomic adds (C zeroed by the caller); next chunk prefetched into registers under the MFMAs.
constexpr int TN_STRIDE = 132;

__global__ __launch_bounds__(GM_THREADS) void gemm_tn_kernel(const float* __restrict__ A, long lda,
                                                             const float* __restrict__ B, long ldb,
                                                             float* __restrict__ C, long ldc, int M, int N, int K,
                                                             int k_per_split) {
    __shared__ __attribute__((aligned(16))) float As[SGA_KC * TN_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[SGA_KC * TN_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid, r = e >> 5, c = (e & 31) * 4;      // 32 k-rows x 32 quads
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const bool kin = k0 + r < kend;
            ra[i] = (kin && m0 + c < M) ? *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + r) * lda + m0 + c) : z;   // M % 4 == 0
            rb[i] = (kin && n0 + c < N) ? *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + r) * ldb + n0 + c) : z;   // N % 4 == 0
        }
    };
    f32x16 acc[4];
    zero_acc<4>(acc);
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += SGA_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid, r = e >> 5, c = (e & 31) * 4;
            *reinterpret_cast<f32x4*>(As + r * TN_STRIDE + c) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + r * TN_STRIDE + c) = rb[i];
        }
        __syncthreads();
        if (k0 + SGA_KC < kend) gload(k0 + SGA_KC);
        const float* ap = As + h * TN_STRIDE + l31;
        const float* bp = Bs + h * TN_STRIDE + wave * 32 + l31;
#pragma unroll
        for (int s = 0; s < SGA_KC / 2; ++s) {
            const float bv = bp[2 * s * TN_STRIDE];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * TN_STRIDE + t * 32], bv, acc[t], 0, 0, 0);
        }
    }
    const int n = n0 + wave * 32 + l31;
    if (n >= N) return;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + t * 32 + mfma32_row(r, h);
            if (m < M) atomicAdd(C + (size_t)m * ldc + n, acc[t][r]);
        }
}

// C[M,N] (+)= A B for A [M,K] (k contiguous) and B [K,N] (n contiguous), no epilogue: A chunks staged [128 m][32 k]
// and read as ds_read_b128 (4 MFMAs per read), B chunks staged row-major [32 k][128 n] as they are and read as
// lane-contiguous ds_read_b32.  Optional split over K with atomic adds.  (dS E of the loss backward: M1 [A, ns] x X1.)
__global__ __launch_bounds__(GM_THREADS) void gemm_nn_kernel(const float* __restrict__ A, long lda,
                                                             const float* __restrict__ B, long ldb,
                                                             float* __restrict__ C, long ldc, int M, int N, int K,
                                                             int accumulate, int k_per_split, int use_atomic) {
    __shared__ __attribute__((aligned(16))) float As[128 * SGA_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[SGA_KC * TN_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    constexpr int V = SGA_KC / 4;
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const int r = e / V, c = (e % V) * 4;                                   // A: 128 rows x 8 quads of k
            ra[i] = (m0 + r < M && k0 + c < kend) ? *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * lda + k0 + c) : z;
            const int kr = e >> 5, cn = (e & 31) * 4;                               // B: 32 k-rows x 32 quads of n
            rb[i] = (k0 + kr < kend && n0 + cn < N) ? *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + kr) * ldb + n0 + cn) : z;
        }
    };
    f32x16 acc[4];
    zero_acc<4>(acc);
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += SGA_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * GM_THREADS + tid;
            *reinterpret_cast<f32x4*>(As + (e / V) * SGA_LDS_STRIDE + (e % V) * 4) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + (e >> 5) * TN_STRIDE + (e & 31) * 4) = rb[i];
        }
        __syncthreads();
        if (k0 + SGA_KC < kend) gload(k0 + SGA_KC);
        const float* ap = As + l31 * SGA_LDS_STRIDE + 4 * h;                        // k = 8q + 4h + r
        const float* bp = Bs + (4 * h) * TN_STRIDE + wave * 32 + l31;
#pragma unroll
        for (int q = 0; q < SGA_KC / 8; ++q) {
            f32x4 a4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[t] = *reinterpret_cast<const f32x4*>(ap + t * 32 * SGA_LDS_STRIDE + 8 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bv = bp[(8 * q + r) * TN_STRIDE];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t][r], bv, acc[t], 0, 0, 0);
            }
        }
    }
    const int n = n0 + wave * 32 + l31;
    if (n >= N) return;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + t * 32 + mfma32_row(r, h);
            if (m < M) {
                float* p = C + (size_t)m * ldc + n;
                if (use_atomic) atomicAdd(p, acc[t][r]);
                else *p = accumulate ? (*p + acc[t][r]) : acc[t][r];
            }
        }
}

// column sums: out[n] (+)= sum_m X[m*ld + n]   (bias gradients)
__global__ void colsum_kernel(const float* __restrict__ X, long ld, int M, int N, float* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rw = threadIdx.x >> 6;                 // 4 row-walkers per block
    float s = 0.f;
    if (n < N)
        for (int m = blockIdx.y * 4 + rw; m < M; m += gridDim.y * 4) s += X[(size_t)m * ld + n];
    __shared__ float red[4][64];
    red[rw][threadIdx.x & 63] = s;
    __syncthreads();
    if (rw == 0 && n < N) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (gridDim.y == 1) out[n] = accumulate ? out[n] + v : v;        // short inputs: plain store, no zeroing launch before
        else atomicAdd(out + n, v);
    }
}

__global__ void cast_f64_f32_kernel(const double* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

}  // namespace

extern "C" int sga_cast_f64_f32(const double* in, float* out, size_t n, void* stream) {
    if (n == 0) return SGA_OK;
    SGA_CHECK_ARG(in && out, "sga_cast_f64_f32: null pointer");
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, n);
    SGA_CHECK_LAUNCH("sga_cast_f64_f32");
    return SGA_OK;
}

static int gemm_launch(int transA, int transB, int M, int N, int K, const void* A, long lda, int a_is_f64,
                       const float* B, long ldb, float* C, long ldc, const float* bias, int accumulate,
                       int act, const float* resid, long ldr, void* stream, double* colstats = nullptr) {
    SGA_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "sga_gemm: negative size");
    // the NT kernel's arithmetic: three exact bf16 planes (gemm_nt3_kernel; fp32-faithful, 6/16 of the fp32 MFMA's matrix time) from K = 256 on
    // -- measured at 163 840 / 40 960 rows (tools/bench_gemm.py): K = 512 -> N = 1024 0.458 -> 0.306 ms, 1024 -> 512 0.407 -> 0.345, 512 -> 256
    // 0.434 -> 0.331, 256 -> 100 0.128 -> 0.121; K = 128 layers are four chunks per tile (latency, not matrix time: 0.080 -> 0.078) and at K = 64 the
    // kernel's 61 KiB of LDS planes cost more occupancy than the MFMAs save (0.054 -> 0.074) -- below 256 the fp32 MFMA kernel stays.  A function of K
    // only, never of M: a batch walked in chunks of rows gets the same bits as the unchunked call.  -DSGA_GEMM_NT_FP32: fp32 MFMA everywhere
    // (the A/B of tools/build_variant.sh).
#ifdef SGA_GEMM_NT_FP32
    const bool nt3 = false;
#else
    const bool nt3 = K >= 256;
#endif
    SGA_CHECK_ARG(act >= 0 && act <= 2, "sga_gemm_ex: act=%d (0 none, 1 relu, 2 leaky-relu 0.2)", act);
    if (M == 0 || N == 0) return SGA_OK;                 // empty output (a zero-row shard): nothing to do, null pointers allowed
    SGA_CHECK_ARG(C && (K == 0 || (A && B)), "sga_gemm: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (transA && !transB && N <= 8 && K >= 64 && !a_is_f64 && !bias && act == 0 && !resid && !colstats) {
        // narrow weight gradient (a function of the shape class only): memory-bound walk over A
        if (!accumulate && hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) { sga_set_error("sga_gemm: memset2d failed"); return SGA_ERR_HIP; }
        const int gxn = (M + 63) / 64;
        int chunks = (4 * sga_num_cus() + gxn - 1) / gxn;
        if (chunks > (K + 31) / 32) chunks = (K + 31) / 32;              // at least 32 k rows per workgroup
        if (chunks < 1) chunks = 1;
        const int kchunk = (K + chunks - 1) / chunks;
        hipLaunchKernelGGL(gemm_tn_narrow_kernel, dim3(gxn, (K + kchunk - 1) / kchunk), dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb,
                           C, ldc, M, N, K, kchunk);
        SGA_CHECK_LAUNCH("sga_gemm");
        return SGA_OK;
    }
    const int gx = (M + 127) / 128, gy = (N + 127) / 128;
    // split K when the output grid cannot fill the chip (weight-gradient shape)
    int splits = 1;
    const int ncu = sga_num_cus();
    if (colstats) {
        // the statistics epilogue lives in the NT kernel only (one workgroup per output tile, no split)
    } else if (gx * gy < ncu && K >= 4096 && act == 0 && !resid) {          // split-K partial sums cannot carry an epilogue
        // ~4 workgroups per CU, at least 256 K-rows each (64 splits of 1024 left 3/4 of the chip idle: 350-470 us per
        // weight gradient at K = 65536 objects)
        splits = min((4 * ncu) / (gx * gy), (K + 255) / 256);
        if (splits < 1) splits = 1;
    } else if (transA && !transB && !a_is_f64 && !bias && act == 0 && !resid && K >= 128 && gx * gy < ncu) {
        // weight gradients at the reference's own batch sizes (K = a few hundred objects): one or two workgroups walking all of K
        // in the generic kernel took 78 us per GEMM; 64-row splits on the TN kernel put the chip to work (~8 us)
        splits = min((4 * ncu) / (gx * gy), (K + 63) / 64);
        if (splits < 1) splits = 1;
    }
    int kper = ((K + splits - 1) / splits + SGA_KC - 1) / SGA_KC * SGA_KC;
    if (kper < SGA_KC) kper = SGA_KC;
    splits = K > 0 ? (K + kper - 1) / kper : 1;
    const int use_atomic = splits > 1;
    if (use_atomic && !accumulate) {
        if (ldc == N) {
            if (hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), s) != hipSuccess) { sga_set_error("sga_gemm: memset failed"); return SGA_ERR_HIP; }
        } else {
            if (hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) { sga_set_error("sga_gemm: memset2d failed"); return SGA_ERR_HIP; }
        }
    }
    const bool a_al = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % 4 == 0);
    const bool b_al = (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (ldb % 4 == 0);
    dim3 grid(gx, gy, splits);
    // (also unsplit when the output grid fills the chip by itself -- the wide-table stash products of configs[4], [ns x 1024..3072] over K = A:
    // they fell to the generic kernel at ~9 TFLOP/s, 14 % of that step; the TN kernel always accumulates atomically, so C is zeroed first
    // unless the caller accumulates)
    const bool tn_big = !use_atomic && K >= 256 && gx * gy >= ncu && act == 0 && !resid && !colstats;      // (the TN kernel has no statistics epilogue)
    if (!a_is_f64 && transA && !transB && (use_atomic || tn_big) && a_al && b_al && M % 4 == 0 && N % 4 == 0 && !bias) {
        if (tn_big && !accumulate) {
            if (ldc == N) {
                if (hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), s) != hipSuccess) { sga_set_error("sga_gemm: memset failed"); return SGA_ERR_HIP; }
            } else {
                if (hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) { sga_set_error("sga_gemm: memset2d failed"); return SGA_ERR_HIP; }
            }
        }
        hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb, C, ldc, M, N, K, kper);
        SGA_CHECK_LAUNCH("sga_gemm");
        return SGA_OK;
    }
    if (!a_is_f64 && !transA && !transB && a_al && b_al && K % 4 == 0 && N % 4 == 0 && !bias && act == 0 && !resid) {
        hipLaunchKernelGGL(gemm_nn_kernel, grid, dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb, C, ldc, M, N, K,
                           accumulate, kper, use_atomic);
        SGA_CHECK_LAUNCH("sga_gemm");
        return SGA_OK;
    }
    if (!a_is_f64 && !transA && transB && splits == 1 && a_al && b_al && K % 4 == 0 && ldc < (1L << 26) && ldr < (1L << 26)) {
        // 64-row tiles when the 128-row grid ends in a mostly idle round (4 workgroups per CU: 163 840 rows x 128 columns = 1 280 tiles on
        // 1 024 slots run two rounds for 1.25 rounds of work; 2 560 half tiles run 2.5)
        const int slots = 4 * ncu, tiles = gx * gy, last = tiles % slots;
        if (tiles > slots && tiles < 3 * slots && last > 0 && last * 2 < slots) {
            if (nt3) hipLaunchKernelGGL(gemm_nt3_kernel<64>, dim3((M + 63) / 64, gy), dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb, C, ldc,
                                        bias, M, N, K, accumulate, act, resid, ldr, colstats);
            else
            hipLaunchKernelGGL(gemm_nt_kernel<64>, dim3((M + 63) / 64, gy), dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb, C, ldc,
                               bias, M, N, K, accumulate, act, resid, ldr, colstats);
            SGA_CHECK_LAUNCH("sga_gemm");
            return SGA_OK;
        }
        if (nt3) hipLaunchKernelGGL(gemm_nt3_kernel<128>, dim3(gx, gy), dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb, C, ldc,
                                    bias, M, N, K, accumulate, act, resid, ldr, colstats);
        else
        hipLaunchKernelGGL(gemm_nt_kernel<128>, dim3(gx, gy), dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, B, ldb, C, ldc,
                           bias, M, N, K, accumulate, act, resid, ldr, colstats);
        SGA_CHECK_LAUNCH("sga_gemm");
        return SGA_OK;
    }
    if (colstats) { sga_set_error("sga_gemm_bnstats: shape not taken by the NT kernel (needs K %% 4 == 0, 16-byte aligned operands)"); return SGA_ERR_ARG; }
    // shapes none of the fast kernels take, narrow and short: one workgroup per 32 x 32 tile.  The choice depends on (N, K) only, never on
    // M: a batch walked in chunks of rows (pct inference) must get the same bits as the unchunked call
    if (!use_atomic && N <= 256 && K <= 512) {
        dim3 g2((M + 31) / 32, (N + 31) / 32);
        if (a_is_f64)
            hipLaunchKernelGGL(gemm_small_kernel<double>, g2, dim3(GM_THREADS), 0, s, static_cast<const double*>(A), lda, transA, B, ldb,
                               transB, C, ldc, bias, M, N, K, accumulate, act, resid, ldr);
        else
            hipLaunchKernelGGL(gemm_small_kernel<float>, g2, dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, transA, B, ldb,
                               transB, C, ldc, bias, M, N, K, accumulate, act, resid, ldr);
        SGA_CHECK_LAUNCH("sga_gemm");
        return SGA_OK;
    }
#ifdef SGA_GEMM_TRACE
    fprintf(stderr, "[gemm generic] tA=%d tB=%d M=%d N=%d K=%d splits=%d f64=%d bias=%d acc=%d act=%d\n", transA, transB, M, N, K, splits, a_is_f64, bias != nullptr, accumulate, act);
#endif
    if (a_is_f64)
        hipLaunchKernelGGL(gemm_kernel<double>, grid, dim3(GM_THREADS), 0, s, static_cast<const double*>(A), lda, transA, B, ldb,
                           transB, C, ldc, bias, M, N, K, accumulate, kper, use_atomic, 0, (int)b_al, act, resid, ldr);
    else
        hipLaunchKernelGGL(gemm_kernel<float>, grid, dim3(GM_THREADS), 0, s, static_cast<const float*>(A), lda, transA, B, ldb,
                           transB, C, ldc, bias, M, N, K, accumulate, kper, use_atomic, (int)a_al, (int)b_al, act, resid, ldr);
    SGA_CHECK_LAUNCH("sga_gemm");
    return SGA_OK;
}

extern "C" int sga_gemm(int transA, int transB, int M, int N, int K, const void* A, long lda, int a_is_f64,
                        const float* B, long ldb, float* C, long ldc, const float* bias, int accumulate,
                        void* stream) {
    return gemm_launch(transA, transB, M, N, K, A, lda, a_is_f64, B, ldb, C, ldc, bias, accumulate, 0, nullptr, 0, stream);
}

// C = act(op(A) op(B) + bias) (+ resid): the per-point layers of the PCT object encoder (conv1d k=1 with the
// eval-mode BatchNorm folded into weight/bias, then ReLU / LeakyReLU(0.2), then the SA residual; pct.py:115-124,
// 223-229, 289-293, 311-315)
extern "C" int sga_gemm_ex(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B,
                           long ldb, float* C, long ldc, const float* bias, int act, const float* resid, long ldr,
                           void* stream) {
    return gemm_launch(transA, transB, M, N, K, A, lda, 0, B, ldb, C, ldc, bias, 0, act, resid, ldr, stream);
}

// C = A B^T (+ bias) over point-major rows (A [M,K], B [N,K]) AND the BatchNorm batch statistics of C in the same launch:
// sums[0..N) = column sums, sums[N..2N) = column sums of squares (fp64, zeroed here) -- what sga_bn_stats would compute in a second
// pass over C (pct.py: every Conv1d / Linear of the encoder feeds a BatchNorm1d).  Needs the NT kernel's shape (K % 4 == 0, aligned).
extern "C" int sga_gemm_bnstats(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                                const float* bias, double* sums, void* stream) {
    SGA_CHECK_ARG(sums && N >= 1, "sga_gemm_bnstats: bad argument");
    if (hipMemsetAsync(sums, 0, (size_t)2 * N * sizeof(double), static_cast<hipStream_t>(stream)) != hipSuccess) { sga_set_error("sga_gemm_bnstats: memset failed"); return SGA_ERR_HIP; }
    return gemm_launch(0, 1, M, N, K, A, lda, 0, B, ldb, C, ldc, bias, 0, 0, nullptr, 0, stream, sums);
}

extern "C" int sga_colsum(const float* X, long ld, int M, int N, float* out, int accumulate, void* stream) {
    SGA_CHECK_ARG((X || M == 0) && (out || N == 0) && M >= 0 && N >= 0, "sga_colsum: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) return SGA_OK;
    int gy = M <= 2048 ? 1 : (M + 255) / 256;                             // up to 2048 rows: one workgroup per 64 columns walks them all
    if (gy > 512) gy = 512;
    if (!accumulate && (gy > 1 || M == 0) && hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s) != hipSuccess) { sga_set_error("sga_colsum: memset failed"); return SGA_ERR_HIP; }
    if (M == 0) return SGA_OK;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, gy), dim3(256), 0, s, X, ld, M, N, out, accumulate);
    SGA_CHECK_LAUNCH("sga_colsum");
    return SGA_OK;
}
