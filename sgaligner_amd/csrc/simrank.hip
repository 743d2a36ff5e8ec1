// Per-pair node similarity + ranking for the alignment metrics.
//
// Replaces the eval_step block of reference src/inference/sgaligner/inference_align_reg.py:125-128
// (emb /= ||emb||; sim = 1 - emb emb^T; rank_list = argsort(sim, dim=1)) fused with what
// utils/alignment.py:3-25,27-41,59-70 then read from the rank list: for a query object q of a pair,
// the 1-based rank of its ground-truth match among all OTHER objects of the pair (self removed by
// value, alignment.py:7,18) and its K nearest other objects with their distances.  The full n x n
// sort and the seven device->host copies of the rank list (alignment.py:4,14,29) disappear.
// Ties: broken by object index (a stable ascending sort); the reference's sort is unstable there.
//
// simrank_staged_kernel / simrank_stream_kernel: the per-pair E E^T blocks on the matrix cores.  A workgroup = 4 waves = 64
// consecutive objects of one pair (its "query rows"); only blocks that hold a query are launched (blk_pair / blk_row), so the work
// spreads over all XCDs whatever the query pattern.  A wave holds its 16 rows as the MFMA A operand in registers and walks the
// pair's objects in 16-column tiles; the 16 x n similarity strip lands in LDS and ranking is per query row with lanes across the
// pair's objects: the rank is a wave sum of "closer than the target", the top-K a K-step wave arg-min.
//   fp32: v_mfma_f32_16x16x4_f32 (exact fp32 products, the headline path);
//   f16 : v_mfma_f32_16x16x16_f16 on the L2-normalised rows converted to half, fp32 accumulate (BASELINE.json configs[4]:
//         "MFMA similarity GEMM at fp16"; |sim error| ~1e-3, tolerance 1e-2).
// pair_metrics_kernel: Hits@1..5 counts and the three SGAR flags per pair on the device (one wave per pair), so the host
// only formats the reference's meter dict.
#include "sga_common.h"

namespace {

constexpr int SR_MAXK = 8;
constexpr int SR_MAXPER = 8;          // objects per lane -> up to 512 objects per pair
constexpr int SR_ROWS = 64;           // query rows per workgroup (4 waves x 16)

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// emb / ||emb|| as fp16, once per call: [T][Dp] with Dp = D padded to 32 (zero filled) -- the operand table of simrank_stream16_kernel
__global__ void normalize_f16_kernel(const float* __restrict__ E, int T, int D, int Dp, _Float16* __restrict__ Eh) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < T; t += gridDim.x * wpb) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = E[(size_t)t * D + d]; ss += v * v; }
        const float inv = 1.f / sqrtf(wave_sum(ss));      // no eps, as in the reference (:126)
        for (int d = lane; d < Dp; d += 64) Eh[(size_t)t * Dp + d] = (_Float16)(d < D ? E[(size_t)t * D + d] * inv : 0.f);
    }
}

__global__ void row_inv_norm_kernel(const float* __restrict__ E, int T, int D, float* __restrict__ inv) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < T; t += gridDim.x * wpb) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = E[(size_t)t * D + d]; ss += v * v; }
        ss = wave_sum(ss);
        if (lane == 0) inv[t] = 1.f / sqrtf(ss);          // no eps, as in the reference (:126)
    }
}

__global__ void fill_int_kernel(int* __restrict__ p, int n, int v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
__global__ void scatter_query_kernel(const int* __restrict__ q_idx, int Q, int* __restrict__ obj_query) {
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) obj_query[q_idx[q]] = q;
}

struct SimArgs {
    const float* E; const float* inv; int D, B;
    const int* pair_off;              // [B+1] object offsets
    const int* blk_pair;              // [n_blocks] pair of each workgroup: only 64-row blocks that hold a query are launched,
    const int* blk_row;               // [n_blocks] its row block inside the pair      so the work spreads over all XCDs
    const int* obj_query;             // [T] query id of an object or -1
    const int* q_tgt;                 // [Q] or null
    int K;
    int* rank; int* topk_idx; float* topk_sim;
    int npad_max;                     // LDS strip width (largest pair, padded to 16) + 1
};

// 16-float K group `q` of row `row`: this lane's 4 values k = 16 q + 4 g4 + r, zero past D, scaled
__device__ __forceinline__ f32x4 load_kgroup(const float* __restrict__ row, int D, int q, int g4, float scale) {
    const int k = 16 * q + 4 * g4;
    f32x4 v;
    if (k + 4 <= D && (D & 3) == 0) v = *reinterpret_cast<const f32x4*>(row + k);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = k + r < D ? row[k + r] : 0.f;
    }
    return v * scale;
}

// ranking of one wave's 16 x n similarity strip S (LDS, written by this wave only), one query row at a time with lanes across the
// pair's objects: rank = wave sum of "closer than the target", top-K = K-step wave arg-min
__device__ __forceinline__ void rank_strip(const SimArgs& a, const float* S, int NP, int my_q, int row0, int o0, int n, int lane) {
    for (int rr = 0; rr < 16; ++rr) {
        const int q = __shfl(my_q, rr, 64);                // lanes 0..15 (g4 == 0) hold the rows' query ids
        if (q < 0) continue;
        const float* srow = S + rr * NP;
        const int al = row0 + rr;                          // pair-local self index
        float sim[SR_MAXPER];
#pragma unroll
        for (int u = 0; u < SR_MAXPER; ++u) { const int j = lane + 64 * u; sim[u] = j < n ? srow[j] : INFINITY; }
        const int tg = a.q_tgt ? a.q_tgt[q] - o0 : -1;
        if (tg >= 0 && tg < n) {
            const float st = srow[tg];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) {
                const int j = lane + 64 * u;
                if (j < n && j != al && j != tg && (sim[u] < st || (sim[u] == st && j < tg))) ++cnt;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0) a.rank[q] = cnt + 1;
        } else if (lane == 0) {
            a.rank[q] = -1;
        }
#pragma unroll
        for (int u = 0; u < SR_MAXPER; ++u) if (lane + 64 * u == al) sim[u] = INFINITY;
        for (int k = 0; k < a.K; ++k) {
            float bv = INFINITY; int bj = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) { const int j = lane + 64 * u; if (sim[u] < bv) { bv = sim[u]; bj = j; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oj = __shfl_xor(bj, o, 64);
                if (ov < bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
            }
            if (lane == 0) { a.topk_idx[(size_t)q * a.K + k] = bv < INFINITY ? bj : -1; a.topk_sim[(size_t)q * a.K + k] = bv; }
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) if (lane + 64 * u == bj) sim[u] = INFINITY;
        }
    }
}

// Staged form (D <= 16 KQ, i.e. the 100 / 200 / 300 / 400-wide tables of the path): the workgroup's 4 waves SHARE every 16-column
// tile of the pair's table.  It is fetched once per workgroup with coalesced 16-byte loads into registers one tile ahead (the loads
// fly under the current tile's MFMAs), parked in an LDS tile [16][RS] and read back as MFMA B fragments (ds_read_b128).  The
// streaming form below made every wave fetch every tile itself, one K group at a time: latency-bound at 5-7 TFLOP/s.
template <int KQ, bool F16>
__global__ __launch_bounds__(256) void simrank_staged_kernel(SimArgs a) {
    constexpr int RS = KQ * 16 + 4;                        // LDS row stride of the B tile (floats): 16-byte aligned, rows 4 slots apart
    constexpr int NLD = (KQ * 16 + 63) / 64;               // float4 loads per thread and tile (16 threads per row)
    extern __shared__ float strip[];                       // [4 waves][16][npad_max] | B tile [16][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int b = a.blk_pair[blockIdx.x];
    const int o0 = a.pair_off[b], n = a.pair_off[b + 1] - o0;
    const int row0 = a.blk_row[blockIdx.x] * SR_ROWS + wave * 16;                 // pair-local first row of this wave
    const int my_row = row0 + l15;
    const int my_q = (row0 < n && my_row < n && g4 == 0) ? a.obj_query[o0 + my_row] : -1;
    const bool wave_has = __ballot(my_q >= 0) != 0ull;
    if (!__syncthreads_or(wave_has ? 1 : 0)) return;       // no query in these 64 rows: nothing to do (uniform per workgroup)

    const int NP = a.npad_max;
    float* S = strip + (size_t)wave * 16 * NP;
    float* sB = strip + (size_t)4 * 16 * NP;
    const int D = a.D, nkq = (D + 15) / 16;
    const int arow = o0 + min(my_row, n - 1);
    const float ia = a.inv[arow];
    const float* __restrict__ ap = a.E + (size_t)arow * D;
    f32x4 areg[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) areg[q] = (wave_has && q < nkq) ? load_kgroup(ap, D, q, g4, ia) : f32x4{0.f, 0.f, 0.f, 0.f};

    // staging role of this thread: row sr of the tile, float4 column groups sc, sc + 16, ...
    const int sr = tid >> 4, sc = tid & 15;
    f32x4 stg[NLD];
    auto fetch = [&](int j0) {
        const float* __restrict__ rp = a.E + (size_t)(o0 + min(j0 + sr, n - 1)) * D;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = (sc + 16 * i) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c + 4 <= D && (D & 3) == 0) v = *reinterpret_cast<const f32x4*>(rp + c);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (c + r < D) v[r] = rp[c + r];
            }
            stg[i] = v;
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = (sc + 16 * i) * 4;
            if (c < KQ * 16) *reinterpret_cast<f32x4*>(sB + sr * RS + c) = stg[i];
        }
    };
    fetch(0);
    park();
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 16) {
        if (j0 + 16 < n) fetch(j0 + 16);                   // next tile's loads fly under this tile's MFMAs
        if (wave_has) {
            const float ib = a.inv[o0 + min(j0 + l15, n - 1)];
            const float* bp = sB + l15 * RS + 4 * g4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (q < nkq) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 16 * q) * (F16 ? ib : 1.f);
                    if (F16) {
                        const f16x4 ah = {(_Float16)areg[q][0], (_Float16)areg[q][1], (_Float16)areg[q][2], (_Float16)areg[q][3]};
                        const f16x4 bh = {(_Float16)bv[0], (_Float16)bv[1], (_Float16)bv[2], (_Float16)bv[3]};
                        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[q][r], bv[r], acc, 0, 0, 0);
                    }
                }
            }
            const float cs = F16 ? 1.f : ib;
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(4 * g4 + r) * NP + j0 + l15] = 1.f - acc[r] * cs;
        }
        __syncthreads();                                   // everyone is done with the tile
        if (j0 + 16 < n) park();
        __syncthreads();
    }
    if (wave_has) rank_strip(a, S, NP, my_q, row0, o0, n, lane);
}

// Streaming form for tables wider than 416 columns (BASELINE.json configs[4]: 1024-d embeddings, 3072-d joint): a wave's A operand
// no longer fits its registers, so both operands are read from global memory one K group at a time.
template <bool F16>
__global__ __launch_bounds__(256) void simrank_stream_kernel(SimArgs a) {
    extern __shared__ float strip[];                       // [4 waves][16][npad_max]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int b = a.blk_pair[blockIdx.x];
    const int o0 = a.pair_off[b], n = a.pair_off[b + 1] - o0;
    const int row0 = a.blk_row[blockIdx.x] * SR_ROWS + wave * 16;                 // pair-local first row of this wave
    if (row0 >= n) return;
    const int my_row = row0 + l15;
    const int my_q = (my_row < n && g4 == 0) ? a.obj_query[o0 + my_row] : -1;
    if (__ballot(my_q >= 0) == 0ull) return;               // no query among this wave's rows (wave-uniform)

    const int NP = a.npad_max;
    float* S = strip + (size_t)wave * 16 * NP;
    const int D = a.D, nkq = (D + 15) / 16;
    const int arow = o0 + min(my_row, n - 1);
    const float ia = a.inv[arow];                          // query rows pre-scaled by their inverse norms (emb /= ||emb||, :126)
    const float* __restrict__ ap = a.E + (size_t)arow * D;
    for (int j0 = 0; j0 < n; j0 += 16) {
        const int brow = o0 + min(j0 + l15, n - 1);
        const float ib = a.inv[brow];
        const float* __restrict__ bp = a.E + (size_t)brow * D;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nkq; ++q) {
            const f32x4 av = load_kgroup(ap, D, q, g4, ia);
            const f32x4 bv = load_kgroup(bp, D, q, g4, F16 ? ib : 1.f);
            if (F16) {
                const f16x4 ah = {(_Float16)av[0], (_Float16)av[1], (_Float16)av[2], (_Float16)av[3]};
                const f16x4 bh = {(_Float16)bv[0], (_Float16)bv[1], (_Float16)bv[2], (_Float16)bv[3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc, 0, 0, 0);
            }
        }
        // D layout: lane&15 = column (object j0 + l15), rows 4 g4 + r.  sim = 1 - dot (dot scaled by the column's inverse norm)
        const float cs = F16 ? 1.f : ib;
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(4 * g4 + r) * NP + j0 + l15] = 1.f - acc[r] * cs;
    }
    __builtin_amdgcn_wave_barrier();                       // the strip is written and read by this wave only (DS ops of a wave are ordered)
    rank_strip(a, S, NP, my_q, row0, o0, n, lane);
}

// fp16 form of the streaming kernel (BASELINE.json configs[4], "MFMA similarity GEMM at fp16"): both operands come from the
// normalised fp16 table written ONCE per call by normalize_f16_kernel, as 16-byte groups of 8 halfs feeding v_mfma_f32_16x16x32_f16 --
// half the bytes and an eighth of the MFMA instructions of the fp32 stream per K (the first version converted fp32 rows to half
// inside this loop, twice per tile, and was slower than exact fp32: 2.84 vs 2.15 ms at 4096 queries on the 3072-d table).
__global__ __launch_bounds__(256) void simrank_stream16_kernel(SimArgs a, const _Float16* __restrict__ Eh, int Dp) {
    extern __shared__ float strip[];                       // [4 waves][16][npad_max]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int b = a.blk_pair[blockIdx.x];
    const int o0 = a.pair_off[b], n = a.pair_off[b + 1] - o0;
    const int row0 = a.blk_row[blockIdx.x] * SR_ROWS + wave * 16;
    if (row0 >= n) return;
    const int my_row = row0 + l15;
    const int my_q = (my_row < n && g4 == 0) ? a.obj_query[o0 + my_row] : -1;
    if (__ballot(my_q >= 0) == 0ull) return;
    const int NP = a.npad_max;
    float* S = strip + (size_t)wave * 16 * NP;
    const _Float16* __restrict__ ap = Eh + (size_t)(o0 + min(my_row, n - 1)) * Dp + 8 * g4;
    const int nk = Dp / 32;
    for (int j0 = 0; j0 < n; j0 += 32) {                   // two 16-column tiles per pass: two independent accumulation chains
        const _Float16* __restrict__ bp0 = Eh + (size_t)(o0 + min(j0 + l15, n - 1)) * Dp + 8 * g4;
        const _Float16* __restrict__ bp1 = Eh + (size_t)(o0 + min(j0 + 16 + l15, n - 1)) * Dp + 8 * g4;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 4
        for (int q = 0; q < nk; ++q) {
            const f16x8 av = *reinterpret_cast<const f16x8*>(ap + 32 * q);
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(bp0 + 32 * q);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(bp1 + 32 * q);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S[(4 * g4 + r) * NP + j0 + l15] = 1.f - acc0[r];
            if (j0 + 16 < NP - 1) S[(4 * g4 + r) * NP + j0 + 16 + l15] = 1.f - acc1[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
    rank_strip(a, S, NP, my_q, row0, o0, n, lane);
}

// Per pair (one wave): Hits@1..5 counts, sum of reciprocal ranks, and SGAR for the modes '2', '50', '100'
// (utils/alignment.py:13-25,27-57): the anchors' top-1 predictions are ordered by ascending distance (stable), and a mode is
// satisfied when all of the first 2 / first half / all of them are correct.
__global__ void pair_metrics_kernel(const int* __restrict__ rank, const int* __restrict__ top1, const float* __restrict__ top1_sim,
                                    int ldk, const int* __restrict__ q_tgt, const int* __restrict__ pair_off,
                                    const int* __restrict__ pair_q_off, int B, float* __restrict__ out /* [B][12] */) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int b = blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += gridDim.x * wpb) {
        const int q0 = pair_q_off[b], na = pair_q_off[b + 1] - q0, o0 = pair_off[b];
        int hits[5] = {0, 0, 0, 0, 0};
        float rr = 0.f;
        int bad2 = 0, bad50 = 0, bad100 = 0;
        for (int i = lane; i < na; i += 64) {
            const int r = rank[q0 + i];
#pragma unroll
            for (int k = 0; k < 5; ++k) hits[k] += (r >= 1 && r <= k + 1) ? 1 : 0;
            rr += r >= 1 ? 1.f / (float)r : 0.f;
            // position of this anchor's top-1 distance among the pair's anchors (stable ascending)
            const float s = top1_sim[(size_t)(q0 + i) * ldk];
            int pos = 0;
            for (int j = 0; j < na; ++j) {
                const float sj = top1_sim[(size_t)(q0 + j) * ldk];
                pos += (sj < s || (sj == s && j < i)) ? 1 : 0;
            }
            const bool wrong = top1[(size_t)(q0 + i) * ldk] != q_tgt[q0 + i] - o0;
            if (wrong) { bad100 = 1; if (pos < 2) bad2 = 1; if (pos < na / 2) bad50 = 1; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int k = 0; k < 5; ++k) hits[k] += __shfl_xor(hits[k], o, 64);
            rr += __shfl_xor(rr, o, 64);
            bad2 |= __shfl_xor(bad2, o, 64); bad50 |= __shfl_xor(bad50, o, 64); bad100 |= __shfl_xor(bad100, o, 64);
        }
        if (lane == 0) {
            float* o = out + (size_t)b * 12;
#pragma unroll
            for (int k = 0; k < 5; ++k) o[k] = (float)hits[k];
            o[5] = (float)na; o[6] = rr;
            o[7] = bad2 ? 0.f : 1.f; o[8] = bad50 ? 0.f : 1.f; o[9] = bad100 ? 0.f : 1.f;
            o[10] = 0.f; o[11] = 0.f;
        }
    }
}

}  // namespace
extern "C" size_t sga_simrank_workspace_bytes_f16(int T, int D);
namespace {

template <bool F16>
int simrank_launch(const float* E, int T, int D, const int32_t* pair_off, const int32_t* blk_pair, const int32_t* blk_row, int n_blocks, int B,
                   int max_pair_objects, const int32_t* q_idx, const int32_t* q_tgt, int Q, int K, int32_t* rank, int32_t* topk_idx,
                   float* topk_sim, void* workspace, size_t workspace_bytes, hipStream_t s, const char* who) {
    SGA_CHECK_ARG(T >= 0 && D >= 1 && B >= 0 && Q >= 0, "%s: bad sizes", who);
    SGA_CHECK_ARG(K >= 0 && K <= SR_MAXK, "%s: K=%d outside [0,%d]", who, K, SR_MAXK);
    SGA_CHECK_ARG(max_pair_objects <= 64 * SR_MAXPER, "%s: a pair has %d objects; at most %d are supported", who, max_pair_objects, 64 * SR_MAXPER);
    if (Q == 0 || T == 0 || B == 0 || n_blocks == 0) return SGA_OK;
    SGA_CHECK_ARG(E && pair_off && blk_pair && blk_row && q_idx && rank && topk_idx && topk_sim, "%s: null pointer", who);
    if (!workspace || workspace_bytes < sga_simrank_workspace_bytes(T)) { sga_set_error("%s: workspace too small", who); return SGA_ERR_WORKSPACE; }
    float* inv = static_cast<float*>(workspace);
    int* obj_query = reinterpret_cast<int*>(inv + T);
    int g = (T + 3) / 4; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(g), dim3(256), 0, s, E, T, D, inv);
    hipLaunchKernelGGL(fill_int_kernel, dim3((T + 255) / 256 > 2048 ? 2048 : (T + 255) / 256), dim3(256), 0, s, obj_query, T, -1);
    hipLaunchKernelGGL(scatter_query_kernel, dim3((Q + 255) / 256 > 2048 ? 2048 : (Q + 255) / 256), dim3(256), 0, s, q_idx, Q, obj_query);
    SimArgs a{};
    a.E = E; a.inv = inv; a.D = D; a.B = B; a.pair_off = pair_off; a.blk_pair = blk_pair; a.blk_row = blk_row; a.obj_query = obj_query; a.q_tgt = q_tgt; a.K = K;
    a.rank = rank; a.topk_idx = topk_idx; a.topk_sim = topk_sim;
    a.npad_max = ((max_pair_objects + 15) / 16) * 16 + 1;
    const int nkq = (D + 15) / 16;
    auto launch = [&](auto kern, size_t lds) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(256), lds, s, a);
    };
    const size_t strip_b = (size_t)4 * 16 * a.npad_max * sizeof(float);
    auto tile_b = [](int kq) { return (size_t)16 * (kq * 16 + 4) * sizeof(float); };
    if (nkq <= 7) launch(simrank_staged_kernel<7, F16>, strip_b + tile_b(7));            // emb_dim 100 (single modality)
    else if (nkq <= 13) launch(simrank_staged_kernel<13, F16>, strip_b + tile_b(13));    // 200
    else if (nkq <= 20) launch(simrank_staged_kernel<20, F16>, strip_b + tile_b(20));    // 300 (P+S+R joint)
    else if (nkq <= 26) launch(simrank_staged_kernel<26, F16>, strip_b + tile_b(26));    // 400 (P+S+R+A joint)
    else if (F16) {                                                                       // wider, fp16: operands from the fp16 table
        const int Dp = (D + 31) / 32 * 32;
        SGA_CHECK_ARG(workspace_bytes >= sga_simrank_workspace_bytes_f16(T, D), "%s: workspace too small for the fp16 table (sga_simrank_workspace_bytes_f16)", who);
        _Float16* Eh = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(workspace) + (sga_simrank_workspace_bytes(T) + 255) / 256 * 256);
        hipLaunchKernelGGL(normalize_f16_kernel, dim3(g), dim3(256), 0, s, E, T, D, Dp, Eh);
        hipFuncSetAttribute(reinterpret_cast<const void*>(simrank_stream16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_b);
        hipLaunchKernelGGL(simrank_stream16_kernel, dim3(n_blocks), dim3(256), strip_b, s, a, Eh, Dp);
    } else launch(simrank_stream_kernel<F16>, strip_b);                                   // wider: both operands streamed per wave
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { sga_set_error("%s: launch failed: %s", who, hipGetErrorString(e_)); return SGA_ERR_HIP; }
    return SGA_OK;
}

}  // namespace

extern "C" size_t sga_simrank_workspace_bytes(int T) { return (sizeof(float) + sizeof(int)) * (size_t)(T > 0 ? T : 1); }
// f16 != 0 and D > 416: + the normalised fp16 copy of the table
extern "C" size_t sga_simrank_workspace_bytes_f16(int T, int D) {
    const size_t base = (sga_simrank_workspace_bytes(T) + 255) / 256 * 256;
    return D > 416 ? base + sizeof(_Float16) * (size_t)(T > 0 ? T : 1) * (size_t)((D + 31) / 32 * 32) : base;
}

extern "C" int sga_simrank(const float* E, int T, int D, const int32_t* pair_off, const int32_t* blk_pair, const int32_t* blk_row, int n_blocks, int B,
                           int max_pair_objects, const int32_t* q_idx, const int32_t* q_tgt, int Q, int K, int32_t* rank,
                           int32_t* topk_idx, float* topk_sim, int f16, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (f16) return simrank_launch<true>(E, T, D, pair_off, blk_pair, blk_row, n_blocks, B, max_pair_objects, q_idx, q_tgt, Q, K, rank, topk_idx,
                                         topk_sim, workspace, workspace_bytes, s, "sga_simrank(f16)");
    return simrank_launch<false>(E, T, D, pair_off, blk_pair, blk_row, n_blocks, B, max_pair_objects, q_idx, q_tgt, Q, K, rank, topk_idx, topk_sim,
                                 workspace, workspace_bytes, s, "sga_simrank");
}

extern "C" int sga_pair_metrics(const int32_t* rank, const int32_t* topk_idx, const float* topk_sim, int K, const int32_t* q_tgt,
                                const int32_t* pair_off, const int32_t* pair_q_off, int B, float* out, void* stream) {
    SGA_CHECK_ARG(B >= 0 && K >= 1, "sga_pair_metrics: bad sizes (K >= 1: the top-1 prediction is needed)");
    if (B == 0) return SGA_OK;
    SGA_CHECK_ARG(rank && topk_idx && topk_sim && q_tgt && pair_off && pair_q_off && out, "sga_pair_metrics: null pointer");
    int g = (B + 3) / 4; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pair_metrics_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), rank, topk_idx, topk_sim, K, q_tgt,
                       pair_off, pair_q_off, B, out);
    SGA_CHECK_LAUNCH("sga_pair_metrics");
    return SGA_OK;
}
