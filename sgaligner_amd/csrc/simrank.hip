// Per-pair node similarity + ranking for the alignment metrics.
//
// Replaces the eval_step block of reference src/inference/sgaligner/inference_align_reg.py:125-128
// (emb /= ||emb||; sim = 1 - emb emb^T; rank_list = argsort(sim, dim=1)) fused with what
// utils/alignment.py:3-25,27-41,59-70 then read from the rank list: for a query object q of a pair,
// the 1-based rank of its ground-truth match among all OTHER objects of the pair (self removed by
// value, alignment.py:7,18) and its K nearest other objects with their distances.  The full n x n
// sort and the seven device->host copies of the rank list (alignment.py:4,14,29) disappear.
// Ties: broken by object index (a stable ascending sort); the reference's sort is unstable there.
// One wave per query: lanes span the pair's objects, the rank is a ballot/popcount-style wave sum,
// the top-K a K-step wave arg-min.  Byte-bound: each query streams its pair's table once from L2.
#include "sga_common.h"

namespace {

constexpr int SR_MAXK = 8;
constexpr int SR_MAXPER = 8;          // objects per lane -> up to 512 objects per pair

__global__ void row_inv_norm_kernel(const float* __restrict__ E, int T, int D, float* __restrict__ inv) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < T; t += gridDim.x * wpb) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = E[(size_t)t * D + d]; ss += v * v; }
        ss = wave_sum(ss);
        if (lane == 0) inv[t] = 1.f / sqrtf(ss);          // no eps, as in the reference (:126)
    }
}

__global__ void simrank_kernel(const float* __restrict__ E, const float* __restrict__ inv, int D,
                               const int* __restrict__ pair_off, const int* __restrict__ q_pair,
                               const int* __restrict__ q_idx, const int* __restrict__ q_tgt, int Q, int K,
                               int* __restrict__ rank, int* __restrict__ topk_idx, float* __restrict__ topk_sim) {
    extern __shared__ float qrow[];                        // [waves][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    float* myq = qrow + (size_t)wave * D;
    for (int q = blockIdx.x * wpb + wave; q < Q; q += gridDim.x * wpb) {
        const int b = q_pair[q];
        const int o0 = pair_off[b], n = pair_off[b + 1] - o0;
        const int a = q_idx[q];                            // global object index of the query
        const float ia = inv[a];
        for (int d = lane; d < D; d += 64) myq[d] = E[(size_t)a * D + d] * ia;
        __builtin_amdgcn_wave_barrier();
        float sim[SR_MAXPER];
#pragma unroll
        for (int u = 0; u < SR_MAXPER; ++u) {
            const int j = lane + 64 * u;
            float dot = 0.f;
            if (j < n) {
                const float* r = E + (size_t)(o0 + j) * D;
                for (int d = 0; d < D; ++d) dot = fmaf(myq[d], r[d], dot);
                dot *= inv[o0 + j];
            }
            sim[u] = j < n ? 1.f - dot : INFINITY;
        }
        const int al = a - o0;                             // pair-local self index
        // ---- rank of the target among the others
        const int tg = q_tgt ? q_tgt[q] - o0 : -1;
        if (tg >= 0 && tg < n) {
            float st = 0.f;
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) { const float v = __shfl(sim[u], tg & 63, 64); if ((tg >> 6) == u) st = v; }
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) {
                const int j = lane + 64 * u;
                if (j < n && j != al && j != tg && (sim[u] < st || (sim[u] == st && j < tg))) ++cnt;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0) rank[q] = cnt + 1;
        } else if (lane == 0 && rank) {
            rank[q] = -1;
        }
        // ---- K nearest others (ascending distance, index tie-break)
#pragma unroll
        for (int u = 0; u < SR_MAXPER; ++u) if (lane + 64 * u == al) sim[u] = INFINITY;
        for (int k = 0; k < K; ++k) {
            float bv = INFINITY; int bj = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) { const int j = lane + 64 * u; if (sim[u] < bv) { bv = sim[u]; bj = j; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oj = __shfl_xor(bj, o, 64);
                if (ov < bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
            }
            if (lane == 0) { topk_idx[(size_t)q * K + k] = bv < INFINITY ? bj : -1; topk_sim[(size_t)q * K + k] = bv; }
#pragma unroll
            for (int u = 0; u < SR_MAXPER; ++u) if (lane + 64 * u == bj) sim[u] = INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" size_t sga_simrank_workspace_bytes(int T) { return sizeof(float) * (size_t)(T > 0 ? T : 1); }

extern "C" int sga_simrank(const float* E, int T, int D, const int32_t* pair_off, int B, int max_pair_objects,
                           const int32_t* q_pair, const int32_t* q_idx, const int32_t* q_tgt, int Q, int K,
                           int32_t* rank, int32_t* topk_idx, float* topk_sim, void* workspace, size_t workspace_bytes,
                           void* stream) {
    SGA_CHECK_ARG(E && pair_off && q_pair && q_idx && rank && topk_idx && topk_sim && T >= 0 && D >= 1 && B >= 0 && Q >= 0,
                  "sga_simrank: bad argument");
    SGA_CHECK_ARG(K >= 0 && K <= SR_MAXK, "sga_simrank: K=%d outside [0,%d]", K, SR_MAXK);
    SGA_CHECK_ARG(max_pair_objects <= 64 * SR_MAXPER, "sga_simrank: a pair has %d objects; at most %d are supported", max_pair_objects, 64 * SR_MAXPER);
    if (!workspace || workspace_bytes < sga_simrank_workspace_bytes(T)) { sga_set_error("sga_simrank: workspace too small"); return SGA_ERR_WORKSPACE; }
    if (Q == 0 || T == 0) return SGA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* inv = static_cast<float*>(workspace);
    int g = (T + 3) / 4; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(g), dim3(256), 0, s, E, T, D, inv);
    int gq = (Q + 3) / 4; if (gq > 8192) gq = 8192;
    hipLaunchKernelGGL(simrank_kernel, dim3(gq), dim3(256), 4 * (size_t)D * sizeof(float), s, E, inv, D, pair_off, q_pair, q_idx,
                       q_tgt, Q, K, rank, topk_idx, topk_sim);
    SGA_CHECK_LAUNCH("sga_simrank");
    return SGA_OK;
}
