// GAT structure encoder: attention softmax-aggregate over each scene graph's edge list, fwd + bwd.
//
// Replaces torch_geometric.nn.GATConv (PyG 2.2.0; un-vendored dependency, req.yml:259) as it is used by
// reference src/aligner/networks/gat.py:36-37,44 (GATConv(in, 128, heads=2), defaults concat=True,
// negative_slope=0.2, add_self_loops=True, bias=True) driven per graph from sg_aligner.py:86-110:
//     h = x W^T ;  a_s[j] = <h[j], att_src>, a_d[i] = <h[i], att_dst>           (per head)
//     edges: input self loops removed, one self loop added per node, duplicates keep multiplicity
//     e_ij = leaky_relu(a_s[j] + a_d[i], 0.2);  alpha_ij = softmax over incoming j (max-subtracted, +1e-16)
//     out[i] = sum_j alpha_ij h[j] + bias
// The dense projections (x W^T and their gradients) run on the generic MFMA GEMM over ALL nodes of the
// batch at once; this file holds the per-graph part.  One workgroup per (graph, head): the head's
// projected features [N,128] live in LDS (row stride 129 -> conflict-free both along and across rows),
// the edge list is scattered once into an LDS multiplicity matrix cnt[i][j] (8-bit, LDS atomics), and
// the per-target softmax is a wavefront job: lanes span the sources j, max / sum by wave shuffles,
// alpha broadcast by v_readlane for the aggregate.  All 2B graphs of a batch go in ONE launch
// (the reference issues 2B sequential GATConv calls).  Graph traffic is byte-bound: 16 B/edge (int64
// pairs, read once per head) + 512 B/node/head in, 512 B/node/head out.
#include "sga_common.h"

namespace {

constexpr int GAT_C = 128;            // channels per head (hidden_units[1:], sg_aligner.py:38)
constexpr int GAT_H = 2;              // heads
constexpr int GAT_HS = GAT_C + 1;     // LDS row stride
constexpr int GAT_MAXN = 128;         // nodes per graph supported by the LDS-resident kernel
constexpr int GAT_THREADS = 256;
constexpr int GAT_TB = 8;             // targets (fwd) / sources (bwd) a wave processes together: LDS reads per FMA / 4
constexpr float GAT_SLOPE = 0.2f;

struct GatLds {
    float* hs;      // [N][129]
    float* dos;     // [N][129]   (bwd only)
    float* as;      // [128]
    float* ad;      // [128]
    float* mx;      // [128]  row max     (bwd)
    float* den;     // [128]  row denom   (bwd)
    float* das;     // [128]  d a_s       (bwd)
    float* dad;     // [128]  d a_d       (bwd)
    unsigned* cnt;  // [N][npad/4] packed u8 multiplicities
};

__device__ __forceinline__ GatLds carve(float* base, int nmax, bool bwd) {
    GatLds l;
    float* p = base;
    l.hs = p; p += nmax * GAT_HS;
    l.dos = p; if (bwd) p += nmax * GAT_HS;
    l.as = p; p += GAT_MAXN;
    l.ad = p; p += GAT_MAXN;
    l.mx = p; p += GAT_MAXN;
    l.den = p; p += GAT_MAXN;
    l.das = p; p += GAT_MAXN;
    l.dad = p; p += GAT_MAXN;
    l.cnt = reinterpret_cast<unsigned*>(p);
    return l;
}

__host__ __device__ inline size_t gat_lds_bytes(int nmax, bool bwd) {
    const int npad = (nmax + 3) & ~3;
    return sizeof(float) * ((size_t)nmax * GAT_HS * (bwd ? 2 : 1) + 6 * GAT_MAXN) + (size_t)nmax * npad;
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : GAT_SLOPE * v; }

// common prologue: load this head's features, attention logits' node parts, multiplicity matrix
__device__ __forceinline__ void gat_prologue(const GatLds& l, const float* __restrict__ H, const float* __restrict__ att_s,
                                             const float* __restrict__ att_d, const long long* __restrict__ edges,
                                             int n0, int N, int e0, int E, int hd, int npad) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < N * GAT_C; e += GAT_THREADS) {
        const int j = e >> 7, c = e & 127;
        l.hs[j * GAT_HS + c] = H[(size_t)(n0 + j) * (GAT_H * GAT_C) + hd * GAT_C + c];
    }
    for (int e = tid; e < N * (npad >> 2); e += GAT_THREADS) l.cnt[e] = 0u;
    __syncthreads();
    const float s0 = att_s[hd * GAT_C + lane], s1 = att_s[hd * GAT_C + 64 + lane];
    const float d0 = att_d[hd * GAT_C + lane], d1 = att_d[hd * GAT_C + 64 + lane];
    for (int j = wave; j < N; j += GAT_THREADS / 64) {
        const float h0 = l.hs[j * GAT_HS + lane], h1 = l.hs[j * GAT_HS + 64 + lane];
        const float vs = wave_sum(h0 * s0 + h1 * s1), vd = wave_sum(h0 * d0 + h1 * d1);
        if (lane == 0) { l.as[j] = vs; l.ad[j] = vd; }
    }
    // edge list -> multiplicities (self loops dropped, out-of-range ids ignored, counts saturate at 255)
    for (int e = tid; e < E; e += GAT_THREADS) {
        const long long sj = edges[(size_t)(e0 + e) * 2 + 0], di = edges[(size_t)(e0 + e) * 2 + 1];
        if (sj != di && sj >= 0 && sj < N && di >= 0 && di < N) {
            const int idx = (int)di * npad + (int)sj;
            const unsigned sh = 8u * (idx & 3);
            const unsigned old = atomicAdd(&l.cnt[idx >> 2], 1u << sh);
            if (((old >> sh) & 255u) == 255u) atomicSub(&l.cnt[idx >> 2], 1u << sh);
        }
    }
    __syncthreads();
    unsigned char* cb = reinterpret_cast<unsigned char*>(l.cnt);
    for (int i = tid; i < N; i += GAT_THREADS) cb[i * npad + i] = 1;       // exactly one self loop per node
    __syncthreads();
}

// softmax row i for sources j = lane and lane + 64: returns alpha (a0, a1), pre-activations, max, denom
__device__ __forceinline__ void softmax_row(const GatLds& l, int i, int N, int npad, int lane, float& a0, float& a1,
                                            float& pre0, float& pre1, float& m, float& den) {
    const unsigned char* cb = reinterpret_cast<const unsigned char*>(l.cnt);
    const float adi = l.ad[i];
    const int j0 = lane, j1 = lane + 64;
    const float c0 = j0 < N ? (float)cb[i * npad + j0] : 0.f;
    const float c1 = j1 < N ? (float)cb[i * npad + j1] : 0.f;
    pre0 = j0 < N ? l.as[j0] + adi : 0.f;
    pre1 = j1 < N ? l.as[j1] + adi : 0.f;
    const float e0 = lrelu(pre0), e1 = lrelu(pre1);
    m = wave_max(fmaxf(c0 > 0.f ? e0 : -INFINITY, c1 > 0.f ? e1 : -INFINITY));
    const float p0 = c0 > 0.f ? c0 * __expf(e0 - m) : 0.f;
    const float p1 = c1 > 0.f ? c1 * __expf(e1 - m) : 0.f;
    den = wave_sum(p0 + p1) + 1e-16f;
    a0 = p0 / den;
    a1 = p1 / den;
}

__global__ __launch_bounds__(GAT_THREADS) void gat_attn_fwd_kernel(
    const float* __restrict__ H, const float* __restrict__ att_s, const float* __restrict__ att_d,
    const float* __restrict__ bias, const long long* __restrict__ edges, const int* __restrict__ node_off,
    const int* __restrict__ edge_off, float* __restrict__ out, int nmax) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const GatLds l = carve(lds_raw, nmax, false);
    const int g = blockIdx.x, hd = blockIdx.y;
    const int n0 = node_off[g], N = node_off[g + 1] - n0, e0 = edge_off[g], E = edge_off[g + 1] - e0;
    const int npad = (nmax + 3) & ~3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (N <= 0) return;
    gat_prologue(l, H, att_s, att_d, edges, n0, N, e0, E, hd, npad);
    const float b0 = bias[hd * GAT_C + lane], b1 = bias[hd * GAT_C + 64 + lane];
    // GAT_TB targets per wave at a time: every h[j] row read from LDS feeds GAT_TB aggregates (the loop is LDS bound)
    for (int ib = wave * GAT_TB; ib < N; ib += (GAT_THREADS / 64) * GAT_TB) {
        float a0[GAT_TB], a1[GAT_TB], acc0[GAT_TB], acc1[GAT_TB];
#pragma unroll
        for (int u = 0; u < GAT_TB; ++u) {
            float pre0, pre1, m, den;
            a0[u] = 0.f; a1[u] = 0.f; acc0[u] = 0.f; acc1[u] = 0.f;
            if (ib + u < N) softmax_row(l, ib + u, N, npad, lane, a0[u], a1[u], pre0, pre1, m, den);
        }
        for (int j = 0; j < N; ++j) {
            const float h0 = l.hs[j * GAT_HS + lane], h1 = l.hs[j * GAT_HS + 64 + lane];
#pragma unroll
            for (int u = 0; u < GAT_TB; ++u) {
                const float a = j < 64 ? __shfl(a0[u], j, 64) : __shfl(a1[u], j - 64, 64);
                acc0[u] = fmaf(a, h0, acc0[u]);
                acc1[u] = fmaf(a, h1, acc1[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < GAT_TB; ++u)
            if (ib + u < N) {
                float* o = out + (size_t)(n0 + ib + u) * (GAT_H * GAT_C) + hd * GAT_C;
                o[lane] = acc0[u] + b0;
                o[64 + lane] = acc1[u] + b1;
            }
    }
}

__global__ __launch_bounds__(GAT_THREADS) void gat_attn_bwd_kernel(
    const float* __restrict__ H, const float* __restrict__ dO, const float* __restrict__ att_s,
    const float* __restrict__ att_d, const long long* __restrict__ edges, const int* __restrict__ node_off,
    const int* __restrict__ edge_off, float* __restrict__ dH, float* __restrict__ d_att_s,
    float* __restrict__ d_att_d, int nmax) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const GatLds l = carve(lds_raw, nmax, true);
    const int g = blockIdx.x, hd = blockIdx.y;
    const int n0 = node_off[g], N = node_off[g + 1] - n0, e0 = edge_off[g], E = edge_off[g + 1] - e0;
    const int npad = (nmax + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (N <= 0) return;
    for (int e = tid; e < N * GAT_C; e += GAT_THREADS) {
        const int j = e >> 7, c = e & 127;
        l.dos[j * GAT_HS + c] = dO[(size_t)(n0 + j) * (GAT_H * GAT_C) + hd * GAT_C + c];
    }
    for (int j = tid; j < GAT_MAXN; j += GAT_THREADS) l.das[j] = 0.f;
    gat_prologue(l, H, att_s, att_d, edges, n0, N, e0, E, hd, npad);

    // ---- phase 1: per target row i -> d a_d[i], partial d a_s[j], row max / denom
    float das0 = 0.f, das1 = 0.f;
    const int j0 = lane < N ? lane : 0, j1 = lane + 64 < N ? lane + 64 : 0;
    for (int ib = wave * GAT_TB; ib < N; ib += (GAT_THREADS / 64) * GAT_TB) {
        float a0[GAT_TB], a1[GAT_TB], pre0[GAT_TB], pre1[GAT_TB], m[GAT_TB], den[GAT_TB], da0[GAT_TB], da1[GAT_TB];
#pragma unroll
        for (int u = 0; u < GAT_TB; ++u) {
            a0[u] = 0.f; a1[u] = 0.f; pre0[u] = 0.f; pre1[u] = 0.f; m[u] = 0.f; den[u] = 1.f; da0[u] = 0.f; da1[u] = 0.f;
            if (ib + u < N) softmax_row(l, ib + u, N, npad, lane, a0[u], a1[u], pre0[u], pre1[u], m[u], den[u]);
        }
        // d alpha_ij = <dO[i], h[j]> for GAT_TB targets i at once: the two h reads per channel are shared
        for (int c = 0; c < GAT_C; ++c) {
            const float h0 = l.hs[j0 * GAT_HS + c], h1 = l.hs[j1 * GAT_HS + c];
#pragma unroll
            for (int u = 0; u < GAT_TB; ++u) {
                const float d = l.dos[min(ib + u, N - 1) * GAT_HS + c];
                da0[u] = fmaf(d, h0, da0[u]);
                da1[u] = fmaf(d, h1, da1[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < GAT_TB; ++u) {
            if (ib + u >= N) continue;
            const float s = wave_sum(a0[u] * da0[u] + a1[u] * da1[u]);
            const float ds0 = a0[u] * (da0[u] - s) * (pre0[u] > 0.f ? 1.f : GAT_SLOPE);
            const float ds1 = a1[u] * (da1[u] - s) * (pre1[u] > 0.f ? 1.f : GAT_SLOPE);
            das0 += ds0;
            das1 += ds1;
            const float dd = wave_sum(ds0 + ds1);
            if (lane == 0) { l.dad[ib + u] = dd; l.mx[ib + u] = m[u]; l.den[ib + u] = den[u]; }
        }
    }
    if (lane < N) atomicAdd(&l.das[lane], das0);
    if (lane + 64 < N) atomicAdd(&l.das[lane + 64], das1);
    __syncthreads();

    // ---- phase 2: per source j -> dH[j] = sum_i alpha_ij dO[i] + d a_s[j] att_s + d a_d[j] att_d ; d att
    const unsigned char* cb = reinterpret_cast<const unsigned char*>(l.cnt);
    const float s0 = att_s[hd * GAT_C + lane], s1 = att_s[hd * GAT_C + 64 + lane];
    const float d0 = att_d[hd * GAT_C + lane], d1 = att_d[hd * GAT_C + 64 + lane];
    float gs0 = 0.f, gs1 = 0.f, gd0 = 0.f, gd1 = 0.f;
    const int i0 = lane, i1 = lane + 64;
    for (int jb = wave * GAT_TB; jb < N; jb += (GAT_THREADS / 64) * GAT_TB) {
        // alpha_ij for i = lane, lane + 64 (vectorised over targets) of GAT_TB sources j, then broadcast per i
        float al0[GAT_TB], al1[GAT_TB], acc0[GAT_TB], acc1[GAT_TB];
#pragma unroll
        for (int u = 0; u < GAT_TB; ++u) {
            al0[u] = 0.f; al1[u] = 0.f; acc0[u] = 0.f; acc1[u] = 0.f;
            const int j = jb + u;
            if (j >= N) continue;
            const float asj = l.as[j];
            if (i0 < N) { const float c = (float)cb[i0 * npad + j]; if (c > 0.f) al0[u] = c * __expf(lrelu(asj + l.ad[i0]) - l.mx[i0]) / l.den[i0]; }
            if (i1 < N) { const float c = (float)cb[i1 * npad + j]; if (c > 0.f) al1[u] = c * __expf(lrelu(asj + l.ad[i1]) - l.mx[i1]) / l.den[i1]; }
        }
        for (int i = 0; i < N; ++i) {
            const float g0 = l.dos[i * GAT_HS + lane], g1 = l.dos[i * GAT_HS + 64 + lane];
#pragma unroll
            for (int u = 0; u < GAT_TB; ++u) {
                const float a = i < 64 ? __shfl(al0[u], i, 64) : __shfl(al1[u], i - 64, 64);
                acc0[u] = fmaf(a, g0, acc0[u]);
                acc1[u] = fmaf(a, g1, acc1[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < GAT_TB; ++u) {
            const int j = jb + u;
            if (j >= N) continue;
            const float dasj = l.das[j], dadj = l.dad[j];
            float* o = dH + (size_t)(n0 + j) * (GAT_H * GAT_C) + hd * GAT_C;
            o[lane] = acc0[u] + dasj * s0 + dadj * d0;
            o[64 + lane] = acc1[u] + dasj * s1 + dadj * d1;
            const float h0 = l.hs[j * GAT_HS + lane], h1 = l.hs[j * GAT_HS + 64 + lane];
            gs0 = fmaf(dasj, h0, gs0); gs1 = fmaf(dasj, h1, gs1);
            gd0 = fmaf(dadj, h0, gd0); gd1 = fmaf(dadj, h1, gd1);
        }
    }
    atomicAdd(d_att_s + hd * GAT_C + lane, gs0);
    atomicAdd(d_att_s + hd * GAT_C + 64 + lane, gs1);
    atomicAdd(d_att_d + hd * GAT_C + lane, gd0);
    atomicAdd(d_att_d + hd * GAT_C + 64 + lane, gd1);
}

// ELU between the layers (gat.py:45-46) and its derivative
__global__ void elu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v > 0.f ? v : expm1f(v);
    }
}
__global__ void elu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        gx[i] = v > 0.f ? gy[i] : gy[i] * expf(v);
    }
}

int check_common(int G, int nmax, const char* who) {
    if (G < 0 || nmax < 0) { sga_set_error("%s: negative size", who); return SGA_ERR_ARG; }
    if (nmax > GAT_MAXN) {
        sga_set_error("%s: a graph has %d nodes; the LDS-resident GAT kernel supports at most %d per graph", who, nmax, GAT_MAXN);
        return SGA_ERR_ARG;
    }
    return SGA_OK;
}

}  // namespace

extern "C" int sga_gat_attn_fwd(const float* H, const float* att_src, const float* att_dst, const float* bias,
                                const int64_t* edges, const int32_t* node_off, const int32_t* edge_off, int G,
                                int nmax, float* out, void* stream) {
    int rc = check_common(G, nmax, "sga_gat_attn_fwd");
    if (rc) return rc;
    SGA_CHECK_ARG(H && att_src && att_dst && bias && node_off && edge_off && out, "sga_gat_attn_fwd: null pointer");
    if (G == 0 || nmax == 0) return SGA_OK;
    const size_t lds = gat_lds_bytes(nmax, false);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gat_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gat_attn_fwd_kernel, dim3(G, GAT_H), dim3(GAT_THREADS), lds, static_cast<hipStream_t>(stream), H, att_src,
                       att_dst, bias, reinterpret_cast<const long long*>(edges), node_off, edge_off, out, nmax);
    SGA_CHECK_LAUNCH("sga_gat_attn_fwd");
    return SGA_OK;
}

extern "C" int sga_gat_attn_bwd(const float* H, const float* dO, const float* att_src, const float* att_dst,
                                const int64_t* edges, const int32_t* node_off, const int32_t* edge_off, int G,
                                int nmax, float* dH, float* d_att_src, float* d_att_dst, void* stream) {
    int rc = check_common(G, nmax, "sga_gat_attn_bwd");
    if (rc) return rc;
    SGA_CHECK_ARG(H && dO && att_src && att_dst && node_off && edge_off && dH && d_att_src && d_att_dst, "sga_gat_attn_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipMemsetAsync(d_att_src, 0, GAT_H * GAT_C * sizeof(float), s);
    hipMemsetAsync(d_att_dst, 0, GAT_H * GAT_C * sizeof(float), s);
    if (G == 0 || nmax == 0) return SGA_OK;
    const size_t lds = gat_lds_bytes(nmax, true);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gat_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gat_attn_bwd_kernel, dim3(G, GAT_H), dim3(GAT_THREADS), lds, s, H, dO, att_src, att_dst,
                       reinterpret_cast<const long long*>(edges), node_off, edge_off, dH, d_att_src, d_att_dst, nmax);
    SGA_CHECK_LAUNCH("sga_gat_attn_bwd");
    return SGA_OK;
}

extern "C" int sga_elu_fwd(const float* x, float* y, size_t n, void* stream) {
    SGA_CHECK_ARG(x && y, "sga_elu_fwd: null pointer");
    if (n == 0) return SGA_OK;
    size_t g = (n + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(elu_fwd_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n);
    SGA_CHECK_LAUNCH("sga_elu_fwd");
    return SGA_OK;
}

extern "C" int sga_elu_bwd(const float* x, const float* gy, float* gx, size_t n, void* stream) {
    SGA_CHECK_ARG(x && gy && gx, "sga_elu_bwd: null pointer");
    if (n == 0) return SGA_OK;
    size_t g = (n + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), x, gy, gx, n);
    SGA_CHECK_LAUNCH("sga_elu_bwd");
    return SGA_OK;
}
