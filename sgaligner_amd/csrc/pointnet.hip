// Per-object PointNet set-abstraction encoder, forward.
//
// Replaces reference src/aligner/networks/pointnet.py:120-175 (PointNetfeat.forward with
// global_feat=True, input_transform=False, feature_transform=False):
//     y[t,c] = max_p relu(W3 relu(W2 relu(W1 x[t,p] + b1) + b2) + b3)[c]
// (the three BatchNorm calls at :141-142,154-155,158-159 discard their result and do not enter y).
//
// CDNA4 design (exact fp32, v_mfma_f32_32x32x2_f32):
//   * one WAVE owns one object and walks its points 32 at a time; the whole 3->64->128->C3 chain for
//     a 32-point tile stays in that wave's registers:
//       layer 1  VALU: lane (h = lane>>5, pt = lane&31) computes the 32 of 64 channels its half feeds
//                to the MFMA K-steps (k = 8q + 4h + r);
//       layer 2  H2^T[ch, pt] = W2 * H1^T : A = W2 slice (LDS), B = H1 (registers).  The C layout
//                (lane = pt, reg r = channel (r&3)+8(r>>2)+4h) IS the A-operand layout of
//       layer 3  Z3[pt, ch]   = H2 * W3^T : A = H2 (registers, straight from layer 2's accumulators),
//                B = W3 slice (LDS).  C layout: lane = channel, regs = points -> the max-pool over
//                points is an in-lane max over 16 registers + one cross-half exchange per object.
//     No activation ever touches LDS or HBM; bias add and ReLU commute with the max where needed.
//   * W2 (32 KiB) and W3 (128 KiB) are re-laid out once per workgroup into LDS in exact per-lane
//     operand order ([block][k-group][lane][4]) so every ds_read_b128 is lane-linear (conflict free)
//     and feeds 4 MFMAs.  That fills the CU's 160 KiB LDS: one persistent 8-wave workgroup per CU.
//   * HBM traffic is the points once (12 B/point) + T*C3*(4+4) B out: the kernel is MFMA-bound
//     (82 304 FLOP/point at C3 = 256).
#include <stdlib.h>
#include <type_traits>
#include "sga_common.h"

namespace {

#ifndef PN_BN_SB
#define PN_BN_SB 0
#endif
// The BN partial sums: every slot belongs to ONE wave, so the adds need no wider scope than the wavefront.  (Measured: the scope changes
// neither the time nor the HBM-side write traffic -- 144 GB per configs[2] launch at 16 atomics per tile: fp64 atomics are carried out
// memory-side on this part whatever their scope.  PN_BN_ATOMIC_DEVICE builds the device-scope form.)
#ifdef PN_BN_ATOMIC_DEVICE
#define PN_BN_ATOMIC_ADD(p, v) unsafeAtomicAdd((p), (v))
#else
#define PN_BN_ATOMIC_ADD(p, v) (void)__hip_atomic_fetch_add((__attribute__((address_space(1))) double*)(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
#endif
constexpr int PN_WAVES = 8;
constexpr size_t PN_P3_LG_BYTES = (size_t)(1024 + 8 * 512) * 16;      // the l planes of W2 and W3 (C3 = 256) in operand order: pointnet_p3_lplanes_kernel
constexpr int PN_THREADS = PN_WAVES * 64;

// SPLIT (few objects: the reference's own batch sizes, single-pair inference): a workgroup takes ONE object at a time and its 8
// waves share the object's 32-point tiles (tile = wave, wave + 8, ...); each wave leaves its running (max, arg-max) in `part`
// [T][8][C3] (value, index) and pointnet_combine_kernel folds the 8 partials.  With one wave per object, T = 320 objects keep 40 of
// the 256 CUs busy for 16 tiles each (0.63 ms); split, every CU works and an object takes 2 tiles per wave.
//
// BN (training forward, the reference's side effect): the three BatchNorm calls of pointnet.py:141-142,154-155,158-159 discard their
// output but fold the batch statistics of the PRE-ReLU conv outputs over all T*P points into running_mean / running_var.  The sums are
// taken here, from the registers that hold those values anyway, into per-lane fp64 accumulators (33 + ... doubles) that a wave keeps over
// all its objects and writes ONCE to bn_part[wave][k][lane]; pointnet_bn_reduce_kernel folds them in a fixed order:
//   layer 1  z1 = W1 x + b1 is affine in x: only the 9 first / second moments of the points are summed (lane = point, half 0 counts);
//            mean and variance follow from W1, b1 in fp64 on the host side of the ABI (9 numbers instead of 128 sums);
//   layer 2  the pre-activations live with lane = point, register = channel -- their per-channel sums are sums over LANES.  Sixteen
//            MFMAs against identity slices (B[k, j] = [j == channel(k)]; exact: one product by 1, the rest zeros) transpose a 32-channel
//            block to the layer-3 layout (lane = channel, registers = points), where the sums are in-lane: +64 MFMAs per 32-point tile
//            on top of 640 (a cross-lane butterfly over 64 registers x 2 moments would be ~640 DPP adds, in-lane accumulators in the
//            lane = point layout 128 registers);
//   layer 3  lane = channel already (bias b3 is added after the max: it enters the mean on the host side, not the variance).
// Replicated tail points (P not a multiple of 32) are masked out.
template <int C3, bool WITH_ARGMAX, bool SPLIT = false, bool BN = false>
__global__ __launch_bounds__(PN_THREADS) void pointnet_fwd_kernel(
    const float* __restrict__ x,   // [T, P, 3]
    const float* __restrict__ w1,  // [64, 3]
    const float* __restrict__ b1,  // [64]
    const float* __restrict__ w2,  // [128, 64]
    const float* __restrict__ b2,  // [128]
    const float* __restrict__ w3,  // [C3, 128]
    const float* __restrict__ b3,  // [C3]
    float* __restrict__ y,         // [T, C3]
    int* __restrict__ argmax,      // [T, C3] or nullptr
    int T, int P, float2* __restrict__ part,               // part: SPLIT only, [T][PN_WAVES][C3] (value, index as float bits)
    double* __restrict__ bn_part = nullptr) {              // BN: [gridDim.x * PN_WAVES][PN_BN_SLOTS(C3)][64] per-lane partial sums
    constexpr int NB3 = C3 / 32;
    constexpr int NBN = BN ? 9 + 2 * 4 + 2 * NB3 : 1;
    // BN: an object's per-lane fp32 partial sums (24 registers) go to the wave's OWN fp64 slots by (uncontended, no-return) atomic adds when
    // the object ends; the host zeroes the slots before the launch.  (Slots 0..8, the point moments, are pointnet_xmoments_kernel's.)
    double* const bn_dst = BN ? bn_part + ((size_t)blockIdx.x * PN_WAVES + (threadIdx.x >> 6)) * NBN * 64 + (threadIdx.x & 63) : nullptr;
    // (the slot addresses are formed from an OPAQUE copy of the base where they are used: hoisted out of the tile loop they are 2 x 33 registers)
    auto bn_add = [&](int slot, float v) {
        if (BN) {
            double* d = bn_dst;
            asm volatile("" : "+v"(d));
            PN_BN_ATOMIC_ADD(d + slot * 64, (double)v);
        }
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w2s = lds;             // [4 cb][8 q][64 lane][4]        = 8192 floats
    float* w3s = lds + 8192;      // [NB3 cb][4 kb][4 g][64 lane][4] = C3*128 floats

    const int tid = threadIdx.x;
    // ---- stage weights into LDS in operand order (once per persistent workgroup)
    for (int d = tid; d < 2048; d += PN_THREADS) {          // W2: 2048 float4 slots
        const int ln = d & 63, q = (d >> 6) & 7, cb = d >> 9;
        const int row = cb * 32 + (ln & 31), k = 8 * q + 4 * (ln >> 5);
        *reinterpret_cast<f32x4*>(w2s + d * 4) = *reinterpret_cast<const f32x4*>(w2 + row * 64 + k);
    }
    for (int d = tid; d < C3 * 32; d += PN_THREADS) {       // W3: C3*128/4 float4 slots
        const int ln = d & 63, g = (d >> 6) & 3, kb = (d >> 8) & 3, cb = d >> 10;
        const int row = cb * 32 + (ln & 31), k = kb * 32 + 8 * g + 4 * (ln >> 5);
        *reinterpret_cast<f32x4*>(w3s + d * 4) = *reinterpret_cast<const f32x4*>(w3 + row * 128 + k);
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, pt = lane & 31;
    const int n_tiles = (P + 31) >> 5;

    const int n_obj = T;
    for (int it = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * PN_WAVES + wave; it < n_obj; it += SPLIT ? (int)gridDim.x : (int)gridDim.x * PN_WAVES) {
        const int t = it;
        const float* xt = x + (size_t)t * P * 3;
        float best[NB3];
        int bidx[NB3];
#pragma unroll
        for (int c = 0; c < NB3; ++c) { best[c] = -INFINITY; bidx[c] = 0; }
        float bacc[BN ? NBN - 9 : 1];                      // BN: this object's per-lane sums (<= 16 tiles x 16 values each), flushed when it ends
#pragma unroll
        for (int k = 0; k < (BN ? NBN - 9 : 1); ++k) bacc[k] = 0.f;

        // (the tile body in two compile-time forms: a tile that replicates the object's last point masks the copies out of the BN sums;
        //  a run-time test inside the folds splits the body into 12 conditional regions, which costs hipcc ~100 registers)
        auto tile_body = [&](int tile, auto tail_c) {
            constexpr bool tail = BN && decltype(tail_c)::value;
            const int p0 = tile * 32;
            // Opaque per-tile copies of the lane ids: the weight reads below are loop-invariant, and
            // without this LICM hoists ~900 registers of them out of the tile loop (-> scratch spills).
            int lane_o = lane, h_o = h;
            asm volatile("" : "+v"(lane_o), "+v"(h_o));
            const int pi = min(p0 + pt, P - 1);          // ragged tail: replicate the last point
            const float x0 = xt[pi * 3 + 0], x1 = xt[pi * 3 + 1], x2 = xt[pi * 3 + 2];

            auto bn_fold = [&](f32x16& v, int slot) {          // (v is dead afterwards: the tail tile zeroes its replicated rows in place)
                if (tail) {
                    const int lim = P - p0 - 4 * h_o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = mfma32_row(r, 0) < lim ? v[r] : 0.f;
                }
                float sm[4], sq[4];                             // four independent chains each (a 16-deep dependent chain is latency, not issue)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sm[r] = v[r]; sq[r] = v[r] * v[r]; }
#pragma unroll
                for (int r = 4; r < 16; ++r) { sm[r & 3] += v[r]; sq[r & 3] = fmaf(v[r], v[r], sq[r & 3]); }
                bacc[BN ? slot - 9 : 0] += (sm[0] + sm[1]) + (sm[2] + sm[3]); bacc[BN ? slot - 8 : 0] += (sq[0] + sq[1]) + (sq[2] + sq[3]);
            };

            // ---- layer 1 (VALU): channels k = 8q + 4h + r
            float h1[32];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = 8 * q + 4 * h_o;
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w1 + k * 3);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w1 + k * 3 + 4);
                const f32x4 wc = *reinterpret_cast<const f32x4*>(w1 + k * 3 + 8);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + k);
                h1[q * 4 + 0] = fmaxf(fmaf(wa[2], x2, fmaf(wa[1], x1, fmaf(wa[0], x0, bb[0]))), 0.f);
                h1[q * 4 + 1] = fmaxf(fmaf(wb[1], x2, fmaf(wb[0], x1, fmaf(wa[3], x0, bb[1]))), 0.f);
                h1[q * 4 + 2] = fmaxf(fmaf(wc[0], x2, fmaf(wb[3], x1, fmaf(wb[2], x0, bb[2]))), 0.f);
                h1[q * 4 + 3] = fmaxf(fmaf(wc[3], x2, fmaf(wc[2], x1, fmaf(wc[1], x0, bb[3]))), 0.f);
            }

            // ---- layer 2 (MFMA): H2^T = W2 * H1^T, accumulators start at the bias
            float h2[64];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + cb * 32 + 8 * g + 4 * h_o);
                    acc[g * 4 + 0] = bb[0]; acc[g * 4 + 1] = bb[1]; acc[g * 4 + 2] = bb[2]; acc[g * 4 + 3] = bb[3];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w2s + ((cb * 8 + q) * 64 + lane_o) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[r], h1[q * 4 + r], acc, 0, 0, 0);
                }
                if (BN) {                                   // Z2 block -> (lane = channel, register = point row) through identity slices
                    f32x16 tr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) tr[r] = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            tr = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[g * 4 + r], (lane_o & 31) == 8 * g + 4 * h_o + r ? 1.f : 0.f, tr, 0, 0, 0);
                    bn_fold(tr, 9 + 2 * cb);
                    __builtin_amdgcn_sched_barrier(PN_BN_SB);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) h2[cb * 16 + r] = fmaxf(acc[r], 0.f);
            }

            // ---- layer 3 (MFMA): Z3 = H2 * W3^T, running max over points
#pragma unroll
            for (int cb = 0; cb < NB3; ++cb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(w3s + (((cb * 4 + kb) * 4 + g) * 64 + lane_o) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h2[kb * 16 + g * 4 + r], wv[r], acc, 0, 0, 0);
                    }
                }
                if (WITH_ARGMAX) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {        // ascending point order, strict > keeps the first max
                        const bool gt = acc[r] > best[cb];
                        best[cb] = gt ? acc[r] : best[cb];
                        bidx[cb] = gt ? (p0 + mfma32_row(r, h)) : bidx[cb];
                    }
                } else {
                    float m = acc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                    best[cb] = fmaxf(best[cb], m);
                }
                if (BN) { bn_fold(acc, 9 + 8 + 2 * cb); __builtin_amdgcn_sched_barrier(PN_BN_SB); }   // behind the max: the tail tile zeroes rows in place; (barrier: hipcc otherwise interleaves the blocks' folds and spills)
            }
        };
        for (int tile = SPLIT ? wave : 0; tile < n_tiles; tile += SPLIT ? PN_WAVES : 1) {
            if (BN && tile * 32 + 32 > P) tile_body(tile, std::true_type{}); else tile_body(tile, std::false_type{});
        }

        if (BN) {
#pragma unroll
            for (int k = 0; k < NBN - 9; ++k) bn_add(9 + k, bacc[k]);
        }
        // ---- combine the two lane halves (they hold disjoint point rows), bias + ReLU, store
#pragma unroll
        for (int cb = 0; cb < NB3; ++cb) {
            const float ov = __shfl_xor(best[cb], 32, 64);
            float v = best[cb];
            int bi = bidx[cb];
            if (WITH_ARGMAX) {
                const int oi = __shfl_xor(bidx[cb], 32, 64);
                const bool take = (ov > v) || (ov == v && oi < bi);
                v = take ? ov : v;
                bi = take ? oi : bi;
                bi = min(bi, P - 1);
            } else {
                v = fmaxf(v, ov);
            }
            if (h == 0) {
                const int c = cb * 32 + pt;
                if (SPLIT) {
                    part[((size_t)t * PN_WAVES + wave) * C3 + c] = float2{v, __int_as_float(bi)};
                } else {
                    y[(size_t)t * C3 + c] = fmaxf(v + b3[c], 0.f);
                    if (WITH_ARGMAX) argmax[(size_t)t * C3 + c] = bi;
                }
            }
        }
    }
}

// First / second moments of the points (slots 0..8 of the BN partials: layer 1 is affine in x), same [wave][slot][lane] layout and the same
// grid as the forward: every lane sums a strided share of the T*P points in fp64 (fp64 products: exact).
__global__ __launch_bounds__(PN_THREADS) void pointnet_xmoments_kernel(const float* __restrict__ x, size_t n, int C3, double* __restrict__ bn_part) {
    const int nbn = 9 + 8 + 2 * (C3 / 32);
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * PN_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * PN_THREADS) {
        const double x0 = x[3 * i], x1 = x[3 * i + 1], x2 = x[3 * i + 2];
        a[0] += x0; a[1] += x1; a[2] += x2;
        a[3] += x0 * x0; a[4] += x0 * x1; a[5] += x0 * x2; a[6] += x1 * x1; a[7] += x1 * x2; a[8] += x2 * x2;
    }
    double* dst = bn_part + ((size_t)blockIdx.x * PN_WAVES + (threadIdx.x >> 6)) * nbn * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 9; ++k) dst[k * 64] = a[k];
}

// Fold the per-lane partial sums of the BN forward in a fixed order (deterministic).  out (doubles):
//   [0, 9)            sum x0, x1, x2, x0x0, x0x1, x0x2, x1x1, x1x2, x2x2 over the T*P points
//   [9, 137)          sum z2[c]          [137, 265)        sum z2[c]^2      (z2 = W2 relu(z1) + b2, bias included)
//   [265, 265 + C3)   sum (z3 - b3)[c]   [265 + C3, + C3)  sum (z3 - b3)[c]^2
// One 64-thread block per output.
__global__ __launch_bounds__(64) void pointnet_bn_reduce_kernel(const double* __restrict__ part, int nwaves, int C3, double* __restrict__ out) {
    const int o = blockIdx.x, nbn = 9 + 8 + 2 * (C3 / 32);
    int slot, l0, l1;                                      // which slot and which lanes of a wave's partials feed output o
    if (o < 9) { slot = o; l0 = 0; l1 = 64; }
    else if (o < 265) { const int c = (o - 9) & 127, sq = (o - 9) >> 7; slot = 9 + 2 * (c >> 5) + sq; l0 = c & 31; l1 = -1; }
    else { const int c = (o - 265) % C3, sq = (o - 265) / C3; slot = 17 + 2 * (c >> 5) + sq; l0 = c & 31; l1 = -1; }
    double acc = 0.0;
    for (int w = threadIdx.x; w < nwaves; w += 64) {
        const double* p = part + ((size_t)w * nbn + slot) * 64;
        if (l1 < 0) acc += p[l0] + p[l0 + 32];
        else for (int l = l0; l < l1; ++l) acc += p[l];
    }
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) out[o] = acc;
}

// fold the PN_WAVES partial (max, arg-max) pairs of the SPLIT forward: larger value wins, equal values -> smaller point index (the
// first maximum, as in the unsplit kernel); then bias + ReLU
__global__ void pointnet_combine_kernel(const float2* __restrict__ part, const float* __restrict__ b3, float* __restrict__ y,
                                        int* __restrict__ argmax, int T, int C3, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * C3) return;
    const int t = i / C3, c = i - t * C3;
    float v = -INFINITY;
    int bi = 0;
    for (int w = 0; w < PN_WAVES; ++w) {
        const float2 pv = part[((size_t)t * PN_WAVES + w) * C3 + c];
        const int oi = __float_as_int(pv.y);
        if (pv.x > v || (pv.x == v && oi < bi) || w == 0) { v = pv.x; bi = oi; }
    }
    y[i] = fmaxf(v + b3[c], 0.f);
    if (argmax) argmax[i] = min(bi, P - 1);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// -------------------------------------------------------------------------------------------------
// The forward with every fp32 operand as THREE exact bf16 terms (mode 4; the default arithmetic of the Python layer, 'bf16x6'):
// x = h + m + l, 8 + 8 + 8 significand bits and fp32's exponent range -- every fp32 value exactly, as in the loss sweeps (sweep3.hip) --
// a product = the six partial products h h' + (h m' + m h') + (m m' + h l' + l h') on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 is exact in
// fp32; the three dropped products are <= 2^-23 of the product, below the fp32 accumulation's own rounding): fp32 arithmetic on the
// reference's fp32 operands (pointnet.py:140-161) at 6 x 32 cycles per 16 k slots where v_mfma_f32_32x32x2_f32 needs 8 x 64.
// The five small partial products of an accumulation go to their OWN accumulator (the 16-bit MFMAs chop what falls ~7 bits below the
// result's last place toward minus infinity whatever the sign: tools/micro/mfma_round_probe.hip), added once at the end.
// Geometry: one wave per object, the 3 -> 64 -> 128 -> C3 chain of a 32-point tile in registers, layer 2's
// accumulators are layer 3's A operand.  LDS: W2's three planes (48 KiB) + the three planes of HALF of W3's output channels at C3 = 256
// (96 KiB; all of them below): a workgroup serves one channel half (blockIdx & 1), two workgroups share an object and both run
// layers 1-2 -- 2 x 96 + 384 = 576 MFMAs of 32 cycles per 32-point tile against 640 of 64.
// BN: the batch sums of the reference's BatchNorm side effect as in pointnet_fwd_kernel<.., BN>: layer 3 in-lane; layer 2 (lane = point)
// through four MFMAs per 32-channel block against a bf16 identity (two planes of z2 re-delivered with lane = channel: 16 significant bits,
// statistics only) -- by the workgroup half that matches the object's parity.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pn_split3_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
    const float r0 = v0 - __builtin_bit_cast(float, hu << 16), r1 = v1 - __builtin_bit_cast(float, hu & 0xffff0000u);
    const unsigned mu = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, mu << 16), s1 = r1 - __builtin_bit_cast(float, mu & 0xffff0000u);
    h = hu; m = mu; l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
__device__ __forceinline__ void pn_split3_8(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned a, b, c;
        pn_split3_pair(v[2 * p], v[2 * p + 1], a, b, c);
        h[p] = a; m[p] = b; l[p] = c;
    }
}

// two planes (16 significant bits, round to nearest at both steps: |v - h - m| <= 2^-17 |v|, unbiased) -- enough for the STATISTICS-only
// transposition of z2 (batch sums over >= 10^4 points; the values layer 3 consumes keep all three planes)
__device__ __forceinline__ void pn_split2_8(const float (&v)[8], u32x4& h, u32x4& m) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2 * p], v[2 * p + 1]}, bf16x2));
        const float r0 = v[2 * p] - __builtin_bit_cast(float, hu << 16), r1 = v[2 * p + 1] - __builtin_bit_cast(float, hu & 0xffff0000u);
        h[p] = hu;
        m[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    }
}

// SPLIT (few objects) as in pointnet_fwd_kernel: a workgroup (pair) takes one object at a time, its 8 waves share the 32-point tiles and leave
// partial (max, arg-max) pairs for pointnet_combine_kernel -- per tile the same arithmetic, so both forms give the same bits.
// LG (C3 = 256, many objects): ONE workgroup serves whole objects -- the h and m planes of W2 and of ALL of W3 fill the 160 KiB of LDS, the
// l planes (used by one of the six products each) are read per use from a 80-KiB operand-ordered copy in global memory (wlg, written by
// pointnet_p3_lplanes_kernel; it lives in L2): no second run of layers 1-2, 480 instead of 576 MFMAs per tile.  Same products in the same
// order per output as the channel-halves form: identical bits.
template <int C3, bool WITH_ARGMAX, bool BN, bool SPLIT = false, bool LG = false>
__global__ __launch_bounds__(PN_THREADS) void pointnet_fwd_p3_kernel(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ y,
    int* __restrict__ argmax, int T, int P, double* __restrict__ bn_part, float2* __restrict__ part, const u32x4* __restrict__ wlg = nullptr) {
    static_assert(!LG || (C3 == 256 && !SPLIT), "l planes from global memory: the C3 = 256 one-wave-per-object form");
    constexpr int HALVES = (C3 == 256 && !LG) ? 2 : 1, CH = C3 / HALVES, NBH = CH / 32, NB3 = C3 / 32;
    constexpr int NBN = 9 + 8 + 2 * NB3;
    constexpr int NPL = LG ? 2 : 3;                          // planes held in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned ldsu[];
    u32x4* w2p = reinterpret_cast<u32x4*>(ldsu);            // [NPL planes][4 cb2][4 ks][64 lane]        NPL x 16 KiB
    u32x4* w3p = w2p + NPL * 16 * 64;                        // [NPL planes][NBH cb3][8 ks3][64 lane]     NPL x CH / 4 KiB
    constexpr int W3N = NBH * 8 * 64;

    const int tid = threadIdx.x;
    const int hf = HALVES == 2 ? ((int)blockIdx.x & 1) : 0;
    const int slot = HALVES == 2 ? ((int)blockIdx.x >> 1) : (int)blockIdx.x, nslot = HALVES == 2 ? ((int)gridDim.x >> 1) : (int)gridDim.x;
    for (int d = tid; d < 16 * 64; d += PN_THREADS) {       // W2 as layer-2 A operand: row = out channel, k-slots = in channels 16 ks + 8 h + j
        const int ln = d & 63, ks = (d >> 6) & 3, cb = d >> 8;
        const float* src = w2 + (cb * 32 + (ln & 31)) * 64 + 16 * ks + 8 * (ln >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j];
        u32x4 ph, pm, pl;
        pn_split3_8(v, ph, pm, pl);
        w2p[d] = ph; w2p[1024 + d] = pm;
        if (!LG) w2p[2048 + d] = pl;
    }
    for (int d = tid; d < W3N; d += PN_THREADS) {           // W3 (this half's channels) as layer-3 B operand: k-slots follow layer 2's C layout
        const int ln = d & 63, ks3 = (d >> 6) & 7, cb = d >> 9;
        const int cb2 = ks3 >> 1, half8 = ks3 & 1, hh = ln >> 5;
        const float* src = w3 + (size_t)(hf * CH + cb * 32 + (ln & 31)) * 128 + cb2 * 32 + 16 * half8 + 4 * hh;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(j & 3) + 8 * (j >> 2)];
        u32x4 ph, pm, pl;
        pn_split3_8(v, ph, pm, pl);
        w3p[d] = ph; w3p[W3N + d] = pm;
        if (!LG) w3p[2 * W3N + d] = pl;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, pt = lane & 31;
    const int n_tiles = (P + 31) >> 5;
    double* const bn_dst = BN ? bn_part + ((size_t)blockIdx.x * PN_WAVES + wave) * NBN * 64 + lane : nullptr;
    auto bn_add = [&](int sl, float v) {
        if (BN) {
            double* d = bn_dst;
            asm volatile("" : "+v"(d));
            PN_BN_ATOMIC_ADD(d + sl * 64, (double)v);
        }
    };
    // bf16 identity slices for the layer-2 transposition: B[k slot (h, e), j] = [j == channel of the slot], k-slot order of layer 2's C layout
    auto make_ident = [&](int half8, int hh, int j) {
        u32x4 r;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int e0 = 2 * pp, e1 = 2 * pp + 1;
            const int c0 = (e0 & 3) + 8 * (2 * half8 + (e0 >> 2)) + 4 * hh, c1 = (e1 & 3) + 8 * (2 * half8 + (e1 >> 2)) + 4 * hh;
            r[pp] = (j == c0 ? 0x3F80u : 0u) | (j == c1 ? 0x3F800000u : 0u);
        }
        return r;
    };
    u32x4 ident[2];                                          // (LG: made where they are used -- 8 registers that kernel does not have)
    if (BN && !LG) { ident[0] = make_ident(0, h, pt); ident[1] = make_ident(1, h, pt); }

    for (int t = SPLIT ? slot : slot * PN_WAVES + wave; t < T; t += SPLIT ? nslot : nslot * PN_WAVES) {
        const float* xt = x + (size_t)t * P * 3;
        float best[NBH];
        int bidx[NBH];
#pragma unroll
        for (int c = 0; c < NBH; ++c) { best[c] = -INFINITY; bidx[c] = 0; }
        // BN: this object's per-lane sums: layer 2 blocks [0, 8), this half's layer 3 blocks (LG: the eight layer-3 blocks' sums go to their
        // slots tile by tile instead -- 16 more registers would spill)
        constexpr int NBACC = BN ? (LG ? 8 : 8 + 2 * NBH) : 1;
        float bacc[NBACC];
#pragma unroll
        for (int k = 0; k < NBACC; ++k) bacc[k] = 0.f;
        const bool do_l2_obj = BN && (HALVES == 1 || (__builtin_amdgcn_readfirstlane(t) & 1) == hf);     // (provably uniform: a scalar branch)

        auto tile_body = [&](int tile, auto tail_c, auto l2_c) {
            constexpr bool tail = BN && decltype(tail_c)::value, DO_L2 = BN && decltype(l2_c)::value;
            const int p0 = tile * 32;
            int lane_o = lane, h_o = h;                       // opaque copies: keep the weight reads inside the tile loop
            asm volatile("" : "+v"(lane_o), "+v"(h_o));
            // Global operands as UNIFORM base + 32-bit lane offset (the saddr form of global_load: the base moves on the scalar unit): with 64-bit
            // per-lane addresses the 128 loads of a tile cost ~570 VALU of address arithmetic (a quarter of the tile's VALU instructions).
            const unsigned loff = (unsigned)lane_o * 16u, hoff = (unsigned)h_o;
            const u32x4* wlg_t = wlg;                          // (opaque per tile: the ~80 scalar bases are made where they are used, not kept in SGPRs over the loop)
            asm volatile("" : "+s"(wlg_t));
            auto gl4 = [&](const void* base, unsigned byte_const, unsigned lane_bytes) {
                return *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(base) + byte_const + lane_bytes);
            };
            auto glu = [&](const void* base, unsigned byte_const, unsigned lane_bytes) {       // (explicitly GLOBAL: behind the opaque base the compiler would emit flat loads)
                typedef const __attribute__((address_space(1))) unsigned char* gptr;
                return *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((gptr)base + byte_const + lane_bytes);
            };
            const int pi = min(p0 + pt, P - 1);
            const float x0 = xt[pi * 3 + 0], x1 = xt[pi * 3 + 1], x2 = xt[pi * 3 + 2];
            auto bn_fold = [&](f32x16& v, int sl) {
                if (tail) {
                    const int lim = P - p0 - 4 * h_o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = mfma32_row(r, 0) < lim ? v[r] : 0.f;
                }
                float sm[4], sq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { sm[r] = v[r]; sq[r] = v[r] * v[r]; }
#pragma unroll
                for (int r = 4; r < 16; ++r) { sm[r & 3] += v[r]; sq[r & 3] = fmaf(v[r], v[r], sq[r & 3]); }
                if (LG && sl >= 8) {
                    bn_add(9 + sl, (sm[0] + sm[1]) + (sm[2] + sm[3])); bn_add(10 + sl, (sq[0] + sq[1]) + (sq[2] + sq[3]));
                } else {
                    bacc[BN && sl < NBACC ? sl : 0] += (sm[0] + sm[1]) + (sm[2] + sm[3]); bacc[BN && sl + 1 < NBACC ? sl + 1 : 0] += (sq[0] + sq[1]) + (sq[2] + sq[3]);
                }
            };

            // ---- layer 1 (VALU, fp32): this lane's 32 channels k = 16 ks + 8 h + j, split for the MFMA B operand
            u32x4 h1p[3][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 16 * ks + 8 * h_o;
                float v[8];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int kk = k + 4 * half;
                    const unsigned kc = 16 * ks + 4 * half;      // kk = kc + 8 h
                    const f32x4 wa = gl4(w1, kc * 12, hoff * 96);
                    const f32x4 wb = gl4(w1, kc * 12 + 16, hoff * 96);
                    const f32x4 wc = gl4(w1, kc * 12 + 32, hoff * 96);
                    const f32x4 bb = gl4(b1, kc * 4, hoff * 32);
                    (void)kk;
                    v[4 * half + 0] = fmaxf(fmaf(wa[2], x2, fmaf(wa[1], x1, fmaf(wa[0], x0, bb[0]))), 0.f);
                    v[4 * half + 1] = fmaxf(fmaf(wb[1], x2, fmaf(wb[0], x1, fmaf(wa[3], x0, bb[1]))), 0.f);
                    v[4 * half + 2] = fmaxf(fmaf(wc[0], x2, fmaf(wb[3], x1, fmaf(wb[2], x0, bb[2]))), 0.f);
                    v[4 * half + 3] = fmaxf(fmaf(wc[3], x2, fmaf(wc[2], x1, fmaf(wc[1], x0, bb[3]))), 0.f);
                }
                pn_split3_8(v, h1p[0][ks], h1p[1][ks], h1p[2][ks]);
            }

            // ---- layer 2: H2^T = W2 H1^T (A = W2 planes from LDS, B = H1 planes); the h h products start at the bias, the small ones at zero
            u32x4 h2p[3][8];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x16 acc, accs;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bb = gl4(b2, (cb * 32 + 8 * g) * 4, hoff * 16);
                    acc[g * 4 + 0] = bb[0]; acc[g * 4 + 1] = bb[1]; acc[g * 4 + 2] = bb[2]; acc[g * 4 + 3] = bb[3];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int d = (cb * 4 + ks) * 64 + lane_o;
                    const u32x4 wh = w2p[d], wm = w2p[1024 + d], wl = LG ? glu(wlg_t, (cb * 4 + ks) * 1024, loff) : w2p[2048 + d];
                    accs = mfma_bf16(wl, h1p[0][ks], accs);
                    accs = mfma_bf16(wh, h1p[2][ks], accs);
                    accs = mfma_bf16(wm, h1p[1][ks], accs);
                    acc = mfma_bf16(wh, h1p[0][ks], acc);
                    accs = mfma_bf16(wm, h1p[0][ks], accs);
                    accs = mfma_bf16(wh, h1p[1][ks], accs);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += accs[r];
                if (DO_L2) {
                    // two planes of z2 itself (16 bits: statistics only), re-delivered with lane = channel through the identity
                    f32x16 tr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) tr[r] = 0.f;
                    u32x4 zp[2][2];
#pragma unroll
                    for (int half8 = 0; half8 < 2; ++half8) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = acc[half8 * 8 + j];
                        pn_split2_8(v, zp[0][half8], zp[1][half8]);
                    }
#pragma unroll
                    for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                        for (int half8 = 0; half8 < 2; ++half8) tr = mfma_bf16(zp[pl][half8], LG ? make_ident(half8, h_o, lane_o & 31) : ident[half8], tr);
                    bn_fold(tr, 2 * cb);
                    __builtin_amdgcn_sched_barrier(PN_BN_SB);
                }
#pragma unroll
                for (int half8 = 0; half8 < 2; ++half8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(acc[half8 * 8 + j], 0.f);
                    pn_split3_8(v, h2p[0][cb * 2 + half8], h2p[1][cb * 2 + half8], h2p[2][cb * 2 + half8]);
                }
            }

            // ---- layer 3: Z3 = H2 W3^T (A = H2 planes, B = this half's W3 planes), running max over points
            // LG: the l-plane fragments come from L2 (~600 cycles): requested PN_LPF steps ahead (a ring of PN_LPF + 1 fragments), not one
#ifndef PN_LPF
#define PN_LPF 3
#endif
            u32x4 wlq[PN_LPF + 1];
            if (LG) {
#pragma unroll
                for (int i = 0; i < PN_LPF; ++i) wlq[i] = glu(wlg_t, (16 + i) * 1024, loff);
            }
#pragma unroll
            for (int cb = 0; cb < NBH; ++cb) {
                f32x16 acc, accs;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accs[r] = 0.f; }
#pragma unroll
                for (int ks3 = 0; ks3 < 8; ++ks3) {
                    const int d = (cb * 8 + ks3) * 64 + lane_o;
                    const int li = cb * 8 + ks3;
                    if (LG && li + PN_LPF < NBH * 8) wlq[(li + PN_LPF) % (PN_LPF + 1)] = glu(wlg_t, (16 + li + PN_LPF) * 1024, loff);
                    const u32x4 wh = w3p[d], wm = w3p[W3N + d], wl = LG ? wlq[li % (PN_LPF + 1)] : w3p[2 * W3N + d];
                    accs = mfma_bf16(h2p[2][ks3], wh, accs);
                    accs = mfma_bf16(h2p[0][ks3], wl, accs);
                    accs = mfma_bf16(h2p[1][ks3], wm, accs);
                    acc = mfma_bf16(h2p[0][ks3], wh, acc);
                    accs = mfma_bf16(h2p[1][ks3], wh, accs);
                    accs = mfma_bf16(h2p[0][ks3], wm, accs);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += accs[r];
                if (WITH_ARGMAX) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {        // ascending point order, strict > keeps the first max
                        const bool gt = acc[r] > best[cb];
                        best[cb] = gt ? acc[r] : best[cb];
                        bidx[cb] = gt ? (p0 + mfma32_row(r, h)) : bidx[cb];
                    }
                } else {
                    float m = acc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                    best[cb] = fmaxf(best[cb], m);
                }
                if (BN) { bn_fold(acc, 8 + 2 * cb); __builtin_amdgcn_sched_barrier(PN_BN_SB); }
            }
        };
        for (int tile = SPLIT ? wave : 0; tile < n_tiles; tile += SPLIT ? PN_WAVES : 1) {
            const bool tl = BN && tile * 32 + 32 > P;
            if (do_l2_obj) { if (tl) tile_body(tile, std::true_type{}, std::true_type{}); else tile_body(tile, std::false_type{}, std::true_type{}); }
            else { if (tl) tile_body(tile, std::true_type{}, std::false_type{}); else tile_body(tile, std::false_type{}, std::false_type{}); }
        }

        if (BN) {
            if (do_l2_obj) {
#pragma unroll
                for (int k = 0; k < 8; ++k) bn_add(9 + k, bacc[k]);
            }
#pragma unroll
            for (int k = 0; k < (LG ? 0 : 2 * NBH); ++k) bn_add(17 + 2 * hf * NBH + k, bacc[BN && !LG ? 8 + k : 0]);
        }
#pragma unroll
        for (int cb = 0; cb < NBH; ++cb) {
            const float ov = __shfl_xor(best[cb], 32, 64);
            float v = best[cb];
            int bi = bidx[cb];
            if (WITH_ARGMAX) {
                const int oi = __shfl_xor(bidx[cb], 32, 64);
                const bool take = (ov > v) || (ov == v && oi < bi);
                v = take ? ov : v;
                bi = take ? oi : bi;
                bi = min(bi, P - 1);
            } else {
                v = fmaxf(v, ov);
            }
            if (h == 0) {
                const int c = hf * CH + cb * 32 + pt;
                if (SPLIT) {
                    part[((size_t)t * PN_WAVES + wave) * C3 + c] = float2{v, __int_as_float(bi)};
                } else {
                    y[(size_t)t * C3 + c] = fmaxf(v + b3[c], 0.f);
                    if (WITH_ARGMAX) argmax[(size_t)t * C3 + c] = bi;
                }
            }
        }
    }
}

// The l planes of W2 and W3 in the MFMA operand order of pointnet_fwd_p3_kernel<.., LG>: out[0, 1024) = W2, out[1024, 1024 + C3 / 32 * 512) = W3.
template <int C3>
__global__ __launch_bounds__(256) void pointnet_p3_lplanes_kernel(const float* __restrict__ w2, const float* __restrict__ w3, u32x4* __restrict__ out) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= 1024 + (C3 / 32) * 512) return;
    float v[8];
    if (d < 1024) {
        const int ln = d & 63, ks = (d >> 6) & 3, cb = d >> 8;
        const float* src = w2 + (cb * 32 + (ln & 31)) * 64 + 16 * ks + 8 * (ln >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j];
    } else {
        const int e = d - 1024, ln = e & 63, ks3 = (e >> 6) & 7, cb = e >> 9;
        const int cb2 = ks3 >> 1, half8 = ks3 & 1, hh = ln >> 5;
        const float* src = w3 + (size_t)(cb * 32 + (ln & 31)) * 128 + cb2 * 32 + 16 * half8 + 4 * hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(j & 3) + 8 * (j >> 2)];
    }
    u32x4 ph, pm, pl;
    pn_split3_8(v, ph, pm, pl);
    out[d] = pl;
}

// Forward arithmetic, chosen PER CALL (the library keeps no mode): 0 = exact fp32 (v_mfma_f32_32x32x2_f32); 4 = three exact bf16 planes, six
// bf16 MFMAs per product (fp32 arithmetic on the bf16 matrix pipe; both launch forms).
template <int C3>
int launch_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
               const float* w3, const float* b3, float* y, int* argmax, int T, int P, hipStream_t stream,
               void* workspace, size_t ws_bytes, int mode, double* bn_part = nullptr, double* bn_out = nullptr) {
    const size_t lds_bytes = (size_t)(8192 + C3 * 128) * sizeof(float);
    int grid = (T + PN_WAVES - 1) / PN_WAVES;
    const int ncu = sga_num_cus();
    if (grid > ncu) grid = ncu;
    const bool split_small = workspace && ws_bytes >= (size_t)T * PN_WAVES * C3 * sizeof(float2) && T < 4 * ncu && P > 32;
    // mode 4: three exact bf16 planes (pointnet_fwd_p3_kernel); at C3 = 256 a workgroup holds half of W3's channels, two share an object
    constexpr int P3_HALVES = C3 == 256 ? 2 : 1;
    const size_t p3_lds = (size_t)(3 * 16 * 64 + 3 * (C3 / P3_HALVES / 32) * 8 * 64) * 16;
    const int p3_slots_max = P3_HALVES == 2 ? (ncu / 2 > 0 ? ncu / 2 : 1) : ncu;
    const int p3_want = split_small ? T : grid;              // split form: one object per workgroup (pair) at a time
    // C3 = 256, many objects, a scratch of PN_P3_LG_BYTES in `workspace`: one workgroup per object, the l planes from global memory (LG)
    const bool p3_lg = C3 == 256 && !split_small && workspace && ws_bytes >= PN_P3_LG_BYTES;
    const int p3_grid = p3_lg ? grid : P3_HALVES * (p3_want < p3_slots_max ? p3_want : p3_slots_max);
    auto go_p3 = [&](double* bnp) {
        if constexpr (C3 == 256) {
            if (p3_lg) {
                u32x4* wlg = static_cast<u32x4*>(workspace);
                hipLaunchKernelGGL(pointnet_p3_lplanes_kernel<C3>, dim3((1024 + 8 * 512 + 255) / 256), dim3(256), 0, stream, w2, w3, wlg);
                const size_t lg_lds = (size_t)(2 * 16 * 64 + 2 * 8 * 8 * 64) * 16;
                auto launch = [&](auto k) {
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg_lds);
                    hipLaunchKernelGGL(k, dim3(p3_grid), dim3(PN_THREADS), lg_lds, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, bnp, static_cast<float2*>(nullptr),
                                       static_cast<const u32x4*>(wlg));
                };
                if (bnp) { if (argmax) launch(pointnet_fwd_p3_kernel<C3, true, true, false, true>); else launch(pointnet_fwd_p3_kernel<C3, false, true, false, true>); }
                else { if (argmax) launch(pointnet_fwd_p3_kernel<C3, true, false, false, true>); else launch(pointnet_fwd_p3_kernel<C3, false, false, false, true>); }
                return;
            }
        }
        float2* part = split_small ? static_cast<float2*>(workspace) : nullptr;
        auto launch = [&](auto k) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p3_lds);
            hipLaunchKernelGGL(k, dim3(p3_grid), dim3(PN_THREADS), p3_lds, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, bnp, part, static_cast<const u32x4*>(nullptr));
        };
        if (split_small) {
            if (bnp) { if (argmax) launch(pointnet_fwd_p3_kernel<C3, true, true, true>); else launch(pointnet_fwd_p3_kernel<C3, false, true, true>); }
            else { if (argmax) launch(pointnet_fwd_p3_kernel<C3, true, false, true>); else launch(pointnet_fwd_p3_kernel<C3, false, false, true>); }
            hipLaunchKernelGGL(pointnet_combine_kernel, dim3((T * C3 + 255) / 256), dim3(256), 0, stream, part, b3, y, argmax, T, C3, P);
        } else {
            if (bnp) { if (argmax) launch(pointnet_fwd_p3_kernel<C3, true, true>); else launch(pointnet_fwd_p3_kernel<C3, false, true>); }
            else { if (argmax) launch(pointnet_fwd_p3_kernel<C3, true, false>); else launch(pointnet_fwd_p3_kernel<C3, false, false>); }
        }
    };
    if (bn_out) {
        // the forward with the batch statistics of the three pre-activations (the reference's BatchNorm side effect): exact fp32 or mode 4
        const bool p3 = mode == 4;
        const bool split = split_small && !p3;
        const int g = p3 ? p3_grid : split ? (T < ncu ? T : ncu) : grid;
        float2* part = split ? static_cast<float2*>(workspace) : nullptr;
        if (hipError_t me = hipMemsetAsync(bn_part, 0, (size_t)g * PN_WAVES * (9 + 8 + 2 * (C3 / 32)) * 64 * sizeof(double), stream); me != hipSuccess) {
            sga_set_error("sga_pointnet_fwd_bn: memset of %zu bytes at %p failed: %s", (size_t)g * PN_WAVES * (9 + 8 + 2 * (C3 / 32)) * 64 * sizeof(double), (void*)bn_part, hipGetErrorString(me));
            return SGA_ERR_HIP;
        }
        auto go = [&](auto k) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(k, dim3(g), dim3(PN_THREADS), lds_bytes, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, part, bn_part);
        };
        if (p3) go_p3(bn_part);
        else if (split) { if (argmax) go(pointnet_fwd_kernel<C3, true, true, true>); else go(pointnet_fwd_kernel<C3, false, true, true>); }
        else { if (argmax) go(pointnet_fwd_kernel<C3, true, false, true>); else go(pointnet_fwd_kernel<C3, false, false, true>); }
        if (split) hipLaunchKernelGGL(pointnet_combine_kernel, dim3((T * C3 + 255) / 256), dim3(256), 0, stream, part, b3, y, argmax, T, C3, P);
        hipLaunchKernelGGL(pointnet_xmoments_kernel, dim3(g), dim3(PN_THREADS), 0, stream, x, (size_t)T * P, C3, bn_part);
        hipLaunchKernelGGL(pointnet_bn_reduce_kernel, dim3(265 + 2 * C3), dim3(64), 0, stream, bn_part, g * PN_WAVES, C3, bn_out);
        SGA_CHECK_LAUNCH("sga_pointnet_fwd_bn");
        return SGA_OK;
    }
    // few objects: one object per workgroup, its tiles dealt to the 8 waves (exact fp32 kernel only; needs the partials workspace)
    if (mode == 0 && split_small) {
        float2* part = static_cast<float2*>(workspace);
        const int g2 = T < ncu ? T : ncu;
        if (argmax) {
            auto k = pointnet_fwd_kernel<C3, true, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(k, dim3(g2), dim3(PN_THREADS), lds_bytes, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, part, static_cast<double*>(nullptr));
        } else {
            auto k = pointnet_fwd_kernel<C3, false, true>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(k, dim3(g2), dim3(PN_THREADS), lds_bytes, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, part, static_cast<double*>(nullptr));
        }
        hipLaunchKernelGGL(pointnet_combine_kernel, dim3((T * C3 + 255) / 256), dim3(256), 0, stream, part, b3, y, argmax, T, C3, P);
        SGA_CHECK_LAUNCH("sga_pointnet_fwd");
        return SGA_OK;
    }
    if (mode == 4) {
        go_p3(nullptr);
    } else if (argmax) {
        auto k = pointnet_fwd_kernel<C3, true>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k, dim3(grid), dim3(PN_THREADS), lds_bytes, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, static_cast<float2*>(nullptr), static_cast<double*>(nullptr));
    } else {
        auto k = pointnet_fwd_kernel<C3, false>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k, dim3(grid), dim3(PN_THREADS), lds_bytes, stream, x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, static_cast<float2*>(nullptr), static_cast<double*>(nullptr));
    }
    SGA_CHECK_LAUNCH("sga_pointnet_fwd");
    return SGA_OK;
}

}  // namespace

extern "C" size_t sga_pointnet_fwd_ws_bytes(int T, int C3) { return (size_t)(T > 0 ? T : 0) * PN_WAVES * (C3 > 0 ? C3 : 0) * sizeof(float2); }

static int pointnet_fwd_impl(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                             const float* b3, float* y, int32_t* argmax, int T, int P, int C3, void* workspace, size_t ws_bytes,
                             int mode, void* stream, double* bn_part = nullptr, double* bn_out = nullptr) {
    SGA_CHECK_ARG(T >= 0 && P >= 1, "sga_pointnet_fwd: need T >= 0 and P >= 1 (got T=%d P=%d)", T, P);
    SGA_CHECK_ARG(mode == 0 || mode == 4, "sga_pointnet_fwd: mode %d (0 = exact fp32, 4 = three exact bf16 planes)", mode);
    // a zero-object shard (T == 0: empty tensors carry null data pointers) is a valid no-op
    SGA_CHECK_ARG((T == 0 || (x && y)) && w1 && b1 && w2 && b2 && w3 && b3, "sga_pointnet_fwd: null pointer");
    if (T == 0) return SGA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (C3) {
        case 256: return launch_fwd<256>(x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, s, workspace, ws_bytes, mode, bn_part, bn_out);
        case 128: return launch_fwd<128>(x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, s, workspace, ws_bytes, mode, bn_part, bn_out);
        case 64: return launch_fwd<64>(x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, s, workspace, ws_bytes, mode, bn_part, bn_out);
        default:
            sga_set_error("sga_pointnet_fwd: out_size C3=%d unsupported (64, 128 or 256: W3 must fit the 160 KiB LDS)", C3);
            return SGA_ERR_ARG;
    }
}

extern "C" int sga_pointnet_fwd_ws(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* w3, const float* b3, float* y, int32_t* argmax, int T, int P, int C3,
                                   void* workspace, size_t ws_bytes, int mode, void* stream) {
    return pointnet_fwd_impl(x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, C3, workspace, ws_bytes, mode, stream);
}

/* The exact-fp32 forward that also delivers the batch statistics the reference's three discarded BatchNorm calls fold into their running
 * buffers (pointnet.py:141-142,154-155,158-159): bn_sums[265 + 2 C3] doubles (layout: pointnet_bn_reduce_kernel). */
extern "C" size_t sga_pointnet_fwd_bn_ws_bytes(int T, int C3) {
    if (T <= 0 || C3 <= 0) return 0;
    const int ncu = sga_num_cus();
    const int g = 2 * T < ncu ? 2 * T : ncu;              // an upper bound of every launch form's workgroup count
    return (size_t)g * PN_WAVES * (9 + 8 + 2 * (C3 / 32)) * 64 * sizeof(double);
}
extern "C" int sga_pointnet_fwd_bn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* w3, const float* b3, float* y, int32_t* argmax, int T, int P, int C3,
                                   void* workspace, size_t ws_bytes, void* bn_workspace, size_t bn_ws_bytes, double* bn_sums, int mode, void* stream) {
    SGA_CHECK_ARG(bn_sums != nullptr, "sga_pointnet_fwd_bn: bn_sums is null");
    SGA_CHECK_ARG(mode == 0 || mode == 4, "sga_pointnet_fwd_bn: mode %d (0 = exact fp32, 4 = three exact bf16 planes)", mode);
    SGA_CHECK_ARG(T == 0 || (bn_workspace && bn_ws_bytes >= sga_pointnet_fwd_bn_ws_bytes(T, C3)),
                  "sga_pointnet_fwd_bn: bn_workspace must hold sga_pointnet_fwd_bn_ws_bytes(T, C3) = %zu bytes (got %zu)", sga_pointnet_fwd_bn_ws_bytes(T, C3), bn_ws_bytes);
    if (T == 0) {
        if (C3 == 64 || C3 == 128 || C3 == 256) hipMemsetAsync(bn_sums, 0, (size_t)(265 + 2 * C3) * sizeof(double), static_cast<hipStream_t>(stream));
    }
    return pointnet_fwd_impl(x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, C3, workspace, ws_bytes, mode, stream, static_cast<double*>(bn_workspace), bn_sums);
}

/* the no-workspace entry: always the exact-fp32 kernel */
extern "C" int sga_pointnet_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, float* y,
                                int32_t* argmax, int T, int P, int C3, void* stream) {
    return pointnet_fwd_impl(x, w1, b1, w2, b2, w3, b3, y, argmax, T, P, C3, nullptr, 0, 0, stream);
}

// =================================================================================================
// Backward (sparse through the max-pool).
//
// Autograd of pointnet.py:140-161: only the arg-max point of each (object, channel) carries gradient,
// so per object the backward touches <= C3 "winner rows" r (row r = channel r, point argmax[t,r]):
//   g_r   = gy[t,r] * (y[t,r] > 0)                         x_r = x[t, argmax[t,r]]
//   H1    = relu(W1 x_r + b1) [C3,64]   Z2 = W2 H1 + b2 [C3,128]   (recomputed, never stored)
//   gb3[r] += g_r ;  gW3[r,:] += g_r relu(Z2)[r,:]         (row r of W3 only sees winner row r)
//   dZ2   = g_r W3[r,:] * (Z2 > 0)      gW2 += dZ2^T H1    gb2 += colsum(dZ2)
//   dZ1   = (dZ2 W2) * (H1 > 0)         gW1 += dZ1^T X     gb1 += colsum(dZ1)
// One wave owns one 32-row tile of the object.  Reductions over rows need the rows on the MFMA K
// dimension, reductions over channels need the channels there: dZ2 is computed once (lane = row) and
// transposed through a wave-private LDS tile for the other orientation; every weight-gradient partial
// accumulates in registers across objects and is flushed once per workgroup.
// =================================================================================================
namespace {

// -------------------------------------------------------------------------------------------------
// Fused backward: ONE recomputation of Z2 per winner-row tile (384 MFMAs instead of 512).
// dZ2 is produced in the lane = row orientation (what dH1 = dZ2 W2 needs as its A operand) and TRANSPOSED through a
// wave-private 32x32 LDS tile per channel block into the lane = channel orientation that gW2 += dZ2^T H1 needs:
// 4 ds_write_b128 + 16 ds_read_b32 per lane instead of 32 more MFMAs (the two-phase predecessor recomputed Z2 in
// each orientation: 512 MFMAs per tile, 10.2 ms at configs[1]; this kernel: 7.7 ms).  One wave per SIMD (the 8 gW2 tiles + 64 gW3
// partials + dH1 need ~400 registers); a workgroup's 4 waves own 4 fixed tile indices -- even workgroups tiles 0-3,
// odd ones 4-7 -- so the gW3 partials stay private to (wave, lane) across objects.
// -------------------------------------------------------------------------------------------------
constexpr int TRS = 36;            // transpose tile row stride (floats): 16-byte aligned rows, conflict-free column reads
constexpr int W3S = 68;            // row stride (floats) of the three-plane backward's W3 rows in LDS (64 channels + 4: 16-byte aligned, rows off each other's banks)

__global__ __launch_bounds__(256) void pointnet_bwd_fused_kernel(
    const float* __restrict__ x, const int* __restrict__ argmax, const float* __restrict__ y,
    const float* __restrict__ gy, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
    float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2,
    float* __restrict__ gw3, float* __restrict__ gb3, int T, int P) {
    constexpr int C3 = 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w2s = lds;               // operand order [4 cb][8 q][64 lane][4]
    float* w2r = lds + 8192;        // row-major [128][64]
    float* trs = lds + 16384;       // [4 waves][32][TRS] transpose tiles; reused as the gW2 combine buffer at the end
    float* w1s = lds + 16384 + 8192;
    float* b1s = w1s + 192;
    float* b2s = b1s + 64;
    const int tid = threadIdx.x;
    for (int d = tid; d < 384; d += 256) w1s[d] = d < 192 ? w1[d] : (d < 256 ? b1[d - 192] : b2[d - 256]);
    for (int d = tid; d < 2048; d += 256) {
        const int ln = d & 63, q = (d >> 6) & 7, cb = d >> 9;
        const int row = cb * 32 + (ln & 31), k = 8 * q + 4 * (ln >> 5);
        *reinterpret_cast<f32x4*>(w2s + d * 4) = *reinterpret_cast<const f32x4*>(w2 + row * 64 + k);
        *reinterpret_cast<f32x4*>(w2r + d * 4) = *reinterpret_cast<const f32x4*>(w2 + d * 4);
    }
    __syncthreads();

    const int lane = tid & 63, wave0 = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wave = (blockIdx.x & 1) * 4 + wave0;           // this wave's winner-row tile, fixed for the whole kernel
    float* tr = trs + wave0 * 32 * TRS;
    float gw3a[64];                 // [cb][gq][r] : gW3[wave*32 + l31][cb*32 + 8gq + 4h + r]
#pragma unroll
    for (int i = 0; i < 64; ++i) gw3a[i] = 0.f;
    f32x16 gw2a[8];                 // [cb][kt] : gW2[cb*32 + row(r,h)][kt*32 + l31]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) gw2a[i][r] = 0.f;
    float gb3a = 0.f, gb2a[4] = {0.f, 0.f, 0.f, 0.f};
    float gw1a[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, gb1a[2] = {0.f, 0.f};

    const int obj0 = (int)blockIdx.x >> 1, ostep = ((int)gridDim.x + 1) >> 1;
    const int n_iter = obj0 < T ? (T - obj0 + ostep - 1) / ostep : 0;
    float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, ng = 0.f;
    if (n_iter > 0) {
        const size_t ri = (size_t)obj0 * C3 + wave * 32 + l31;
        const int p0 = min(max(argmax[ri], 0), P - 1);
        const float* xp = x + ((size_t)obj0 * P + p0) * 3;
        nx0 = xp[0]; nx1 = xp[1]; nx2 = xp[2];
        ng = y[ri] > 0.f ? gy[ri] : 0.f;
    }
    for (int it = 0; it < n_iter; ++it) {
        const int t = obj0 + it * ostep;
        int lane_o = lane, h_o = h;
        asm volatile("" : "+v"(lane_o), "+v"(h_o));          // keep weight reads inside the loop (see fwd)
        const int l31_o = lane_o & 31;
        const int c = wave * 32 + l31_o;                     // this lane's winner row (lane = row layouts)
        const float x0 = nx0, x1 = nx1, x2 = nx2, g = ng;
        const int tn = obj0 + min(it + 1, n_iter - 1) * ostep;
        const size_t rin = (size_t)tn * C3 + wave * 32 + l31_o;
        const int pn_raw = argmax[rin];
        const float yn = y[rin], gyn = gy[rin];
        if (h == 0) gb3a += g;

        // ---- H1, lane = row layout (k = 8q + 4h + r)
        float h1[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 8 * q + 4 * h_o;
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w1s + k * 3);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w1s + k * 3 + 4);
            const f32x4 wc = *reinterpret_cast<const f32x4*>(w1s + k * 3 + 8);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + k);
            h1[q * 4 + 0] = fmaxf(fmaf(wa[2], x2, fmaf(wa[1], x1, fmaf(wa[0], x0, bb[0]))), 0.f);
            h1[q * 4 + 1] = fmaxf(fmaf(wb[1], x2, fmaf(wb[0], x1, fmaf(wa[3], x0, bb[1]))), 0.f);
            h1[q * 4 + 2] = fmaxf(fmaf(wc[0], x2, fmaf(wb[3], x1, fmaf(wb[2], x0, bb[2]))), 0.f);
            h1[q * 4 + 3] = fmaxf(fmaf(wc[3], x2, fmaf(wc[2], x1, fmaf(wc[1], x0, bb[3]))), 0.f);
        }
        // ---- H1, lane = k1 / regs = rows layout, from two K = 2 MFMAs per 32 channels:
        //      D[row][k1] = x0 W1[k1][0] + x1 W1[k1][1] + x2 W1[k1][2] + 1 * b1[k1]  (A: lane = row, h = k; B: lane = k1, h = k)
        f32x16 h1c[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int k1 = kt * 32 + l31_o;
            const float bA = w1s[k1 * 3 + h_o];
            const float bB = h_o ? b1s[k1] : w1s[k1 * 3 + 2];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h_o ? x1 : x0, bA, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h_o ? 1.f : x2, bB, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) h1c[kt][r] = fmaxf(acc[r], 0.f);
        }
        {   // the next tile's point: its index arrived during the H1 section
            const int pn = min(max(pn_raw, 0), P - 1);
            const float* xpn = x + ((size_t)tn * P + pn) * 3;
            nx0 = xpn[0]; nx1 = xpn[1]; nx2 = xpn[2];
            ng = yn > 0.f ? gyn : 0.f;
        }

        f32x16 dh1[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh1[kt][r] = 0.f;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            __builtin_amdgcn_sched_barrier(0);
            // Z2[row = lane][ch2 = cb*32 + 8gq + 4h + r]
            f32x16 acc;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + cb * 32 + 8 * gq + 4 * h_o);
                acc[gq * 4 + 0] = bb[0]; acc[gq * 4 + 1] = bb[1]; acc[gq * 4 + 2] = bb[2]; acc[gq * 4 + 3] = bb[3];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w2s + ((cb * 8 + q) * 64 + lane_o) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[r], h1[q * 4 + r], acc, 0, 0, 0);
            }
            // dZ2 = g * W3[c][ch2] * (Z2 > 0); gW3 partials; the tile goes to LDS for the transpose
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 w3v = *reinterpret_cast<const f32x4*>(w3 + (size_t)c * 128 + cb * 32 + 8 * gq + 4 * h_o);
                f32x4 dz;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = acc[gq * 4 + r];
                    gw3a[cb * 16 + gq * 4 + r] = fmaf(g, fmaxf(z, 0.f), gw3a[cb * 16 + gq * 4 + r]);
                    dz[r] = (z > 0.f ? g : 0.f) * w3v[r];
                    acc[gq * 4 + r] = dz[r];
                }
                *reinterpret_cast<f32x4*>(tr + l31_o * TRS + 8 * gq + 4 * h_o) = dz;       // tr[row][ch2 local]
            }
            // dH1[row][k1] += sum_ch2 dZ2[row][ch2] W2[ch2][k1]
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float* wrow = w2r + (cb * 32 + mfma32_row(s, h_o)) * 64 + l31_o;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    dh1[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[s], wrow[kt * 32], dh1[kt], 0, 0, 0);
            }
            // lane = ch2, regs = rows: gb2 and gW2 += dZ2^T H1   (wave-private tile: the waitcnt of the reads orders them
            // after this wave's own writes; no barrier)
            float colsum = 0.f;
            f32x16 dzt;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                dzt[s] = tr[mfma32_row(s, h_o) * TRS + l31_o];
                colsum += dzt[s];
            }
            gb2a[cb] += colsum;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    gw2a[cb * 2 + kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dzt[s], h1c[kt][s], gw2a[cb * 2 + kt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- dZ1 = dH1 * (H1 > 0) in the lane = k1 layout; gW1 / gb1 (x of row(s,h) by lane shuffle)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int src = mfma32_row(s, h);
            const float sx0 = __shfl(x0, src, 64), sx1 = __shfl(x1, src, 64), sx2 = __shfl(x2, src, 64);
            const float dz0 = h1c[0][s] > 0.f ? dh1[0][s] : 0.f;
            const float dz1 = h1c[1][s] > 0.f ? dh1[1][s] : 0.f;
            gw1a[0][0] = fmaf(dz0, sx0, gw1a[0][0]); gw1a[0][1] = fmaf(dz0, sx1, gw1a[0][1]); gw1a[0][2] = fmaf(dz0, sx2, gw1a[0][2]);
            gw1a[1][0] = fmaf(dz1, sx0, gw1a[1][0]); gw1a[1][1] = fmaf(dz1, sx1, gw1a[1][1]); gw1a[1][2] = fmaf(dz1, sx2, gw1a[1][2]);
            gb1a[0] += dz0;
            gb1a[1] += dz1;
        }
    }

    // ---- flush the per-workgroup partials
#pragma unroll
    for (int i = 0; i < 64; ++i)
        atomicAdd(gw3 + (size_t)(wave * 32 + l31) * 128 + (i >> 4) * 32 + 8 * ((i >> 2) & 3) + 4 * h + (i & 3), gw3a[i]);
    if (h == 0) atomicAdd(gb3 + wave * 32 + l31, gb3a);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int k1 = kt * 32 + l31;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = gw1a[kt][d] + __shfl_xor(gw1a[kt][d], 32, 64);
            if (h == 0) atomicAdd(gw1 + k1 * 3 + d, v);
        }
        const float vb = gb1a[kt] + __shfl_xor(gb1a[kt], 32, 64);
        if (h == 0) atomicAdd(gb1 + k1, vb);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const float v = gb2a[cb] + __shfl_xor(gb2a[cb], 32, 64);
        if (h == 0) atomicAdd(gb2 + cb * 32 + l31, v);
    }
    // gW2: combine the 4 waves in LDS (the transpose tiles are dead now; 8192 floats needed, 4*32*TRS = 4608 available ->
    // two halves), then one global atomic per element
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        for (int d = tid; d < 4096; d += 256) trs[d] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(trs + (((half * 4 + i) >> 1) * 32 + mfma32_row(r, h) - half * 64) * 64 + ((half * 4 + i) & 1) * 32 + l31, gw2a[half * 4 + i][r]);
        __syncthreads();
        for (int d = tid; d < 4096; d += 256) atomicAdd(gw2 + half * 4096 + d, trs[d]);
        __syncthreads();
    }
}


// -------------------------------------------------------------------------------------------------
// The same fused backward with the three GEMMs of a winner-row tile on three exact bf16 planes (mode 4, the default arithmetic; six bf16
// MFMAs per product as in the forward): 288 v_mfma_f32_32x32x16_bf16 of 32 cycles per tile instead of 384 v_mfma_f32_32x32x2_f32 of 64.
//   Z2^T = W2 H1^T      A = W2 planes, operand order (LDS, as the forward's layer 2)   B = H1 planes (lane = row)        D: lane = row, regs = ch2
//   dH1  = dZ2 W2       A = dZ2 planes: the D registers of Z2 ARE the A k-slots        B = W2 planes in that k-slot order D: lane = k1,  regs = rows
//   gW2 += dZ2^T H1     A = planes of dZ2^T (transposed through the wave's LDS tile)   B = planes of H1 (lane = k1)     D: lane = k1,  regs = ch2
// Z2 is recomputed with the forward kernel's products in the forward kernel's order: the ReLU masks of the backward are the forward's own bits.
// The five small partial products of every accumulation go to their own accumulator (the 16-bit MFMAs chop toward minus infinity what falls
// below the result's last place: tools/micro/mfma_round_probe.hip); gW2's products of a tile start from zero and are added to the running
// fp32 sums on the VALU (round to nearest), so that nothing small is ever added onto a large MFMA accumulator across the million objects
// of a batch.  Everything else -- H1, the ReLU masks, gW3 / gb3, dZ1 and gW1 / gb1 -- is the fp32 VALU code of the kernel above.
// Work split: the operand planes cost registers the fp32 kernel spends on accumulators, so a wave owns a winner-row tile AND one half of the
// 128 layer-2 channels (64 running gW2 sums + 32 gW3 sums per lane instead of 128 + 64): workgroup kinds (tiles 0-3 | 4-7) x (ch2 0-63 | 64-127).
// dH1 = dZ2 W2 is linear in dZ2, so each half's partial dH1 goes through the ReLU mask and into gW1 / gb1 on its own.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pointnet_bwd_p3_kernel(
    const float* __restrict__ x, const int* __restrict__ argmax, const float* __restrict__ y,
    const float* __restrict__ gy, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
    float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2,
    float* __restrict__ gw3, float* __restrict__ gb3, int T, int P) {
    constexpr int C3 = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned ldsb[];
    u32x4* w2a = reinterpret_cast<u32x4*>(ldsb);            // [3 planes][2 ci][4 ks][64 lane]         A operand of Z2^T = W2 H1^T (this half's channels)  24 KiB
    u32x4* w2b = w2a + 3 * 512;                              // [3 planes][2 ci][2 half8][2 kt][64 lane] B operand of dH1 = dZ2 W2                            24 KiB
    float* trs = reinterpret_cast<float*>(w2b + 3 * 512);    // [4 waves][32][TRS] transpose tiles; reused as the gW2 combine buffer at the end
    float* w3s = trs + 4 * 32 * TRS;                         // [128 rows c of this tile group][64 ch2 of this half + 4 pad]: loop invariant          34 KiB
    float* w1s = w3s + 128 * W3S;
    float* b1s = w1s + 192;
    float* b2s = b1s + 64;
    const int tid = threadIdx.x;
    const int kind = (int)blockIdx.x & 3, tg = kind & 1, chh = kind >> 1;        // tile group, ch2 half
    for (int d = tid; d < 128 * 16; d += 256) {
        const int row = d >> 4, q = d & 15;
        *reinterpret_cast<f32x4*>(w3s + row * W3S + 4 * q) = *reinterpret_cast<const f32x4*>(w3 + (size_t)(tg * 128 + row) * 128 + chh * 64 + 4 * q);
    }
    for (int d = tid; d < 384; d += 256) w1s[d] = d < 192 ? w1[d] : (d < 256 ? b1[d - 192] : b2[d - 256]);
    for (int d = tid; d < 512; d += 256) {
        const int ln = d & 63;
        {   // A operand: row = out channel cb * 32 + (ln & 31), k-slots = in channels 16 ks + 8 (ln >> 5) + j
            const int ks = (d >> 6) & 3, cb = 2 * chh + (d >> 8);
            const float* src = w2 + (cb * 32 + (ln & 31)) * 64 + 16 * ks + 8 * (ln >> 5);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[j];
            u32x4 ph, pm, pl;
            pn_split3_8(v, ph, pm, pl);
            w2a[d] = ph; w2a[512 + d] = pm; w2a[1024 + d] = pl;
        }
        {   // B operand: column = k1 = kt * 32 + (ln & 31), k-slot (hh, j) = ch2 = cb * 32 + 16 half8 + 4 hh + (j & 3) + 8 (j >> 2): the D layout of Z2
            const int kt = (d >> 6) & 1, half8 = (d >> 7) & 1, cb = 2 * chh + (d >> 8), hh = ln >> 5;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w2[(cb * 32 + 16 * half8 + 4 * hh + (j & 3) + 8 * (j >> 2)) * 64 + kt * 32 + (ln & 31)];
            u32x4 ph, pm, pl;
            pn_split3_8(v, ph, pm, pl);
            w2b[d] = ph; w2b[512 + d] = pm; w2b[1024 + d] = pl;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave0 = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wave = tg * 4 + wave0;                         // this wave's winner-row tile, fixed for the whole kernel
    float* tr = trs + wave0 * 32 * TRS;
    float gw3a[32];                 // [ci][r] : gW3[wave*32 + l31][(2 chh + ci)*32 + mfma32_row(r, h)]
#pragma unroll
    for (int i = 0; i < 32; ++i) gw3a[i] = 0.f;
    f32x16 gw2a[4];                 // [ci][kt] : gW2[(2 chh + ci)*32 + row(r,h)][kt*32 + l31], running fp32 sums (VALU adds of each tile's fresh products)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) gw2a[i][r] = 0.f;
    float gb3a = 0.f, gb2a[2] = {0.f, 0.f};
    float gw1a[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, gb1a[2] = {0.f, 0.f};

    const int obj0 = (int)blockIdx.x >> 2, ostep = ((int)gridDim.x + 3) >> 2;
    const int n_iter = obj0 < T ? (T - obj0 + ostep - 1) / ostep : 0;
    float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, ng = 0.f;
    if (n_iter > 0) {
        const size_t ri = (size_t)obj0 * C3 + wave * 32 + l31;
        const int p0 = min(max(argmax[ri], 0), P - 1);
        const float* xp = x + ((size_t)obj0 * P + p0) * 3;
        nx0 = xp[0]; nx1 = xp[1]; nx2 = xp[2];
        ng = y[ri] > 0.f ? gy[ri] : 0.f;
    }
    for (int it = 0; it < n_iter; ++it) {
        int lane_o = lane, h_o = h;
        asm volatile("" : "+v"(lane_o), "+v"(h_o));          // keep weight reads inside the loop (see fwd)
        const int l31_o = lane_o & 31;
        const float x0 = nx0, x1 = nx1, x2 = nx2, g = ng;
        const int tn = obj0 + min(it + 1, n_iter - 1) * ostep;
        const size_t rin = (size_t)tn * C3 + wave * 32 + l31_o;
        const int pn_raw = argmax[rin];
        const float yn = y[rin], gyn = gy[rin];
        if (h == 0 && chh == 0) gb3a += g;

        // ---- H1, lane = row: this lane's 32 channels k = 16 ks + 8 h + j, split for the MFMA B operand (as the forward's layer 1)
        u32x4 h1p[3][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 16 * ks + 8 * h_o;
            float v[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kk = k + 4 * half;
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w1s + kk * 3);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w1s + kk * 3 + 4);
                const f32x4 wc = *reinterpret_cast<const f32x4*>(w1s + kk * 3 + 8);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + kk);
                v[4 * half + 0] = fmaxf(fmaf(wa[2], x2, fmaf(wa[1], x1, fmaf(wa[0], x0, bb[0]))), 0.f);
                v[4 * half + 1] = fmaxf(fmaf(wb[1], x2, fmaf(wb[0], x1, fmaf(wa[3], x0, bb[1]))), 0.f);
                v[4 * half + 2] = fmaxf(fmaf(wc[0], x2, fmaf(wb[3], x1, fmaf(wb[2], x0, bb[2]))), 0.f);
                v[4 * half + 3] = fmaxf(fmaf(wc[3], x2, fmaf(wc[2], x1, fmaf(wc[1], x0, bb[3]))), 0.f);
            }
            pn_split3_8(v, h1p[0][ks], h1p[1][ks], h1p[2][ks]);
        }
        // ---- H1, lane = k1 / regs = rows, from two K = 2 fp32 MFMAs per 32 channels (as the kernel above), and its planes as B operand of gW2
        // (the fp32 values are not kept: the ReLU mask of dZ1 at the end is read off the h plane -- bf16(v) != 0 exactly when v > 0 for v >= 0)
        u32x4 h1cp[3][2][2];                                 // [plane][kt][half8]: k-slot (h, j) = row mfma32_row(half8 * 8 + j, h)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int k1 = kt * 32 + l31_o;
            const float bA = w1s[k1 * 3 + h_o];
            const float bB = h_o ? b1s[k1] : w1s[k1 * 3 + 2];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h_o ? x1 : x0, bA, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h_o ? 1.f : x2, bB, acc, 0, 0, 0);
#pragma unroll
            for (int half8 = 0; half8 < 2; ++half8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(acc[half8 * 8 + j], 0.f);
                pn_split3_8(v, h1cp[0][kt][half8], h1cp[1][kt][half8], h1cp[2][kt][half8]);
            }
        }
        {   // the next tile's point: its index arrived during the H1 section
            const int pn = min(max(pn_raw, 0), P - 1);
            const float* xpn = x + ((size_t)tn * P + pn) * 3;
            nx0 = xpn[0]; nx1 = xpn[1]; nx2 = xpn[2];
            ng = yn > 0.f ? gyn : 0.f;
        }

        f32x16 dh1[2], dh1s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dh1[kt][r] = 0.f; dh1s[kt][r] = 0.f; }
        // Z2[row = lane][ch2 = cb*32 + mfma32_row(r, h)] of BOTH channel blocks: the h h products start at the bias, the small ones at zero; products
        // and their order per accumulator as in the forward kernel's layer 2 (the same bits), the two blocks' chains interleaved so that two
        // consecutive MFMAs never share an accumulator
        f32x16 z2a[2], z2s[2];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + (2 * chh + ci) * 32 + 8 * gq + 4 * h_o);
                z2a[ci][gq * 4 + 0] = bb[0]; z2a[ci][gq * 4 + 1] = bb[1]; z2a[ci][gq * 4 + 2] = bb[2]; z2a[ci][gq * 4 + 3] = bb[3];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) z2s[ci][r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 wh[2], wm[2], wl[2];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int d = (ci * 4 + ks) * 64 + lane_o;
                wh[ci] = w2a[d]; wm[ci] = w2a[512 + d]; wl[ci] = w2a[1024 + d];
            }
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) z2s[ci] = mfma_bf16(wl[ci], h1p[0][ks], z2s[ci]);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) z2s[ci] = mfma_bf16(wh[ci], h1p[2][ks], z2s[ci]);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) z2s[ci] = mfma_bf16(wm[ci], h1p[1][ks], z2s[ci]);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) z2a[ci] = mfma_bf16(wh[ci], h1p[0][ks], z2a[ci]);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) z2s[ci] = mfma_bf16(wm[ci], h1p[0][ks], z2s[ci]);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) z2s[ci] = mfma_bf16(wh[ci], h1p[1][ks], z2s[ci]);
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            const f32x16 acc = z2a[ci], accs = z2s[ci];
            // dZ2 = g * W3[c][ch2] * (Z2 > 0); gW3 partials; the tile goes to LDS for the transpose
            float dz[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 w3v = *reinterpret_cast<const f32x4*>(w3s + (wave0 * 32 + l31_o) * W3S + ci * 32 + 8 * gq + 4 * h_o);
                f32x4 dzv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = acc[gq * 4 + r] + accs[gq * 4 + r];
                    gw3a[ci * 16 + gq * 4 + r] = fmaf(g, fmaxf(z, 0.f), gw3a[ci * 16 + gq * 4 + r]);
                    dzv[r] = (z > 0.f ? g : 0.f) * w3v[r];
                    dz[gq * 4 + r] = dzv[r];
                }
                *reinterpret_cast<f32x4*>(tr + l31_o * TRS + 8 * gq + 4 * h_o) = dzv;       // tr[row][ch2 local]
            }
            // dH1[row][k1] += sum_ch2 dZ2[row][ch2] W2[ch2][k1]: the D registers of Z2 are the A operand's k-slots (two K = 16 steps)
#pragma unroll
            for (int half8 = 0; half8 < 2; ++half8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = dz[half8 * 8 + j];
                u32x4 ah, am, al;
                pn_split3_8(v, ah, am, al);
                // (the two k1 halves alternate: two consecutive MFMAs never share an accumulator -- an LDS read or a VALU instruction between two
                //  dependent MFMAs breaks their back-to-back forwarding, +43 cycles each, and a wave alone on its SIMD has nobody to hide it)
                u32x4 wh[2], wm[2], wl[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int d = ((ci * 2 + half8) * 2 + kt) * 64 + lane_o;
                    wh[kt] = w2b[d]; wm[kt] = w2b[512 + d]; wl[kt] = w2b[1024 + d];
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) dh1s[kt] = mfma_bf16(al, wh[kt], dh1s[kt]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) dh1s[kt] = mfma_bf16(ah, wl[kt], dh1s[kt]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) dh1s[kt] = mfma_bf16(am, wm[kt], dh1s[kt]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) dh1[kt] = mfma_bf16(ah, wh[kt], dh1[kt]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) dh1s[kt] = mfma_bf16(am, wh[kt], dh1s[kt]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) dh1s[kt] = mfma_bf16(ah, wm[kt], dh1s[kt]);
            }
            // lane = ch2, regs = rows: gb2 and gW2 += dZ2^T H1   (wave-private tile: the waitcnt of the reads orders them after this wave's
            // own writes; no barrier)
            float colsum = 0.f;
            float dzt[16];
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                dzt[s_] = tr[mfma32_row(s_, h_o) * TRS + l31_o];
                colsum += dzt[s_];
            }
            gb2a[ci] += colsum;
            u32x4 tp[3][2];
#pragma unroll
            for (int half8 = 0; half8 < 2; ++half8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = dzt[half8 * 8 + j];
                pn_split3_8(v, tp[0][half8], tp[1][half8], tp[2][half8]);
            }
            {
                f32x16 ga[2], gs_[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ga[kt][r] = 0.f; gs_[kt][r] = 0.f; }
#pragma unroll
                for (int half8 = 0; half8 < 2; ++half8) {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) gs_[kt] = mfma_bf16(tp[2][half8], h1cp[0][kt][half8], gs_[kt]);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) gs_[kt] = mfma_bf16(tp[0][half8], h1cp[2][kt][half8], gs_[kt]);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) gs_[kt] = mfma_bf16(tp[1][half8], h1cp[1][kt][half8], gs_[kt]);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) ga[kt] = mfma_bf16(tp[0][half8], h1cp[0][kt][half8], ga[kt]);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) gs_[kt] = mfma_bf16(tp[1][half8], h1cp[0][kt][half8], gs_[kt]);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) gs_[kt] = mfma_bf16(tp[0][half8], h1cp[1][kt][half8], gs_[kt]);
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) gw2a[ci * 2 + kt][r] += ga[kt][r] + gs_[kt][r];
            }
        }
        // ---- dZ1 = (this half's part of dH1) * (H1 > 0) in the lane = k1 layout; gW1 / gb1 (x of row(s,h) by lane shuffle)
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const int src = mfma32_row(s_, h);
            const float sx0 = __shfl(x0, src, 64), sx1 = __shfl(x1, src, 64), sx2 = __shfl(x2, src, 64);
            const unsigned m0 = h1cp[0][0][s_ >> 3][(s_ & 7) >> 1], m1 = h1cp[0][1][s_ >> 3][(s_ & 7) >> 1];       // pair (j, j + 1) of k-slot j = s_ & 7
            const bool on0 = ((s_ & 1) ? (m0 >> 16) : (m0 & 0xffffu)) != 0, on1 = ((s_ & 1) ? (m1 >> 16) : (m1 & 0xffffu)) != 0;
            const float dz0 = on0 ? dh1[0][s_] + dh1s[0][s_] : 0.f;
            const float dz1 = on1 ? dh1[1][s_] + dh1s[1][s_] : 0.f;
            gw1a[0][0] = fmaf(dz0, sx0, gw1a[0][0]); gw1a[0][1] = fmaf(dz0, sx1, gw1a[0][1]); gw1a[0][2] = fmaf(dz0, sx2, gw1a[0][2]);
            gw1a[1][0] = fmaf(dz1, sx0, gw1a[1][0]); gw1a[1][1] = fmaf(dz1, sx1, gw1a[1][1]); gw1a[1][2] = fmaf(dz1, sx2, gw1a[1][2]);
            gb1a[0] += dz0;
            gb1a[1] += dz1;
        }
    }

    // ---- flush the per-workgroup partials (gw3a's index r is the D register: ch2 = cb*32 + mfma32_row(r, h))
#pragma unroll
    for (int i = 0; i < 32; ++i)
        atomicAdd(gw3 + (size_t)(wave * 32 + l31) * 128 + (2 * chh + (i >> 4)) * 32 + mfma32_row(i & 15, h), gw3a[i]);
    if (h == 0 && chh == 0) atomicAdd(gb3 + wave * 32 + l31, gb3a);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int k1 = kt * 32 + l31;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = gw1a[kt][d] + __shfl_xor(gw1a[kt][d], 32, 64);
            if (h == 0) atomicAdd(gw1 + k1 * 3 + d, v);
        }
        const float vb = gb1a[kt] + __shfl_xor(gb1a[kt], 32, 64);
        if (h == 0) atomicAdd(gb1 + k1, vb);
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const float v = gb2a[ci] + __shfl_xor(gb2a[ci], 32, 64);
        if (h == 0) atomicAdd(gb2 + (2 * chh + ci) * 32 + l31, v);
    }
    // gW2 (this half's 64 rows): combine the 4 waves in LDS (the transpose tiles are dead now), then one global atomic per element
    __syncthreads();
    for (int d = tid; d < 4096; d += 256) trs[d] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            atomicAdd(trs + ((i >> 1) * 32 + mfma32_row(r, h)) * 64 + (i & 1) * 32 + l31, gw2a[i][r]);
    __syncthreads();
    for (int d = tid; d < 4096; d += 256) atomicAdd(gw2 + chh * 4096 + d, trs[d]);
}

}  // namespace

extern "C" int sga_pointnet_bwd(const float* x, const int32_t* argmax, const float* y, const float* gy,
                                const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* w3, float* gw1, float* gb1, float* gw2, float* gb2, float* gw3,
                                float* gb3, int T, int P, int C3, int mode, void* stream) {
    SGA_CHECK_ARG(C3 == 256, "sga_pointnet_bwd: out_size C3=%d unsupported (256 only)", C3);
    SGA_CHECK_ARG(mode == 0 || mode == 4, "sga_pointnet_bwd: mode %d (0 = exact fp32 MFMA, 4 = three exact bf16 planes)", mode);
    SGA_CHECK_ARG(T >= 0 && P >= 1, "sga_pointnet_bwd: bad sizes");
    SGA_CHECK_ARG((T == 0 || (x && argmax && y && gy)) && w1 && b1 && w2 && b2 && w3 && gw1 && gb1 && gw2 && gb2 && gw3 && gb3,
                  "sga_pointnet_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gb1 == gw1 + 192 && gw2 == gb1 + 64 && gb2 == gw2 + 8192 && gw3 == gb2 + 128 && gb3 == gw3 + 32768) {
        hipMemsetAsync(gw1, 0, (192 + 64 + 8192 + 128 + 32768 + 256) * sizeof(float), s);      // one flat buffer (ops.py): one launch
    } else {
        hipMemsetAsync(gw1, 0, 64 * 3 * sizeof(float), s);
        hipMemsetAsync(gb1, 0, 64 * sizeof(float), s);
        hipMemsetAsync(gw2, 0, 128 * 64 * sizeof(float), s);
        hipMemsetAsync(gb2, 0, 128 * sizeof(float), s);
        hipMemsetAsync(gw3, 0, 256 * 128 * sizeof(float), s);
        hipMemsetAsync(gb3, 0, 256 * sizeof(float), s);
    }
    if (T == 0) return SGA_OK;
    const int g2 = 2 * T < sga_num_cus() ? 2 * T : (sga_num_cus() & ~1);      // workgroups come in (tiles 0-3, tiles 4-7) pairs
    if (mode == 4) {
        const int g4 = 4 * T < sga_num_cus() ? 4 * T : (sga_num_cus() & ~3);  // ... here in quadruples: (tile group) x (half of the layer-2 channels)
        const size_t lds_b = (size_t)2 * 3 * 512 * 16 + (4 * 32 * TRS + 128 * W3S + 384) * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_bwd_p3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL(pointnet_bwd_p3_kernel, dim3(g4), dim3(256), lds_b, s, x, argmax, y, gy, w1, b1, w2, b2, w3, gw1, gb1, gw2, gb2, gw3, gb3, T, P);
    } else {
        const size_t lds_f = (3 * 8192 + 384) * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(pointnet_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
        hipLaunchKernelGGL(pointnet_bwd_fused_kernel, dim3(g2), dim3(256), lds_f, s, x, argmax, y, gy, w1, b1, w2, b2, w3, gw1, gb1, gw2, gb2, gw3, gb3, T, P);
    }
    SGA_CHECK_LAUNCH("sga_pointnet_bwd");
    return SGA_OK;
}
