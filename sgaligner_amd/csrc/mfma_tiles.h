// LDS-tile + fp32-MFMA building blocks shared by gemm.hip and contrastive.hip (gfx950).
//
// Conventions
//   * LDS operand tiles are stored [row][k] with a row stride of (KC + 4) floats, KC a multiple of 8:
//     (KC+4)/4 is odd, so the 16 rows a ds_read_b128 lane group touches land on 16 distinct 16-byte
//     slots of the 256-byte bank row -> conflict-free ds_read_b128.
//   * v_mfma_f32_32x32x2_f32 K-steps are issued in groups of 4: for group q the lower lane half
//     (h = 0) supplies k = 8q .. 8q+3 and the upper half k = 8q+4 .. 8q+7, so one ds_read_b128 per
//     operand feeds 4 MFMAs.  Only the summation order over k depends on this choice.
//   * C/D layout of the 32x32 tile: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma once
#include "sga_common.h"

#define SGA_KC 32                 // K-chunk staged per barrier
#define SGA_LDS_STRIDE (SGA_KC + 4)

// Copy rows [row0, row0+NROWS) x cols [k0, k0+SGA_KC) of a row-major fp32 matrix (leading dim ld,
// `nrows` valid rows, `ncols` valid cols, ncols % 4 == 0, base 16-byte aligned, ld % 4 == 0) into an
// LDS tile [NROWS][SGA_LDS_STRIDE]; out-of-range elements are zero-filled.
template <int NROWS, int NTHREADS>
__device__ __forceinline__ void lds_load_rows(float* __restrict__ tile, const float* __restrict__ g, int ld,
                                              int row0, int nrows, int k0, int ncols, int tid) {
    constexpr int V = SGA_KC / 4;                       // float4 per row
#pragma unroll
    for (int e = tid; e < NROWS * V; e += NTHREADS) {
        const int r = e / V, c = (e % V) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int gr = row0 + r, gc = k0 + c;
        if (gr < nrows && gc < ncols) v = *reinterpret_cast<const f32x4*>(g + (size_t)gr * ld + gc);
        *reinterpret_cast<f32x4*>(tile + r * SGA_LDS_STRIDE + c) = v;
    }
}

// acc[t] (t < NT) += A_t * B^T over one staged K-chunk, where the MFMA "A" operand rows come from
// a_tile rows t*32 + (lane&31) and the "B" operand rows from b_rows (this lane's row, already offset).
// Result layout: col (lane&31) <-> the B row, row(r,h) <-> A row t*32 + row.
template <int NT>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[NT], const float* __restrict__ a_tile,
                                           const float* __restrict__ b_row, int lane) {
    const int h4 = (lane >> 5) * 4;
    const float* ap = a_tile + (lane & 31) * SGA_LDS_STRIDE + h4;
    const float* bp = b_row + h4;
#pragma unroll
    for (int q = 0; q < SGA_KC / 8; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 8 * q);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + t * 32 * SGA_LDS_STRIDE + 8 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc[t], 0, 0, 0);
        }
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}
