// Per-object farthest-point sampling (the step immediately before the hot path).
//
// Replaces utils/point_cloud.py:61-89 (pcl_farthest_sample) as called per object and per resolution by
// preprocessing/scan3r/preprocess.py:96-98: starting from point `start`, npoint times { emit the current farthest
// point; d_i = min(d_i, |x_i - x_far|^2); farthest = first arg-max of d }.  The reference runs this as a NumPy loop in
// fp32 (the .npy vertices are 'f4') with (dx*dx + dy*dy) + dz*dz summed left to right and np.argmax's first-maximum
// rule; the kernel keeps exactly that arithmetic (no FMA contraction, ties -> smaller index), so the sampled INDEX
// sequence is bit-identical.
//
// One workgroup (256 threads) per object.  Objects up to 256*PPT points keep their points and running distances in
// registers (PPT = 8 or 32); larger ones walk global memory with the distances in a caller-provided scratch.  Each
// of the npoint rounds is: per-lane update + arg-max, wave arg-max by DPP-free shuffles, 4-wave arg-max through
// LDS (one barrier; double-buffered slots), broadcast of the winner's coordinates.  The rounds are sequentially
// dependent, so the kernel is latency bound (about 1 us per round); throughput comes from all objects of a scan (and
// all scans of a batch) running concurrently.
#include "sga_common.h"

namespace {

constexpr int FPS_THREADS = 256;

struct Cand { float d; int i; };

__device__ __forceinline__ Cand better(Cand a, Cand b) {          // larger distance; on a tie the smaller index
    return (b.d > a.d || (b.d == a.d && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ float sqdist(float x, float y, float z, float cx, float cy, float cz) {
    const float dx = __fsub_rn(x, cx), dy = __fsub_rn(y, cy), dz = __fsub_rn(z, cz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));   // np.sum(..., -1) order, no FMA
}

// PPT > 0: points + distances in registers.  PPT == 0: global walk, distances in `scratch`.
template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ pts, const int* __restrict__ off,
                                                          const int* __restrict__ start, int npoint,
                                                          int* __restrict__ out, float* __restrict__ scratch,
                                                          const int* __restrict__ work, int nwork) {
    __shared__ float s_d[2][4];
    __shared__ int s_i[2][4];
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int obj = work[w];
    const int p0 = off[obj], n = off[obj + 1] - p0;
    const float* P = pts + (size_t)p0 * 3;
    float* dist = PPT == 0 ? scratch + p0 : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float px[PPT > 0 ? PPT : 1], py[PPT > 0 ? PPT : 1], pz[PPT > 0 ? PPT : 1], pd[PPT > 0 ? PPT : 1];
    if (PPT > 0) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * FPS_THREADS + tid;
            const bool ok = i < n;
            px[k] = ok ? P[(size_t)i * 3 + 0] : 0.f;
            py[k] = ok ? P[(size_t)i * 3 + 1] : 0.f;
            pz[k] = ok ? P[(size_t)i * 3 + 2] : 0.f;
            pd[k] = ok ? 1e10f : -1.f;                       // np.ones(N) * 1e10 (exact in fp32); padding can never win
        }
    } else {
        for (int i = tid; i < n; i += FPS_THREADS) dist[i] = 1e10f;
    }
    int far = start[obj];
    for (int r = 0; r < npoint; ++r) {
        if (tid == 0) out[(size_t)obj * npoint + r] = far;
        const float cx = P[(size_t)far * 3 + 0], cy = P[(size_t)far * 3 + 1], cz = P[(size_t)far * 3 + 2];
        Cand best{-2.f, 0x7fffffff};
        if (PPT > 0) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const float d = sqdist(px[k], py[k], pz[k], cx, cy, cz);
                if (d < pd[k]) pd[k] = d;                    // mask = dist < distance; padding (-1) never updates
                best = better(best, Cand{pd[k], k * FPS_THREADS + tid});
            }
        } else {
            for (int i = tid; i < n; i += FPS_THREADS) {
                const float d = sqdist(P[(size_t)i * 3 + 0], P[(size_t)i * 3 + 1], P[(size_t)i * 3 + 2], cx, cy, cz);
                float cur = dist[i];
                if (d < cur) { cur = d; dist[i] = d; }
                best = better(best, Cand{cur, i});
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            Cand other{__shfl_xor(best.d, o, 64), __shfl_xor(best.i, o, 64)};
            best = better(best, other);
        }
        const int sl = r & 1;                                // double-buffered slots: one barrier per round
        if (lane == 0) { s_d[sl][wave] = best.d; s_i[sl][wave] = best.i; }
        __syncthreads();
        Cand b{s_d[sl][0], s_i[sl][0]};
#pragma unroll
        for (int k = 1; k < 4; ++k) b = better(b, Cand{s_d[sl][k], s_i[sl][k]});
        far = b.i;
    }
}

}  // namespace

extern "C" size_t sga_fps_scratch_floats(int total_points) { return total_points > 0 ? (size_t)total_points : 1; }

// work lists: objects are bucketed on the host side of the C ABI by size; `work_*` are device int32 arrays of object ids.
extern "C" int sga_fps(const float* pts, const int32_t* offsets, int n_obj, const int32_t* start, int npoint,
                       const int32_t* work_small, int n_small, const int32_t* work_mid, int n_mid,
                       const int32_t* work_large, int n_large, int32_t* out_idx, float* scratch, void* stream) {
    SGA_CHECK_ARG(pts && offsets && start && out_idx, "sga_fps: null pointer");
    SGA_CHECK_ARG(n_obj >= 0 && npoint >= 1 && n_small >= 0 && n_mid >= 0 && n_large >= 0, "sga_fps: bad sizes");
    SGA_CHECK_ARG(n_small + n_mid + n_large == n_obj, "sga_fps: work lists (%d+%d+%d) do not cover the %d objects",
                  n_small, n_mid, n_large, n_obj);
    SGA_CHECK_ARG((n_small == 0 || work_small) && (n_mid == 0 || work_mid) && (n_large == 0 || (work_large && scratch)),
                  "sga_fps: missing work list / scratch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_small) hipLaunchKernelGGL(fps_kernel<8>, dim3(n_small), dim3(FPS_THREADS), 0, s, pts, offsets, start, npoint, out_idx, scratch, work_small, n_small);
    if (n_mid) hipLaunchKernelGGL(fps_kernel<32>, dim3(n_mid), dim3(FPS_THREADS), 0, s, pts, offsets, start, npoint, out_idx, scratch, work_mid, n_mid);
    if (n_large) hipLaunchKernelGGL(fps_kernel<0>, dim3(n_large), dim3(FPS_THREADS), 0, s, pts, offsets, start, npoint, out_idx, scratch, work_large, n_large);
    SGA_CHECK_LAUNCH("sga_fps");
    return SGA_OK;
}
