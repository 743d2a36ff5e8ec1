// Convex-hull candidate filter for the per-object hull barycentre of the preprocessing step in front of the path
// (reference preprocessing/scan3r/preprocess.py:93-96: hull = scipy.spatial.ConvexHull(obj_pcl); barycentre = mean of
// hull.points[hull.vertices]).
//
// Qhull is O(N log N) per object on the host and most of an object's points are interior.  This kernel marks, for every object
// of a batch in one launch, the points that CAN be hull vertices; the hull of the survivors is the hull of the object (interior
// points never are vertices and removing them does not change the hull), so the host runs Qhull on a few per cent of the points
// and gets the same vertex set.  Per object (one workgroup):
//   1. support points of 26 directions (the +-axes, the 12 edge and the 8 corner diagonals of the cube);
//   2. the facets of THEIR convex hull P (<= 26 vertices, P is inside the object's hull): every triple of support points whose
//      plane has all 26 on one side is a supporting plane of P;
//   3. a point strictly inside every facet half-space, by a margin of 1e-5 of the object's extent, is interior to P, hence to the
//      hull: discarded.  Everything else is kept.
// Soundness (a hull vertex is never discarded) needs EVERY facet of P in the list -- a missing facet enlarges the tested region
// beyond P -- so nothing is ever dropped from it: coplanar triples all stay (no de-duplication), and an object with a triple of
// three DISTINCT but nearly collinear support points (a facet whose plane fp32 cannot determine) keeps all its points, as do
// flat / collinear objects, objects with < 4 planes and objects whose list overflows.
// fp32 throughout, on coordinates translated to the object's first support point (the rounding of a plane evaluation then scales
// with the object's extent, not with its distance from the origin); the margin adds 16 ulp of the largest raw coordinate for
// the rounding of that translation itself.
#include "sga_common.h"

namespace {

constexpr int HU_THREADS = 256;
constexpr int HU_NDIR = 13;                  // direction pairs: +d gives the max, -d the min support
constexpr int HU_NSUP = 26;
constexpr int HU_MAXPLANES = 512;

__constant__ float hu_dirs[HU_NDIR][3] = {
    {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
    {1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1},
    {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}};

__global__ __launch_bounds__(HU_THREADS) void hull_candidates_kernel(const float* __restrict__ pts, const int* __restrict__ offsets,
                                                                     unsigned char* __restrict__ keep, int* __restrict__ n_planes_out) {
    __shared__ float s_val[HU_NSUP][HU_THREADS / 64];
    __shared__ int s_idx[HU_NSUP][HU_THREADS / 64];
    __shared__ float s_sup[HU_NSUP][3];
    __shared__ float s_plane[HU_MAXPLANES][4];
    __shared__ int s_np, s_overflow;
    __shared__ float s_cmax[HU_THREADS / 64];
    const int obj = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = offsets[obj], n = offsets[obj + 1] - p0;
    const float* __restrict__ P = pts + (size_t)p0 * 3;
    unsigned char* __restrict__ K = keep + p0;
    if (n <= 0) return;

    // ---- 1. support points: per-thread best over its strided points, wave reduce, then across waves
    float bv[HU_NSUP];
    int bi[HU_NSUP];
#pragma unroll
    for (int k = 0; k < HU_NSUP; ++k) { bv[k] = -INFINITY; bi[k] = 0; }
    float cmax = 0.f;
    for (int i = tid; i < n; i += HU_THREADS) {
        const float x = P[3 * i], y = P[3 * i + 1], z = P[3 * i + 2];
        cmax = fmaxf(cmax, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
#pragma unroll
        for (int d = 0; d < HU_NDIR; ++d) {
            const float v = hu_dirs[d][0] * x + hu_dirs[d][1] * y + hu_dirs[d][2] * z;
            if (v > bv[2 * d]) { bv[2 * d] = v; bi[2 * d] = i; }
            if (-v > bv[2 * d + 1]) { bv[2 * d + 1] = -v; bi[2 * d + 1] = i; }
        }
    }
#pragma unroll
    for (int k = 0; k < HU_NSUP; ++k) {
        float v = bv[k]; int ix = bi[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64); const int oi = __shfl_xor(ix, o, 64);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if (lane == 0) { s_val[k][wave] = v; s_idx[k][wave] = ix; }
    }
    cmax = wave_max(cmax);
    if (lane == 0) s_cmax[wave] = cmax;
    if (tid == 0) { s_np = 0; s_overflow = 0; }
    __syncthreads();
    if (tid < HU_NSUP) {
        float v = s_val[tid][0]; int ix = s_idx[tid][0];
        for (int w = 1; w < HU_THREADS / 64; ++w)
            if (s_val[tid][w] > v || (s_val[tid][w] == v && s_idx[tid][w] < ix)) { v = s_val[tid][w]; ix = s_idx[tid][w]; }
        s_sup[tid][0] = P[3 * ix]; s_sup[tid][1] = P[3 * ix + 1]; s_sup[tid][2] = P[3 * ix + 2];
    }
    __syncthreads();
    // object extent (from the axis supports) sets the tolerances
    const float ext = fmaxf(fmaxf(s_sup[0][0] - s_sup[1][0], s_sup[2][1] - s_sup[3][1]), s_sup[4][2] - s_sup[5][2]);
    float cm = s_cmax[0];
#pragma unroll
    for (int w = 1; w < HU_THREADS / 64; ++w) cm = fmaxf(cm, s_cmax[w]);
    const float tol = 1e-6f * ext, margin = 1e-5f * ext + 16.f * 1.1920929e-7f * cm;
    const float ox = s_sup[0][0], oy = s_sup[0][1], oz = s_sup[0][2];          // local origin: the +x support point

    // ---- 2. supporting planes of the 26-point polytope: triples (i < j < k) with every support point on one side
    for (int t = tid; t < HU_NSUP * HU_NSUP * HU_NSUP; t += HU_THREADS) {
        const int i = t / (HU_NSUP * HU_NSUP), j = (t / HU_NSUP) % HU_NSUP, k = t % HU_NSUP;
        if (!(i < j && j < k)) continue;
        const float ax = s_sup[i][0], ay = s_sup[i][1], az = s_sup[i][2];
        const float ux = s_sup[j][0] - ax, uy = s_sup[j][1] - ay, uz = s_sup[j][2] - az;
        const float vx = s_sup[k][0] - ax, vy = s_sup[k][1] - ay, vz = s_sup[k][2] - az;
        float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
        const float nn = sqrtf(nx * nx + ny * ny + nz * nz);
        if (!(nn > 1e-4f * ext * ext)) {                               // (nearly) collinear triple: no plane fp32 can determine
            const float uu = ux * ux + uy * uy + uz * uz, vv = vx * vx + vy * vy + vz * vz;
            const float wx = vx - ux, wy = vy - uy, wz = vz - uz, ww = wx * wx + wy * wy + wz * wz;
            // three DISTINCT points: it may be a sliver facet of P that the list would miss -> keep every point of the object.
            // (Triples with a repeated support point -- one point is the support of several directions -- are no facets at all.)
            if (uu > tol * tol && vv > tol * tol && ww > tol * tol) s_overflow = 1;
            continue;
        }
        nx /= nn; ny /= nn; nz /= nn;
        float lo = 0.f, hi = 0.f;
        for (int q = 0; q < HU_NSUP; ++q) {
            const float dq = nx * (s_sup[q][0] - ax) + ny * (s_sup[q][1] - ay) + nz * (s_sup[q][2] - az);
            lo = fminf(lo, dq); hi = fmaxf(hi, dq);
        }
        if (hi > tol && lo < -tol) continue;                           // points on both sides: not a supporting plane
        if (hi > tol) { nx = -nx; ny = -ny; nz = -nz; }                // orient the normal outwards (all points at n.x <= d)
        const float d = nx * (ax - ox) + ny * (ay - oy) + nz * (az - oz);     // plane offset in translated coordinates
        const int slot = atomicAdd(&s_np, 1);
        if (slot < HU_MAXPLANES) { s_plane[slot][0] = nx; s_plane[slot][1] = ny; s_plane[slot][2] = nz; s_plane[slot][3] = d; }
        else s_overflow = 1;
    }
    __syncthreads();
    int np = min(s_np, HU_MAXPLANES);
    __syncthreads();
    // a proper polytope has >= 4 facets; anything else (flat / degenerate object, plane list overflow) keeps every point
    const bool cull = !s_overflow && np >= 4;
    if (tid == 0 && n_planes_out) n_planes_out[obj] = cull ? np : 0;

    // ---- 3. keep a point unless it is strictly inside every facet half-space
    for (int i = tid; i < n; i += HU_THREADS) {
        bool inside = cull;
        if (cull) {
            const float x = P[3 * i] - ox, y = P[3 * i + 1] - oy, z = P[3 * i + 2] - oz;
            for (int q = 0; q < np && inside; ++q)
                inside = s_plane[q][0] * x + s_plane[q][1] * y + s_plane[q][2] * z - s_plane[q][3] < -margin;
        }
        K[i] = inside ? 0 : 1;
    }
}

}  // namespace

extern "C" int sga_hull_candidates(const float* pts, const int32_t* offsets, int n_obj, unsigned char* keep, int32_t* n_planes,
                                   void* stream) {
    SGA_CHECK_ARG(n_obj >= 0, "sga_hull_candidates: bad sizes");
    if (n_obj == 0) return SGA_OK;
    SGA_CHECK_ARG(pts && offsets && keep, "sga_hull_candidates: null pointer");
    hipLaunchKernelGGL(hull_candidates_kernel, dim3(n_obj), dim3(HU_THREADS), 0, static_cast<hipStream_t>(stream), pts, offsets, keep, n_planes);
    SGA_CHECK_LAUNCH("sga_hull_candidates");
    return SGA_OK;
}
