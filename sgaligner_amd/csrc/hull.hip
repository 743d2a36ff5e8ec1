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

// ------------------------------------------------------------------------------------------------------------------------------
// Hull VERTICES on the device (round 3): gift wrapping on the survivors of the filter above, in fp64, with a certificate.
//
// One workgroup per object (candidates in LDS, translated to their first point; bitwise copies of an earlier point -- scans hold exact
// duplicates -- take no part and are never flagged).  The hull is wrapped facet by facet: for a
// directed edge (u, v) whose other facet (v, u, w) is known, the third vertex of this side is the point d that every other point lies
// behind -- found by a Jarvis scan "d <- q whenever orient(u, v, d, q) > 0" started at w (all points lie within the half-turn from
// that facet, so the comparator is a total order), every lane over its points, then a tree reduction with the same comparator.  The
// first edge comes from the xy-projection (lowest-x point, then the 2-D Jarvis step), the first facet from a scan started at the
// vertical plane through it.  Directed edges are kept in an LDS hash set; each open edge is wrapped once.
// Certificate (what makes the result EXACT, not just plausible): after wrapping, (i) every directed edge has its reverse (a closed
// surface), (ii) F = 2 V - 4, and (iii) every point that is not one of a facet's three vertices lies behind that facet's plane by more than 1e-9 of max(extent, |coordinate|), five orders above both the
// fp64 rounding of the test and Qhull's own coplanarity tolerance.  Then the facets are the boundary of the convex hull and their
// vertices are its vertices, the set Qhull returns.  Anything else -- fewer than 4 points, more than HG_MAXN candidates, coplanar or
// nearly coplanar points on a facet (lattices, flat objects), a failed check -- sets status != 0 and the caller runs Qhull on that
// object's candidates (utils/point_cloud.py), so the answer is the reference's in every case.
namespace {

constexpr int HG_THREADS = 256;
constexpr int HG_MAXN = 512;                     // candidates per object handled here (V <= 512 -> F <= 1020, 3060 directed edges)
constexpr int HG_MAXF = 2 * HG_MAXN;
constexpr int HG_HASH = 8192;                    // open-addressing set of directed edges (key = u * 1024 + v + 1)

struct HgLds {
    double x[HG_MAXN], y[HG_MAXN], z[HG_MAXN];
    int hash[HG_HASH];
    short fa[HG_MAXF], fb[HG_MAXF], fc[HG_MAXF];          // facets
    short qu[3 * HG_MAXF], qv[3 * HG_MAXF], qw[3 * HG_MAXF];   // open edges (u, v) with the known facet's third vertex w
    int red[HG_THREADS];
    int n_f, q_head, q_tail, fail, scal[4];
    unsigned char isv[HG_MAXN];
    unsigned char dup[HG_MAXN];                      // 1: a bitwise copy of an earlier point (scans hold exact duplicates): takes no part
};

__device__ __forceinline__ double hg_orient(const HgLds& l, int a, int b, int c, int q) {      // ((b-a) x (c-a)) . (q-a)
    const double ux = l.x[b] - l.x[a], uy = l.y[b] - l.y[a], uz = l.z[b] - l.z[a];
    const double vx = l.x[c] - l.x[a], vy = l.y[c] - l.y[a], vz = l.z[c] - l.z[a];
    const double wx = l.x[q] - l.x[a], wy = l.y[q] - l.y[a], wz = l.z[q] - l.z[a];
    return (uy * vz - uz * vy) * wx + (uz * vx - ux * vz) * wy + (ux * vy - uy * vx) * wz;
}

__device__ __forceinline__ bool hg_insert(HgLds& l, int u, int v) {      // one thread; false if (u, v) was already there
    const int key = u * 1024 + v + 1;
    unsigned h = ((unsigned)key * 2654435761u) >> 19;                     // 13 bits
    for (int probe = 0; probe < HG_HASH; ++probe) {
        const int cur = l.hash[h];
        if (cur == key) return false;
        if (cur == 0) { l.hash[h] = key; return true; }
        h = (h + 1) & (HG_HASH - 1);
    }
    return false;
}
__device__ __forceinline__ bool hg_has(const HgLds& l, int u, int v) {
    const int key = u * 1024 + v + 1;
    unsigned h = ((unsigned)key * 2654435761u) >> 19;
    for (int probe = 0; probe < HG_HASH; ++probe) {
        const int cur = l.hash[h];
        if (cur == key) return true;
        if (cur == 0) return false;
        h = (h + 1) & (HG_HASH - 1);
    }
    return false;
}

// The point every other point lies behind, for the directed edge (u, v), scanning from `start` (>= 0: a point index; -1: the vertical
// plane through the edge, for the very first facet).  All threads call it; the result is returned to all.
__device__ int hg_wrap(HgLds& l, int n, int u, int v, int start) {
    const int tid = threadIdx.x;
    int d = start;
    // start < 0 (first facet): the supporting plane is the vertical one through the edge; any point off the edge's line starts the scan
    const double ux = l.x[v] - l.x[u], uy = l.y[v] - l.y[u], uz = l.z[v] - l.z[u];
    auto beats = [&](int cur, int q) -> bool { return hg_orient(l, u, v, cur, q) > 0.0; };     // q in front of the plane (u, v, cur)?
    for (int q = tid; q < n; q += HG_THREADS) {
        if (q == u || q == v || q == d || l.dup[q]) continue;
        if (d < 0) {
            // from the vertical supporting plane every point is "in front or on": take the first point that is not on the edge's line
            const double cx = uy * (l.z[q] - l.z[u]) - uz * (l.y[q] - l.y[u]);
            const double cy = uz * (l.x[q] - l.x[u]) - ux * (l.z[q] - l.z[u]);
            const double cz = ux * (l.y[q] - l.y[u]) - uy * (l.x[q] - l.x[u]);
            if (cx != 0.0 || cy != 0.0 || cz != 0.0) d = q;
            continue;
        }
        if (beats(d, q)) d = q;
    }
    l.red[tid] = d;
    __syncthreads();
    for (int s = HG_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const int a = l.red[tid], b = l.red[tid + s];
            int r = a;
            if (a < 0) r = b;
            else if (b >= 0 && b != a && beats(a, b)) r = b;
            l.red[tid] = r;
        }
        __syncthreads();
    }
    const int res = l.red[0];
    __syncthreads();
    return res;
}

__global__ __launch_bounds__(HG_THREADS) void hull_vertices_kernel(const double* __restrict__ pts, const int* __restrict__ offsets,
                                                                   unsigned char* __restrict__ is_vertex, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hg_raw[];
    HgLds& l = *reinterpret_cast<HgLds*>(hg_raw);
    const int obj = blockIdx.x, tid = threadIdx.x;
    const int p0 = offsets[obj], n = offsets[obj + 1] - p0;
    if (n < 4 || n > HG_MAXN) { if (tid == 0) status[obj] = n < 4 ? 1 : 2; return; }
    // translate to the first point (exact for nearby doubles), find extent and largest coordinate for the tolerance
    const double ox = pts[3 * (size_t)p0], oy = pts[3 * (size_t)p0 + 1], oz = pts[3 * (size_t)p0 + 2];
    double mx = 0.0, ext = 0.0;
    for (int i = tid; i < n; i += HG_THREADS) {
        const double X = pts[3 * (size_t)(p0 + i)], Y = pts[3 * (size_t)(p0 + i) + 1], Z = pts[3 * (size_t)(p0 + i) + 2];
        l.x[i] = X - ox; l.y[i] = Y - oy; l.z[i] = Z - oz;
        l.isv[i] = 0;
        mx = fmax(mx, fmax(fabs(X), fmax(fabs(Y), fabs(Z))));
        ext = fmax(ext, fmax(fabs(X - ox), fmax(fabs(Y - oy), fabs(Z - oz))));
    }
    for (int i = tid; i < HG_HASH; i += HG_THREADS) l.hash[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += HG_THREADS) {                               // O(n^2 / threads): n <= 512
        unsigned char dp = 0;
        for (int j = 0; j < i && !dp; ++j) dp = (l.x[j] == l.x[i] && l.y[j] == l.y[i] && l.z[j] == l.z[i]) ? 1 : 0;
        l.dup[i] = dp;
    }
    // workgroup max of (mx, ext) through the reduction buffer (as bit patterns of non-negative floats: order preserving)
    l.red[tid] = __float_as_int((float)fmax(mx, ext));
    if (tid == 0) { l.n_f = 0; l.q_head = 0; l.q_tail = 0; l.fail = 0; }
    __syncthreads();
    for (int s = HG_THREADS / 2; s > 0; s >>= 1) { if (tid < s) l.red[tid] = max(l.red[tid], l.red[tid + s]); __syncthreads(); }
    const double tol = 1e-9 * (double)__int_as_float(l.red[0]) * 1.0000002;
    __syncthreads();

    // ---- first edge: lowest x (then y, z, index), then the 2-D Jarvis step in the xy-projection
    int best = -1;
    for (int i = tid; i < n; i += HG_THREADS)
        if (!l.dup[i] && (best < 0 || l.x[i] < l.x[best] || (l.x[i] == l.x[best] && (l.y[i] < l.y[best] || (l.y[i] == l.y[best] && l.z[i] < l.z[best]))))) best = i;
    l.red[tid] = best;
    __syncthreads();
    for (int s = HG_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const int a = l.red[tid], b = l.red[tid + s];
            int r = a;
            if (a < 0) r = b;
            else if (b >= 0 && (l.x[b] < l.x[a] || (l.x[b] == l.x[a] && (l.y[b] < l.y[a] || (l.y[b] == l.y[a] && (l.z[b] < l.z[a] || (l.z[b] == l.z[a] && b < a))))))) r = b;
            l.red[tid] = r;
        }
        __syncthreads();
    }
    const int a0 = l.red[0];
    __syncthreads();
    // second vertex: q beats c when it is to the right of a0 -> c in projection (all points end up on the left of a0 -> b0)
    int c2 = -1;
    auto right_of = [&](int c, int q) {
        const double cr = (l.x[c] - l.x[a0]) * (l.y[q] - l.y[a0]) - (l.y[c] - l.y[a0]) * (l.x[q] - l.x[a0]);
        return cr < 0.0;
    };
    for (int q = tid; q < n; q += HG_THREADS) {
        if (q == a0 || l.dup[q] || (l.x[q] == l.x[a0] && l.y[q] == l.y[a0])) continue;           // same projection as a0: no direction
        if (c2 < 0 || right_of(c2, q)) c2 = q;
    }
    l.red[tid] = c2;
    __syncthreads();
    for (int s = HG_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const int a = l.red[tid], b = l.red[tid + s];
            int r = a;
            if (a < 0) r = b;
            else if (b >= 0 && right_of(a, b)) r = b;
            l.red[tid] = r;
        }
        __syncthreads();
    }
    const int b0 = l.red[0];
    __syncthreads();
    if (b0 < 0) { if (tid == 0) status[obj] = 3; return; }                              // every point on one vertical line
    // first facet: wrap around (a0, b0) from the vertical supporting plane
    const int c0 = hg_wrap(l, n, a0, b0, -1);
    if (c0 < 0) { if (tid == 0) status[obj] = 3; return; }
    // hg_wrap from the virtual plane may have picked an arbitrary starting point; run it again from that point to the true extreme
    const int c1 = hg_wrap(l, n, a0, b0, c0);
    if (tid == 0) {
        l.fa[0] = (short)a0; l.fb[0] = (short)b0; l.fc[0] = (short)c1; l.n_f = 1;
        hg_insert(l, a0, b0); hg_insert(l, b0, c1); hg_insert(l, c1, a0);
        // open edges: the reverses, each with the facet's remaining vertex as the start of its scan
        l.qu[0] = (short)b0; l.qv[0] = (short)a0; l.qw[0] = (short)c1;
        l.qu[1] = (short)c1; l.qv[1] = (short)b0; l.qw[1] = (short)a0;
        l.qu[2] = (short)a0; l.qv[2] = (short)c1; l.qw[2] = (short)b0;
        l.q_tail = 3;
    }
    __syncthreads();
    // ---- wrap every open edge
    for (int guard = 0; guard < 3 * HG_MAXF; ++guard) {
        if (l.q_head >= l.q_tail || l.fail) break;
        const int u = l.qu[l.q_head], v = l.qv[l.q_head], w = l.qw[l.q_head];
        const bool done = hg_has(l, u, v);
        __syncthreads();
        if (tid == 0) l.q_head++;
        if (done) { __syncthreads(); continue; }
        const int d = hg_wrap(l, n, u, v, w);
        if (tid == 0) {
            if (d < 0 || d == w || l.n_f >= HG_MAXF || l.q_tail + 2 >= 3 * HG_MAXF) l.fail = 4;
            else {
                const int f = l.n_f++;
                l.fa[f] = (short)u; l.fb[f] = (short)v; l.fc[f] = (short)d;
                if (!hg_insert(l, u, v) || !hg_insert(l, v, d) || !hg_insert(l, d, u)) l.fail = 5;      // an edge wrapped twice
                if (!hg_has(l, d, v)) { const int t = l.q_tail++; l.qu[t] = (short)d; l.qv[t] = (short)v; l.qw[t] = (short)u; }
                if (!hg_has(l, u, d)) { const int t = l.q_tail++; l.qu[t] = (short)u; l.qv[t] = (short)d; l.qw[t] = (short)v; }
            }
        }
        __syncthreads();
    }
    if (l.q_head < l.q_tail && !l.fail) { if (tid == 0) l.fail = 6; }
    __syncthreads();
    const int F = l.n_f;
    // ---- certificate
    int bad = l.fail;
    for (int f = 0; f < F && !bad; ++f) {
        const int a = l.fa[f], b = l.fb[f], c = l.fc[f];
        if (tid == 0) { l.isv[a] = 1; l.isv[b] = 1; l.isv[c] = 1; }
        if (tid < 3) {                                                        // closed surface: every directed edge has its reverse
            const int e0 = tid == 0 ? a : (tid == 1 ? b : c), e1 = tid == 0 ? b : (tid == 1 ? c : a);
            if (!hg_has(l, e1, e0)) bad = 7;
        }
        const double ux = l.x[b] - l.x[a], uy = l.y[b] - l.y[a], uz = l.z[b] - l.z[a];
        const double vx = l.x[c] - l.x[a], vy = l.y[c] - l.y[a], vz = l.z[c] - l.z[a];
        double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
        const double nn = sqrt(nx * nx + ny * ny + nz * nz);
        if (!(nn > 0.0)) { bad = 8; break; }
        nx /= nn; ny /= nn; nz /= nn;
        if (nn < tol * tol) bad = 8;                                          // a sliver facet: its plane is not determined
        for (int q = tid; q < n; q += HG_THREADS) {
            if (q == a || q == b || q == c || l.dup[q]) continue;          // copies stand or fall with the point they repeat
            const double dist = nx * (l.x[q] - l.x[a]) + ny * (l.y[q] - l.y[a]) + nz * (l.z[q] - l.z[a]);
            if (!(dist < -tol)) bad = 9;                                      // on / near / in front of the plane: not certified
        }
    }
    l.red[tid] = bad;
    __syncthreads();
    for (int s = HG_THREADS / 2; s > 0; s >>= 1) { if (tid < s) l.red[tid] = max(l.red[tid], l.red[tid + s]); __syncthreads(); }
    bad = l.red[0];
    __syncthreads();
    if (!bad) {
        int nv = 0;
        for (int i = tid; i < n; i += HG_THREADS) nv += l.isv[i];
        l.red[tid] = nv;
        __syncthreads();
        for (int s = HG_THREADS / 2; s > 0; s >>= 1) { if (tid < s) l.red[tid] += l.red[tid + s]; __syncthreads(); }
        if (F != 2 * l.red[0] - 4) bad = 10;                                  // Euler's relation for a triangulated closed surface
        __syncthreads();
    }
    if (tid == 0) status[obj] = bad;
    if (!bad) for (int i = tid; i < n; i += HG_THREADS) is_vertex[p0 + i] = l.isv[i];
}

}  // namespace

extern "C" int sga_hull_max_candidates(void) { return HG_MAXN; }

extern "C" int sga_hull_vertices(const double* pts, const int32_t* offsets, int n_obj, unsigned char* is_vertex, int32_t* status, void* stream) {
    SGA_CHECK_ARG(n_obj >= 0, "sga_hull_vertices: bad sizes");
    if (n_obj == 0) return SGA_OK;
    SGA_CHECK_ARG(pts && offsets && is_vertex && status, "sga_hull_vertices: null pointer");
    hipFuncSetAttribute(reinterpret_cast<const void*>(hull_vertices_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(HgLds));
    hipLaunchKernelGGL(hull_vertices_kernel, dim3(n_obj), dim3(HG_THREADS), sizeof(HgLds), static_cast<hipStream_t>(stream), pts, offsets, is_vertex, status);
    SGA_CHECK_LAUNCH("sga_hull_vertices");
    return SGA_OK;
}
