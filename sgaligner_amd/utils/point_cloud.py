"""Host-side mirror of the reference's utils/point_cloud.py sampling helpers, backed by the HIP FPS kernel.

`pcl_farthest_sample(point, npoint, return_idxs)` keeps the reference signature and semantics (utils/point_cloud.py:
61-89): N < npoint -> random draw with replacement on the host (np.random.choice, :70-73); otherwise the first sample is
drawn with np.random.randint(0, N) (:77) and the remaining ones come from csrc/fps.hip (bit-identical index sequence).
`farthest_point_sample_batch` is the batched form preprocessing wants: all objects of a scan (or of many scans) in
one launch."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..ops import _p, _stream

SMALL_MAX, MID_MAX = 2048, 8192          # register-resident kernel variants (8 / 32 points per lane)


def farthest_point_sample_batch(points, offsets, npoint: int, start) -> torch.Tensor:
    """points [sum N, 3] float32 CUDA tensor (objects packed back to back), offsets [n_obj+1] (host ints or tensor),
    start [n_obj] first sample per object.  Returns idx [n_obj, npoint] int32 (object-local), on the GPU."""
    if not (isinstance(points, torch.Tensor) and points.is_cuda and points.dtype == torch.float32):
        raise RuntimeError('farthest_point_sample_batch: points must be a float32 CUDA tensor (no CPU fallback)')
    pts = points.contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError(f'points must be [N,3], got {tuple(pts.shape)}')
    off = np.asarray(offsets.cpu() if isinstance(offsets, torch.Tensor) else offsets, dtype=np.int64)
    n_obj = len(off) - 1
    sizes = np.diff(off)
    if n_obj < 0 or off[0] != 0 or off[-1] != pts.shape[0] or (sizes < 0).any():
        raise ValueError('offsets must be a monotone prefix array covering all points')
    if (sizes < npoint).any():
        raise ValueError('every object needs N >= npoint (the N < npoint branch is a host-side random draw)')
    st = np.asarray(start.cpu() if isinstance(start, torch.Tensor) else start, dtype=np.int64)
    if st.shape != (n_obj,) or (st < 0).any() or (st >= sizes).any():
        raise ValueError('start must hold one in-range index per object')
    dev = pts.device
    ids = np.arange(n_obj, dtype=np.int32)
    small, mid, large = ids[sizes <= SMALL_MAX], ids[(sizes > SMALL_MAX) & (sizes <= MID_MAX)], ids[sizes > MID_MAX]
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
    d_off, d_start = to_dev(off), to_dev(st)
    d_small, d_mid, d_large = to_dev(small), to_dev(mid), to_dev(large)
    out = torch.empty((n_obj, npoint), device=dev, dtype=torch.int32)
    L = _lib.lib()
    scratch = torch.empty((L.sga_fps_scratch_floats(int(pts.shape[0])) if len(large) else 1,), device=dev, dtype=torch.float32)
    if n_obj:
        rc = L.sga_fps(_p(pts), _p(d_off), n_obj, _p(d_start), npoint, _p(d_small), len(small), _p(d_mid), len(mid),
                       _p(d_large), len(large), _p(out), _p(scratch), _stream())
        _lib.check(rc, 'sga_fps')
    return out


def pcl_farthest_sample(point, npoint, return_idxs=False):
    """Reference signature (utils/point_cloud.py:61): point [N, D] numpy array -> sampled [npoint, D] (and the indices)."""
    N, D = point.shape
    if N < npoint:
        indices = np.random.choice(point.shape[0], npoint)
        return point[indices]
    farthest = np.random.randint(0, N)
    xyz = torch.from_numpy(np.ascontiguousarray(point[:, :3], dtype=np.float32)).cuda()
    idx = farthest_point_sample_batch(xyz, [0, N], npoint, [farthest])[0].cpu().numpy().astype(np.int32)
    if return_idxs:
        return point[idx], idx
    return point[idx]


# ---- convex-hull barycentre (preprocessing/scan3r/preprocess.py:93-96) ---------------------------------------------------
def hull_candidate_mask_batch(points, offsets):
    """points [sum N, 3] float32 CUDA tensor (objects packed back to back), offsets [n_obj+1] host ints.  Returns
    (keep [sum N] bool CUDA tensor, n_planes [n_obj] int32 CUDA tensor): points that can be hull vertices (csrc/hull.hip)."""
    if not (isinstance(points, torch.Tensor) and points.is_cuda and points.dtype == torch.float32):
        raise RuntimeError('hull_candidate_mask_batch: points must be a float32 CUDA tensor (no CPU fallback)')
    pts = points.contiguous()
    off = np.asarray(offsets, dtype=np.int64)
    n_obj = len(off) - 1
    if n_obj < 0 or off[0] != 0 or off[-1] != pts.shape[0] or (np.diff(off) < 0).any():
        raise ValueError('offsets must be a monotone prefix array covering all points')
    d_off = torch.from_numpy(off.astype(np.int32)).to(pts.device)
    keep = torch.empty((max(int(pts.shape[0]), 1),), device=pts.device, dtype=torch.uint8)
    npl = torch.zeros((max(n_obj, 1),), device=pts.device, dtype=torch.int32)
    _lib.check(_lib.lib().sga_hull_candidates(_p(pts), _p(d_off), n_obj, _p(keep), _p(npl), _stream()), 'sga_hull_candidates')
    return keep[:pts.shape[0]].bool(), npl[:n_obj]


HULL_ON_DEVICE = True          # tests flip it to cross-check the device hull against Qhull on the same candidates


def hull_vertices_batch(cand64, offsets):
    """cand64 [sum n, 3] float64 numpy (objects packed back to back), offsets [n_obj+1].  Returns (is_vertex [sum n] bool numpy,
    status [n_obj] int numpy): the device gift-wrapping hull with its certificate (csrc/hull.hip, sga_hull_vertices); status != 0
    marks the objects the caller has to hand to Qhull."""
    off = np.asarray(offsets, dtype=np.int64)
    n_obj = len(off) - 1
    if n_obj <= 0 or off[-1] == 0:
        return np.zeros((int(off[-1]) if len(off) else 0,), dtype=bool), np.ones((max(n_obj, 0),), dtype=np.int32)
    d_pts = torch.from_numpy(np.ascontiguousarray(cand64, dtype=np.float64)).cuda()
    d_off = torch.from_numpy(off.astype(np.int32)).cuda()
    isv = torch.zeros((int(off[-1]),), device='cuda', dtype=torch.uint8)
    status = torch.full((n_obj,), -1, device='cuda', dtype=torch.int32)
    _lib.check(_lib.lib().sga_hull_vertices(_p(d_pts), _p(d_off), n_obj, _p(isv), _p(status), _stream()), 'sga_hull_vertices')
    return isv.cpu().numpy().astype(bool), status.cpu().numpy()


def convex_hull_barycenters_batch(point_list, return_info=False):
    """Barycentre of the convex-hull vertices of every object (list of [N_i, 3] numpy arrays), as preprocess.py:93-96 computes
    it per object: cx, cy, cz = mean of hull.points[hull.vertices, 0 / 1 / 2].  Three steps, no per-object host loop on the common path:
    (1) one launch filters all objects down to their hull candidates (sga_hull_candidates); (2) one launch wraps every object's
    candidates into its hull vertices in fp64, with a certificate (sga_hull_vertices); (3) the vertex means are segment sums.  Objects
    the device declines (fewer than 4 or more than 512 candidates, coplanar / near-degenerate facets: lattices, flat objects) go to
    Qhull -- scipy, the reference's own dependency -- on their candidates, so every answer is the reference's."""
    sizes = [int(p.shape[0]) for p in point_list]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_obj = len(point_list)
    if off[-1] == 0:
        return (np.zeros((n_obj, 3)), {'device': 0, 'qhull': 0}) if return_info else np.zeros((n_obj, 3))
    src = np.concatenate([np.asarray(p)[:, :3] for p in point_list])          # the objects' own values (float32 scans stay float32)
    flat = src if src.dtype == np.float32 else src.astype(np.float32)
    keep, _ = hull_candidate_mask_batch(torch.from_numpy(np.ascontiguousarray(flat)).cuda(), off)
    keep = keep.cpu().numpy()
    # fp32 can merge distinct fp64 points; the filter only ever DISCARDS points that are interior by a margin far above that rounding
    idx = np.flatnonzero(keep)
    counts = np.add.reduceat(keep.astype(np.int64), np.minimum(off[:-1], len(keep) - 1)) * (np.diff(off) > 0)
    coff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    cand = src[idx].astype(np.float64, copy=False)                            # exact: float32 -> float64
    out = np.zeros((n_obj, 3))
    todo = np.ones((n_obj,), dtype=bool)
    if HULL_ON_DEVICE:
        isv, status = hull_vertices_batch(cand, coff)
        ok = status == 0
        if ok.any():
            w = isv.astype(np.float64)
            seg = np.minimum(coff[:-1], max(len(w) - 1, 0))
            nz = counts > 0
            nv = np.where(nz, np.add.reduceat(w, seg), 0.0)
            for c in range(3):
                sm = np.where(nz, np.add.reduceat(cand[:, c] * w, seg), 0.0)
                out[ok, c] = sm[ok] / nv[ok]
            todo = ~ok
    n_q = 0
    if todo.any():
        from scipy.spatial import ConvexHull
        for i in np.flatnonzero(todo):
            p = np.asarray(point_list[i])
            hull = ConvexHull(p[keep[off[i]:off[i + 1]]])          # candidates in the object's own dtype / values
            v = hull.points[hull.vertices]
            out[i] = (np.mean(v[:, 0]), np.mean(v[:, 1]), np.mean(v[:, 2]))
            n_q += 1
    if return_info:
        return out, {'device': int(n_obj - n_q), 'qhull': int(n_q)}
    return out


def convex_hull_barycenter(obj_pcl):
    """Single-object form: (cx, cy, cz) exactly as preprocess.py:93-96 binds them."""
    c = convex_hull_barycenters_batch([obj_pcl])[0]
    return float(c[0]), float(c[1]), float(c[2])
