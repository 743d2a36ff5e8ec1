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


def convex_hull_barycenters_batch(point_list):
    """Barycentre of the convex-hull vertices of every object (list of [N_i, 3] numpy arrays), as preprocess.py:93-96 computes
    it per object: cx, cy, cz = mean of hull.points[hull.vertices, 0 / 1 / 2].  One kernel launch filters all objects down to
    their hull candidates; Qhull (scipy, the reference's own dependency) then runs on those only and finds the same vertices."""
    from scipy.spatial import ConvexHull
    sizes = [int(p.shape[0]) for p in point_list]
    off = np.concatenate([[0], np.cumsum(sizes)])
    if off[-1] == 0:
        return np.zeros((len(point_list), 3))
    flat = np.concatenate([np.asarray(p)[:, :3] for p in point_list]).astype(np.float32, copy=False)
    keep, _ = hull_candidate_mask_batch(torch.from_numpy(np.ascontiguousarray(flat)).cuda(), off)
    keep = keep.cpu().numpy()
    out = np.zeros((len(point_list), 3))
    for i, p in enumerate(point_list):
        cand = np.asarray(p)[keep[off[i]:off[i + 1]]]          # candidates in the object's own dtype / values
        hull = ConvexHull(cand)
        v = hull.points[hull.vertices]
        out[i] = (np.mean(v[:, 0]), np.mean(v[:, 1]), np.mean(v[:, 2]))
    return out


def convex_hull_barycenter(obj_pcl):
    """Single-object form: (cx, cy, cz) exactly as preprocess.py:93-96 binds them."""
    c = convex_hull_barycenters_batch([obj_pcl])[0]
    return float(c[0]), float(c[1]), float(c[2])
