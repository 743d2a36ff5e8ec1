"""CPU restatement of the reference's per-object farthest-point sampling.  TEST INFRASTRUCTURE ONLY (see
oracle/sga_oracle.py's header): imported by tests/, bench tools and __graft_entry__.smoke(), never by the product path.

Follows utils/point_cloud.py:61-89 (pcl_farthest_sample): same dtypes (fp32 points, float64 `distance` array holding
fp32 values), same update rule (`mask = dist < distance`), np.argmax's first-maximum rule.  The one change: the first
sample `start` is an argument instead of `np.random.randint(0, N)` (:77), so the sequence is reproducible.
Pinned against the reference function itself (np.random.randint patched to return `start`) by
oracle/make_golden.py -> tests/golden/fps_*.npz.
"""
import numpy as np


def farthest_point_sample_idx(point: np.ndarray, npoint: int, start: int) -> np.ndarray:
    """point [N, D>=3] float32, N >= npoint.  Returns the npoint sampled indices (int32)."""
    N = point.shape[0]
    assert N >= npoint, 'the N < npoint branch of the reference is a random draw with replacement (point_cloud.py:70-73)'
    xyz = point[:, :3]
    centroids = np.zeros((npoint,))
    distance = np.ones((N,)) * 1e10
    farthest = int(start)
    for i in range(npoint):
        centroids[i] = farthest
        centroid = xyz[farthest, :]
        dist = np.sum((xyz - centroid) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = np.argmax(distance, -1)
    return centroids.astype(np.int32)
