"""CPU restatement of the reference's per-object convex-hull barycentre.  TEST INFRASTRUCTURE ONLY (see oracle/sga_oracle.py's
header): imported by tests/ and tools, never by the product path.

Follows preprocessing/scan3r/preprocess.py:93-96 line by line:
    hull = ConvexHull(obj_pcl)
    cx = np.mean(hull.points[hull.vertices,0]); cy = ...[:,1]; cz = ...[:,2]
`scipy.spatial.ConvexHull` (Qhull) is the reference's own dependency and is present here, so the restatement IS the reference
arithmetic; the four lines live inside `process_scan` (:40-211), which cannot be imported without the 3RScan files, hence the
restatement.  tests/golden/hull_cases.npz pins it on real example_data objects (oracle/make_golden.py gen_hull)."""
import numpy as np
from scipy.spatial import ConvexHull


def hull_barycenter(obj_pcl: np.ndarray):
    hull = ConvexHull(obj_pcl)
    cx = np.mean(hull.points[hull.vertices, 0])
    cy = np.mean(hull.points[hull.vertices, 1])
    cz = np.mean(hull.points[hull.vertices, 2])
    return np.array([cx, cy, cz]), np.sort(hull.vertices)
