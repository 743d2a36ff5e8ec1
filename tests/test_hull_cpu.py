"""SURVEY 8(f) rank 4, second half: the per-object convex-hull barycentre (preprocessing/scan3r/preprocess.py:93-96).  The oracle
(scipy.spatial.ConvexHull + the reference's three np.mean calls) against the committed vectors generated from the reference's
example scan and from shapes with coplanar / collinear structure."""
import numpy as np

from conftest import load_golden


def test_hull_oracle_golden():
    from oracle import hull_oracle
    g = load_golden('hull_cases')
    n = int(g['n_cases'])
    assert n >= 8
    for k in range(n):
        bc, verts = hull_oracle.hull_barycenter(g[f'pts{k}'])
        assert np.array_equal(verts, g[f'verts{k}']), k
        assert np.allclose(bc, g[f'bc{k}'], rtol=0, atol=1e-12), k
