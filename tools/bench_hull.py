"""Convex-hull barycentre of a batch of objects: reference arithmetic (Qhull on every point, preprocess.py:93-96) vs the GPU candidate
filter + Qhull on the survivors.  python tools/bench_hull.py [n_obj=400] [points=20000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import hull_oracle
from sgaligner_amd.utils import point_cloud
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 400
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(0)
objs = [(rng.standard_normal((npts, 3)) * rng.uniform(0.2, 2.0, size=3)).astype(np.float32) for _ in range(n_obj)]
point_cloud.convex_hull_barycenters_batch(objs[:4])
t0 = time.time(); a, info = point_cloud.convex_hull_barycenters_batch(objs, return_info=True); t1 = time.time()
b = np.stack([hull_oracle.hull_barycenter(o)[0] for o in objs]); t2 = time.time()
import torch
off = np.concatenate([[0], np.cumsum([len(o) for o in objs])])
keep, _ = point_cloud.hull_candidate_mask_batch(torch.from_numpy(np.concatenate(objs)).cuda(), off)
print(f'{n_obj} objects x {npts} points: filter + device hull ({info}) {1e3*(t1-t0):.0f} ms, Qhull on all points {1e3*(t2-t1):.0f} ms ({(t2-t1)/(t1-t0):.1f}x); '
      f'kept {100*float(keep.float().mean()):.1f} % of the points; max |barycentre diff| {np.abs(a-b).max():.2e}')
