"""FPS throughput: one scan-like batch (objects of 300..20000 points, 512 samples each) on the GPU vs the numpy oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import fps_oracle
from sgaligner_amd.utils import point_cloud as pc

rng = np.random.default_rng(0)
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 2048            # ~40 scans x ~50 objects
sizes = np.clip((rng.lognormal(7.6, 0.9, n_obj)).astype(np.int64), 512, 60000)
npoint = 512
pts = torch.from_numpy((rng.standard_normal((int(sizes.sum()), 3))).astype(np.float32)).cuda()
off = np.concatenate([[0], np.cumsum(sizes)])
start = [0] * n_obj
out = pc.farthest_point_sample_batch(pts, off, npoint, start)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = pc.farthest_point_sample_batch(pts, off, npoint, start)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 3
k = 8
t0 = time.perf_counter()
p_cpu = pts.cpu().numpy()
for j in range(k):
    ref = fps_oracle.farthest_point_sample_idx(p_cpu[off[j]:off[j + 1]], npoint, 0)
    assert np.array_equal(ref, out[j].cpu().numpy())
cpu = (time.perf_counter() - t0) / k
# algorithmic traffic: every round reads each point of the object once (12 B) + 4 B of running distance r/w
work = float((sizes * npoint).sum())
print(f'FPS: {n_obj} objects, {int(sizes.sum())} points ({int(sizes.min())}..{int(sizes.max())} per object), {npoint} samples each: '
      f'GPU {gpu*1e3:.2f} ms = {n_obj/gpu:.0f} objects/s, {work/gpu/1e9:.1f} G point-visits/s; '
      f'numpy oracle {cpu*1e3:.1f} ms/object (first {k} objects, 1 core) -> {1/cpu:.1f} objects/s')
