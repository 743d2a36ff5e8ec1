"""Step latency at the reference's own batch sizes (configs/scan3r/*.yaml: batch_size 4..8 pairs): where launches and host code,
not kernels, set the pace.  python tools/bench_small.py [pairs=4] [objects=40] [modules=point,gat,rel,attr]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd.synthetic import make_batch, to_device
from sgaligner_amd.trainer import AlignerSteps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mods = (sys.argv[3] if len(sys.argv) > 3 else 'point,gat,rel,attr').split(',')
steps = AlignerSteps(mods, device='cuda', seed=42)
dds = [to_device(make_batch(B, N, 512, seed=7 + i, ragged=True), 'cuda') for i in range(4)]
for i in range(8):
    steps.forward_backward(dds[i % 4])
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter()
for i in range(n):
    steps.forward_backward(dds[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f'{B} pairs x ~{N} objects x 512 pts, {"+".join(mods)}: {dt * 1e3:.2f} ms/step wall = {B / dt:.0f} pairs/s')
