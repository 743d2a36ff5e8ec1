"""Inference path throughput: encoder forward (no grad) + per-pair similarity ranking / Hits@K (simrank kernel) on
BASELINE.json configs[1]-sized validation batches (every common object is an anchor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd.synthetic import make_batch
from sgaligner_amd.trainer import AlignerSteps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
steps.model.eval()
dd = make_batch(B, 64, 512, seed=5, device='cuda', gen_device='cuda', anchors='val')
def once():
    with torch.no_grad():
        out = steps.test_step(0, dd)
        return steps.eval_step(0, dd, out)
m = once(); once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m = once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
with torch.no_grad():
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5):
        out = steps.test_step(0, dd)
    torch.cuda.synchronize(); enc = (time.perf_counter() - t1) / 5
print(f'inference: {B} pairs x 64 objects x 512 pts: {dt*1e3:.1f} ms per batch = {B/dt:.0f} pairs/s '
      f'(encoder forward {enc*1e3:.1f} ms, ranking + metrics {1e3*(dt-enc):.1f} ms); Hits@1 {m[1]["correct"]}/{m[1]["total"]} on random weights')
