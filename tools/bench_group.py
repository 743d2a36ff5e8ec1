"""Step time with reference-sized loss groups (loss_group = b) at the BASELINE configs[1] batch: python tools/bench_group.py [b=4] [pairs=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dd = make_batch_fast(B, 64, 512, seed=43, device='cuda')
for lg in ('global', b):
    steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42, loss_group=lg)
    for _ in range(3):
        steps.forward_backward(dd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        steps.forward_backward(dd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f'loss_group={lg}: {dt * 1e3:.2f} ms/step = {B / dt:.0f} pairs/s ({B} pairs x 64 objects x 512 pts, point+gat+rel)')
