"""The reference-default 'pct' training step alone (scan3r_ground_truth.yaml: pct+gat+rel+attr, 4 pairs x ~40 objects x 512 pts): wall time per step.
  python tools/bench_pct_step.py [steps=40]      (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd.trainer import AlignerSteps
from sgaligner_amd.synthetic import make_batch, to_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda:0')
steps = AlignerSteps(['pct', 'gat', 'rel', 'attr'], device=dev, seed=42)
dd = [to_device(make_batch(4, 40, 512, seed=7 + i, ragged=True), dev) for i in range(4)]
for i in range(6):
    steps.forward_backward(dd[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    steps.forward_backward(dd[i % 4])
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / n
print(f'pct step: {el * 1e3:.3f} ms = {4 / el:.1f} pairs/s ({sum(int(d["tot_obj_pts"].shape[0]) for d in dd) / 4:.0f} objects per step)')
