import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd.aligner import losses as L
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps
dd = make_batch_fast(512, 64, 512, seed=43, device='cuda')
for fused in (True, False):
    L.FUSED_JOINT = fused
    steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
    for _ in range(2):
        steps.forward_backward(dd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        steps.forward_backward(dd)
    torch.cuda.synchronize()
    print('FUSED_JOINT', fused, (time.perf_counter() - t0) / 5 * 1e3, 'ms/step')
L.FUSED_JOINT = True
