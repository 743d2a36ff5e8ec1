import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from sgaligner_amd.aligner.networks.pct import NaivePCT
Tt, N = 1024, 512
mt = NaivePCT().cuda().train()
xt = torch.randn(Tt, 3, N, device='cuda')
cot = torch.randn(Tt, 256, device='cuda')
for _ in range(2):
    mt.zero_grad(set_to_none=True)
    (mt(xt) * cot).sum().backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    mt.zero_grad(set_to_none=True)
    (mt(xt) * cot).sum().backward()
torch.cuda.synchronize()
print('train ms', (time.perf_counter() - t0) / 5 * 1e3, 'peak GiB', torch.cuda.max_memory_allocated() / 2**30)
