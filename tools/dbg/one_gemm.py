import os, sys
sys.path.insert(0, '/root/repo')
import torch
from sgaligner_amd.aligner.networks.pct import _gemm_ex
m, K, N = 2097152, 128, 128
a = torch.randn(m, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda'); out = torch.empty(m, N, device='cuda')
for _ in range(3):
    _gemm_ex(a, w, b, act=1, out=out)
torch.cuda.synchronize()
