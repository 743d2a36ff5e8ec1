import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops
T, P = 4096, 64
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [torch.randn(64, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1,
     torch.randn(128, 64, device='cuda') * 0.1, torch.randn(128, device='cuda') * 0.1,
     torch.randn(256, 128, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1]
sums = torch.empty(265 + 512, device='cuda', dtype=torch.float64)
y0, a0 = ops.pointnet_forward(x, *w, want_argmax=True)
y1, a1 = ops.pointnet_forward(x, *w, want_argmax=True, bn_sums=sums)
torch.cuda.synchronize()
d = (y0 - y1).abs()
print('max diff', d.max().item())
bad = d > 1e-5
print('bad by (object parity, channel half):')
for par in (0, 1):
    for hf in (0, 1):
        print(par, hf, int(bad[par::2, hf * 128:(hf + 1) * 128].sum()), 'of', bad[par::2, hf * 128:(hf + 1) * 128].numel())
print('bad per 32-channel block:', [int(bad[:, c * 32:(c + 1) * 32].sum()) for c in range(8)])
print('first bad object rows:', bad.any(1).nonzero()[:10].flatten().tolist())
