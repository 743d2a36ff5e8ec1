import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops, _lib
print('is_available', torch.cuda.is_available())
for T, P in ((42, 64), (42, 40), (3, 64), (1100, 64)):
    x = torch.randn(T, P, 3, device='cuda')
    w = [torch.randn(64, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1, torch.randn(128, 64, device='cuda') * 0.1, torch.randn(128, device='cuda') * 0.1,
         torch.randn(256, 128, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1]
    sums = torch.empty(265 + 512, device='cuda', dtype=torch.float64)
    try:
        y, am = ops.pointnet_forward(x, *w, want_argmax=True, bn_sums=sums)
        torch.cuda.synchronize()
        print(T, P, 'ok', float(y.abs().max()))
    except Exception as e:
        print(T, P, 'FAILED', e)
