"""Micro-benchmark of the PointNet forward kernel (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [torch.randn(64, 3, device='cuda') * 0.2, torch.zeros(64, device='cuda'),
     torch.randn(128, 64, device='cuda') * 0.1, torch.zeros(128, device='cuda'),
     torch.randn(256, 128, device='cuda') * 0.1, torch.zeros(256, device='cuda')]
for am in (False, True):
    for _ in range(2):
        ops.pointnet_forward(x, *w, want_argmax=am)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 5
    for _ in range(n):
        ops.pointnet_forward(x, *w, want_argmax=am)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    fl = 82304.0 * T * P
    print(f'pointnet_fwd argmax={am} T={T} P={P}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f}% of fp32 peak)')
