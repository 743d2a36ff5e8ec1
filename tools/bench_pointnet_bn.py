"""PointNet training forward with and without the fused BatchNorm statistics (sga_pointnet_fwd_bn), HIP events on the launch stream.
  python tools/bench_pointnet_bn.py [T=131072] [P=512]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [torch.randn(64, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1,
     torch.randn(128, 64, device='cuda') * 0.1, torch.randn(128, device='cuda') * 0.1,
     torch.randn(256, 128, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1]
sums = torch.empty(265 + 512, device='cuda', dtype=torch.float64)
for tag, bn in (('plain', None), ('with BN sums', sums), ('plain', None), ('with BN sums', sums)):
    for _ in range(2):
        ops.pointnet_forward(x, *w, want_argmax=True, bn_sums=bn)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 5
    for _ in range(n):
        ops.pointnet_forward(x, *w, want_argmax=True, bn_sums=bn)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(f'T={T} P={P} {tag:14s} {ms:8.3f} ms   {82304.0 * T * P / ms / 1e9:.1f} TFLOP/s (forward FLOPs only)')
