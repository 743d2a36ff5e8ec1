"""PointNet backward (sga_pointnet_bwd) in both arithmetics -- fp32 MFMA ('f32') and three exact bf16 planes ('bf16x6', the default) -- on the same
forward (same arg-max points): HIP events on the launch stream, and the default's parameter gradients against the fp32 kernel's.
  python tools/bench_pointnet_bwd.py [T=65536]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P = 512
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [(torch.randn(64, 3, device='cuda') * 0.2).requires_grad_(True), torch.zeros(64, device='cuda', requires_grad=True),
     (torch.randn(128, 64, device='cuda') * 0.1).requires_grad_(True), torch.zeros(128, device='cuda', requires_grad=True),
     (torch.randn(256, 128, device='cuda') * 0.1).requires_grad_(True), torch.zeros(256, device='cuda', requires_grad=True)]
cot = torch.randn(T, 256, device='cuda')
y = ops.pointnet(x, *w)                     # forward once (default arithmetic): both backward kernels see the same arg-max points
res = {}
for mode in ('f32', 'bf16x6'):
    old = ops.set_mfma_mode(mode)
    try:
        for _ in range(2):
            for p in w:
                p.grad = None
            y.backward(cot, retain_graph=True)
        torch.cuda.synchronize()
        res[mode] = [p.grad.clone() for p in w]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            y.backward(cot, retain_graph=True)
        e.record(); torch.cuda.synchronize()
        print(f'pointnet bwd mode={mode} T={T}: {s.elapsed_time(e)/5:.3f} ms')
    finally:
        ops.set_mfma_mode(old)
names = ['conv1.weight', 'conv1.bias', 'conv2.weight', 'conv2.bias', 'conv3.weight', 'conv3.bias']
for n, a, b in zip(names, res['f32'], res['bf16x6']):
    print(f'   {n:13s} max |three planes - fp32 MFMA| / max |fp32 MFMA| = {float((a - b).abs().max() / a.abs().max()):.2e}')
