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
y = ops.pointnet(x, *w)
for _ in range(2):
    y.backward(cot, retain_graph=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    y.backward(cot, retain_graph=True)
e.record(); torch.cuda.synchronize()
print(f'pointnet bwd T={T}: {s.elapsed_time(e)/5:.3f} ms')
