"""Decode sga_loss_aa_planes and check that the decoded rows reproduce x_i . x_j (tail columns swapped on one side)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import _lib
from sgaligner_amd.ops import _p, _stream
L = _lib.lib(); A = 300
g = torch.Generator(device='cuda').manual_seed(1)
z = torch.zeros(2 * A + 32, 104, device='cuda')
z[:2 * A, :100] = torch.nn.functional.normalize(torch.randn(2 * A, 100, device='cuda', generator=g) + 3.0, dim=1)
h = torch.empty(2 * A + 1, 104, device='cuda')
_lib.check(L.sga_loss_aa_planes(_p(z), 2 * A, _p(h), _stream()), 'planes'); torch.cuda.synchronize()
raw = h[:2 * A].contiguous().view(torch.float16).view(2 * A, 208).double()
cols = torch.zeros(2 * A, 104, dtype=torch.float64, device='cuda')
for q in range(3):
    cols[:, 32 * q:32 * q + 32] = raw[:, 64 * q:64 * q + 32] + raw[:, 64 * q + 32:64 * q + 64]
cols[:, 96:104] = raw[:, 192:200] + raw[:, 200:208]
cols /= 4096.0
print('mean row', h[2 * A, :4].tolist(), h[2 * A, 100].item(), 'true mean', z[:2 * A, :4].mean(0).tolist(), 0.5 * float((z[:2 * A].double().mean(0) ** 2).sum()))
print('col 100/101 of row 0', cols[0, 100].item(), cols[0, 101].item())
sw = cols.clone(); sw[:, 100] = cols[:, 101]; sw[:, 101] = cols[:, 100]
S = cols[:A] @ sw[A:].t()
Sref = z[:A].double() @ z[A:2 * A].double().t()
print('max |S - Sref|', float((S - Sref).abs().max()))
