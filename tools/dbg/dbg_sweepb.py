import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgaligner_amd import ops, _lib
from sgaligner_amd.ops import _p, _ptr_array, _stream
from sgaligner_amd.synthetic import make_batch_fast
L = _lib.lib()
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 20)
dd = make_batch_fast(B, N, 4, seed=3, device='cuda')
T = int(dd['tot_obj_pts'].shape[0])
s = ops.IndexSets.of(dd, 'cuda', T)
g = torch.Generator(device='cuda').manual_seed(0)
M = int(os.environ.get("DBG_M", "3"))
zs = []
for m in range(M):
    e = torch.randn(T, 100, device='cuda', generator=g)
    if os.environ.get('DBG_SAME') and m > 0:
        e = e_first
    e_first = e if m == 0 else e_first
    z = torch.zeros((s.R + 32, 104), device='cuda'); nrm = torch.empty(s.R, device='cuda')
    L.sga_loss_gather(_p(e), T, 100, _p(s.idx), s.R, _p(z), 104, _p(nrm), _stream())
    zs.append(z)
beta = torch.tensor([0.2, 0.5, 0.3], device="cuda")[:M].contiguous()
slots = 1 + L.sga_loss_slots()
nt = M + 1
sums_a = torch.empty((slots, nt, 8), device='cuda', dtype=torch.float64)
sums_b = torch.empty_like(sums_a)
_lib.check(L.sga_loss_multi_sums(_ptr_array(zs), M, 100, _p(beta), s.A, s.J1, s.J2, 0.1, 1.0, _p(sums_a), 0, s.A, _stream()), 'a')
nb = L.sga_loss_split_bytes(s.A, s.J1, s.J2)
zbs = []
for z in zs:
    zb = torch.empty(nb, device='cuda', dtype=torch.uint8)
    _lib.check(L.sga_loss_split_tables(_p(z), s.A, s.J1, s.J2, _p(zb), _stream()), 'split')
    zbs.append(zb)
_lib.check(L.sga_loss_multi_sums_bf16x3(_ptr_array(zbs), M, _p(beta), s.A, s.J1, s.J2, 0.1, 1.0, _p(sums_b), 0, s.A, _stream()), 'b')
torch.cuda.synchronize()
print('A', s.A, 'J', s.J1, s.J2)
print('sums rel err per table/fam', ((sums_a[0] - sums_b[0]).abs() / sums_a[0].abs()).cpu().numpy())
gs = torch.randn(nt, 8, device="cuda", dtype=torch.float64, generator=torch.Generator(device="cuda").manual_seed(1)).abs() * 1e-3
def grad(fn, tabs):
    dzs = [torch.zeros((s.R, 104), device='cuda') for _ in range(M)]
    gam = torch.empty((slots, M), device='cuda', dtype=torch.float64)
    if fn == 'a':
        _lib.check(L.sga_loss_multi_grad(_ptr_array(tabs), M, 100, _p(beta), s.A, s.J1, s.J2, 0.1, 1.0, _p(gs), _ptr_array(dzs), _p(gam), 0, s.A, _stream()), 'ga')
    else:
        _lib.check(L.sga_loss_multi_grad_bf16x3(_ptr_array(tabs), M, _p(beta), s.A, s.J1, s.J2, 0.1, 1.0, _p(gs), _ptr_array(dzs), _p(gam), 0, s.A, _stream()), 'gb')
    torch.cuda.synchronize()
    return dzs, gam[0].clone()
da, ga = grad('a', zs)
db, gb = grad('b', zbs)
print('gamma', ga.cpu().numpy(), gb.cpu().numpy())
A, J1, J2 = s.A, s.J1, s.J2
for m in range(M):
    d = (da[m] - db[m]).abs()
    sc = da[m].abs().max().item()
    segs = {'x1': (0, A), 'x2': (A, 2 * A), 'n1': (2 * A, 2 * A + J1), 'n2': (2 * A + J1, 2 * A + J1 + J2)}
    print('table', m, 'scale', sc, {k: float(d[lo:hi].max()) for k, (lo, hi) in segs.items()})
    bad = (d.max(1).values > 1e-3 * sc).nonzero().flatten()
    print('   bad rows', bad[:20].tolist(), 'count', bad.numel(), ' bad cols of first bad row', (d[bad[0]] > 1e-3 * sc).nonzero().flatten().tolist() if bad.numel() else None)
