"""Symmetric anchors x anchors walk (sga_loss_anchor_multi_bwd_sym + sga_loss_stash_grad_sym) against the ordinary one
(sga_loss_anchor_multi_bwd + sga_loss_stash_grad) at the C ABI: terms, dL/d(sums), dL/dbeta, dZ of a whole walk; then timing of one block.
  python tools/dbg/aa_sym_check.py [anchors=2100] [rows_per_block=512] [M=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgaligner_amd import _lib
from sgaligner_amd.ops import _p, _ptr_array, _stream
A = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 512
M = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = _lib.lib(); st = _stream(); dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
zs = []
for m in range(M):
    z = torch.zeros(2 * A + 32, 104, device=dev)
    z[:2 * A, :100] = torch.nn.functional.normalize(torch.randn(2 * A, 100, device=dev, generator=g), dim=1)
    zs.append(z)
nt = M + 1; n_terms = nt + 2 * M
slots = 1 + L.sga_loss_slots()
sums = torch.rand(nt, 8, device=dev, dtype=torch.float64, generator=g) * 1e3 + 1e3
beta = torch.softmax(torch.randn(M, device=dev, generator=g), 0)
coef = (torch.rand(3 * M + 1, device=dev, generator=g) + 0.5) * 1e-2
zarr = _ptr_array(zs)


def walk(sym):
    dz = [torch.zeros(2 * A + 32, 104, device=dev) for _ in range(M)]
    terms = torch.zeros(n_terms, device=dev, dtype=torch.float64)
    gs = torch.zeros(nt, 8, device=dev, dtype=torch.float64); gam = torch.zeros(M, device=dev, dtype=torch.float64)
    gsc = torch.empty(slots + 1, nt, 8, device=dev, dtype=torch.float64)
    gam2 = torch.empty(slots, M, device=dev, dtype=torch.float64)
    out = torch.empty(slots * n_terms, device=dev, dtype=torch.float64)
    for lo in range(0, A, NS):
        hi = min(lo + NS, A); ns = hi - lo
        if sym:
            m1 = [torch.full(((A - lo) * ns,), float('nan'), device=dev) for _ in range(M)]
            m2 = [torch.full((max(1, (A - hi) * ns),), float('nan'), device=dev) for _ in range(M)]
            _lib.check(L.sga_loss_anchor_multi_bwd_sym(zarr, M, _p(beta), A, _p(sums), 0.5, 0.1, 1.0, _p(coef), _ptr_array(m1), _ptr_array(m2),
                                                       _p(gsc), _p(gam2), lo, hi, _p(out), st), 'sym')
            for k in range(M):
                _lib.check(L.sga_loss_stash_grad_sym(_p(m1[k]), _p(m2[k]), _p(zs[k]), A, 104, _p(dz[k]), lo, hi, st), 'sgs')
        else:
            m1 = [torch.full((A * ns,), float('nan'), device=dev) for _ in range(M)]
            _lib.check(L.sga_loss_anchor_multi_bwd(zarr, M, _p(beta), A, _p(sums), 0.5, 0.1, 1.0, _p(coef), _ptr_array(m1), _p(gsc), _p(gam2),
                                                   lo, hi, _p(out), st), 'aa')
            for k in range(M):
                _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zs[k]), A, 104, _p(dz[k]), lo, hi, st), 'sg')
        terms += out[:n_terms]; gs += gsc[0]; gam += gam2[0]
    torch.cuda.synchronize()
    return terms, gs, gam, dz


a, b = walk(False), walk(True)
rel = lambda x, y: ((x - y).abs().max() / y.abs().max().clamp_min(1e-300)).item()
print(f'A={A} rows/block={NS} M={M}: terms {rel(b[0], a[0]):.2e}  dL/dsums {rel(b[1], a[1]):.2e}  dL/dbeta {rel(b[2], a[2]):.2e}  '
      f'dZ {max(rel(x, y) for x, y in zip(b[3], a[3])):.2e}   (relative to the largest entry; finite: {all(torch.isfinite(x).all().item() for x in b[3])})')
