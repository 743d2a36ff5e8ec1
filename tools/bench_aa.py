"""Time the anchors x anchors kernel (sga_loss_anchor_multi_bwd, TERMS build) and its stash GEMMs alone on random unit rows.
  python tools/bench_aa.py [anchors=19456] [rows_per_block=2048] [M=3] [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as ct
import numpy as np, torch
from sgaligner_amd import _lib
from sgaligner_amd.ops import _p, _ptr_array, _stream
A = int(sys.argv[1]) if len(sys.argv) > 1 else 19456
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
M = int(sys.argv[3]) if len(sys.argv) > 3 else 3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = _lib.lib(); st = _stream(); dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
zs = []
for m in range(M):
    z = torch.zeros(2 * A + 32, 104, device=dev)
    z[:2 * A, :100] = torch.nn.functional.normalize(torch.randn(2 * A, 100, device=dev, generator=g), dim=1)
    zs.append(z)
nt = M + 1
slots = 1 + L.sga_loss_slots()
sums = torch.rand(nt, 8, device=dev, dtype=torch.float64, generator=g) * 1e5 + 1e5
beta = torch.full((M,), 1.0 / M, device=dev)
coef = (torch.rand(3 * M + 1, device=dev, generator=g) + 0.5) * 1e-4
m1 = [torch.empty(A * NS, device=dev) for _ in range(M)]
gsc = torch.empty(slots + 1, nt, 8, device=dev, dtype=torch.float64)
gam2 = torch.empty(slots, M, device=dev, dtype=torch.float64)
out = torch.empty(slots * (nt + 2 * M), device=dev, dtype=torch.float64)
dz = [torch.zeros(2 * A + 32, 104, device=dev) for _ in range(M)]
zarr = _ptr_array(zs)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tk, tg = [], []
for r in range(reps + 1):
    e[0].record()
    _lib.check(L.sga_loss_anchor_multi_bwd(zarr, M, _p(beta), A, _p(sums), 0.5, 0.1, 1.0, _p(coef), _ptr_array(m1), _p(gsc), _p(gam2), 0, NS, _p(out), st), 'aa')
    e[1].record()
    for k in range(M):
        _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zs[k]), A, 104, _p(dz[k]), 0, NS, st), 'sg')
    e[2].record()
    torch.cuda.synchronize()
    if r:
        tk.append(e[0].elapsed_time(e[1])); tg.append(e[1].elapsed_time(e[2]))
el = float(NS) * A
print(f'A x A kernel (M={M}, {NS} rows x {A} anchors): {np.median(tk):.3f} ms = {np.median(tk) * 1e6 / el:.4f} ns per (i, j) pair; stash GEMMs {np.median(tg):.3f} ms '
      f'({2.0 * 2 * M * el * 104 / np.median(tg) / 1e9:.1f} TFLOP/s); checksum {float(out[:nt + 2 * M].sum()):.6e} {float(m1[0].abs().sum()):.6e}')
# symmetric block: rows [0, NS) x columns [0, A), both elements of every pair -> compare HALF its time with the ordinary block above
if M <= 4 and NS % 32 == 0:
    m1s = [torch.empty(A * NS, device=dev) for _ in range(M)]
    m2s = [torch.empty((A - NS) * NS, device=dev) for _ in range(M)]
    ts, tgs = [], []
    for r in range(reps + 1):
        e[0].record()
        _lib.check(L.sga_loss_anchor_multi_bwd_sym(zarr, M, _p(beta), A, _p(sums), 0.5, 0.1, 1.0, _p(coef), _ptr_array(m1s), _ptr_array(m2s), _p(gsc), _p(gam2),
                                                   0, NS, _p(out), st), 'sym')
        e[1].record()
        for k in range(M):
            _lib.check(L.sga_loss_stash_grad_sym(_p(m1s[k]), _p(m2s[k]), _p(zs[k]), A, 104, _p(dz[k]), 0, NS, st), 'sgs')
        e[2].record()
        torch.cuda.synchronize()
        if r:
            ts.append(e[0].elapsed_time(e[1])); tgs.append(e[1].elapsed_time(e[2]))
    pairs = float(NS) * A * 2 - float(NS) * NS
    print(f'symmetric block: kernel {np.median(ts):.3f} ms = {np.median(ts) * 1e6 / pairs:.4f} ns per ordered pair; stash GEMMs {np.median(tgs):.3f} ms')
