"""Randomised cross-check of the loss_group kernels on the GPU: MFMA forms of the group similarity / gradient kernels against the
VALU forms they replaced (terms and every gradient), random (pairs, objects, b, M, raggedness, anchor mode).
  python tools/fuzz_groups.py [seconds=120] [seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgaligner_amd import _lib, ops
from sgaligner_amd.synthetic import make_batch
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n = 0
worst = 0.0
L = _lib.lib()
while time.time() < t_end:
    B, N, M, b = int(rng.integers(1, 30)), int(rng.integers(5, 80)), int(rng.integers(1, 5)), int(rng.integers(1, 13))
    dd = make_batch(B, N, 4, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), anchors=('val', 'train')[int(rng.integers(2))])
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(M)]
    w0 = torch.randn(M, 1, device='cuda', generator=g)
    res = {}
    for valu in (0, 1):
        ops.GROUP_LOSS_VALU = bool(valu)
        tabs = [t.clone().requires_grad_(True) for t in base]
        w = w0.clone().requires_grad_(True)
        out, gr = ops.grouped_contrastive_terms(tabs, w if M > 1 else None, dict(dd), b)
        cot = torch.linspace(0.5, 1.5, out.numel(), device='cuda', dtype=out.dtype).view_as(out)
        (out * cot).sum().backward()
        res[valu] = (out.detach().double(), [t.grad for t in tabs], w.grad if M > 1 else None)
    ops.GROUP_LOSS_VALU = False
    a, c = res[0], res[1]
    e = float(((a[0] - c[0]).abs() / c[0].abs().clamp_min(1e-3)).max()) if a[0].numel() else 0.0
    for x, y in zip(a[1], c[1]):
        e = max(e, float((x - y).abs().max() / max(1e-6, float(y.abs().max()))))
    if M > 1:
        e = max(e, float((a[2] - c[2]).abs().max() / max(1e-6, float(c[2].abs().max()))))
    worst = max(worst, e)
    assert e < 2e-3, (B, N, M, b, e)
    n += 1
print(f'fuzz ok: {n} grouped-loss cases, MFMA vs VALU kernels, worst rel. difference {worst:.2e}')
