"""Randomised cross-check of the fp16 tile core (csrc/wide16.hip) and the epilogue-only A x A kernels on WIDE tables: mode 'f16' against the
exact-fp32 wide-table path on the same ragged batch -- loss terms and every table gradient within the mode's 1e-2 -- over random (pairs,
objects, width, number of tables, workspace size = number of anchor-row blocks); and the exact-fp32 path with the similarity blocks formed
beforehand against the same path forced through several row blocks (additivity).
  python tools/fuzz_wide16.py [seconds=120] [seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n = 0
worst_t = worst_g = worst_add = 0.0
keep = ops.STASH_BYTES
while time.time() < t_end:
    B, N = int(rng.integers(1, 14)), int(rng.integers(6, 64))
    D = int(rng.integers(129, 900))
    nt = int(rng.integers(1, 4))
    dd = make_batch(B, N, 4, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), anchors=('val', 'train')[int(rng.integers(2))])
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    base = [torch.randn(T, D, device='cuda', generator=g) + 0.2 * k for k in range(nt)]
    cot = torch.rand(nt + 2 * (nt - 1 if nt > 1 else 0), device='cuda', generator=g) + 0.5
    s0 = ops.IndexSets.of(dd, 'cuda', T)
    if s0.A == 0:
        continue
    small = int(2 * 2 * max(s0.J1, s0.J2, 1) * (int(rng.integers(8, 200)) + 8) + 4096) if rng.integers(2) else None
    res = {}
    for tag, mode, stash in (('f32', 'f32', None), ('f32b', 'f32', small), ('f16', 'f16', small)):
        if tag == 'f32b' and small is None:
            res[tag] = res['f32']
            continue
        tabs = [b.clone().requires_grad_(True) for b in base]
        old = ops.set_mfma_mode(mode)
        try:
            if stash is not None:
                ops.STASH_BYTES = stash
            sums, _ = ops.contrastive_terms(tabs, dict(dd))
            (sums * cot).sum().backward()
            torch.cuda.synchronize()
        finally:
            ops.set_mfma_mode(old)
            ops.STASH_BYTES = keep
        res[tag] = (sums.detach().double(), [t.grad.double() for t in tabs])
    for tag, tol_t, tol_g in (('f16', 1e-2, 1e-2), ('f32b', 1e-4, 1e-4)):
        rel = ((res[tag][0] - res['f32'][0]).abs() / res['f32'][0].abs().clamp_min(1e-12)).max().item()
        gerr = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(res[tag][1], res['f32'][1]))
        assert np.isfinite(rel) and np.isfinite(gerr), (tag, B, N, D, nt, small)
        assert rel < tol_t and gerr < tol_g, (tag, rel, gerr, B, N, D, nt, small, s0.A, s0.J1, s0.J2)
        if tag == 'f16':
            worst_t, worst_g = max(worst_t, rel), max(worst_g, gerr)
        else:
            worst_add = max(worst_add, rel, gerr)
    n += 1
print(f'fuzz_wide16: {n} random wide-table cases; f16 vs fp32: worst loss-term error {worst_t:.2e}, worst gradient error {worst_g:.2e} of its maximum '
      f'(tolerance 1e-2); fp32 in row blocks vs one block: {worst_add:.2e}')
