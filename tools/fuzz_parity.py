"""Randomised cross-checks on the GPU (no oracle: implementation against implementation, plus additivity properties):
  * fused multi-table loss (sweep16 / sweep16x2 / fused A x A kernels / loss head) vs the general per-table kernels with the joint
    table as an independent table: terms and every gradient, random (pairs, objects, M, anchor mode, raggedness);
  * anchor-sharded evaluation (random cuts) sums to the unsharded one;
  * PointNet forward split over a workgroup's waves vs one wave per object: outputs and arg-max bit-identical, random (T, P).
  python tools/fuzz_parity.py [seconds=300] [seed=0] [big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch, make_batch_fast

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
BIG = len(sys.argv) > 3 and sys.argv[3] == 'big'
t_end = time.time() + budget
n_loss = n_pn = n_shard = 0
worst = 0.0
while time.time() < t_end:
    # ---- loss
    M = int(rng.integers(2, 5))
    if BIG:       # enough rows for several K-splits per owner block and hundreds of owner blocks (uniform scenes)
        B, N = int(rng.integers(64, 160)), int(rng.choice([32, 48, 64]))
        dd = make_batch_fast(B, N, 4, seed=int(rng.integers(1 << 30)), device='cuda', anchors=('val', 'train')[int(rng.integers(2))])
        T = int(dd['tot_obj_pts'].shape[0])
    else:
        B, N = int(rng.integers(1, 40)), int(rng.integers(6, 70))
        dd = make_batch(B, N, 4, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), anchors=('val', 'train')[int(rng.integers(2))])
        T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(M)]
    w0 = torch.randn(M, 1, device='cuda', generator=g)
    cot = torch.rand(3 * M + 1, device='cuda', generator=g) + 0.5
    res = {}
    for fused in (True, False):
        tabs = [b.clone().requires_grad_(True) for b in base]
        w = w0.clone().requires_grad_(True)
        if fused:
            sums, s = ops.fused_contrastive_terms(tabs, w, dict(dd))
        else:
            ws = torch.softmax(w, dim=0)
            joint = torch.cat([ws[m] * torch.nn.functional.normalize(tabs[m], dim=1) for m in range(M)], dim=1)
            sums, s = ops.contrastive_terms(tabs + [joint], dict(dd))
        (sums * cot).sum().backward()
        res[fused] = (sums.detach().double(), [t.grad for t in tabs], w.grad)
    a, b = res[True], res[False]
    if s.A == 0:
        continue
    e = float(((a[0] - b[0]).abs() / b[0].abs().clamp_min(1e-6)).max())
    for m in range(M):
        e = max(e, float((a[1][m] - b[1][m]).abs().max() / max(1e-6, float(b[1][m].abs().max()))))
    e = max(e, float((a[2] - b[2]).abs().max() / max(1e-6, float(b[2].abs().max()))))
    worst = max(worst, e)
    assert e < 2e-3, ('loss', B, N, M, e)
    n_loss += 1
    # ---- anchor shards: terms and gradients are additive over a random partition of the anchors
    if s.A >= 3 and rng.integers(2):
        cuts = sorted(set([0, s.A] + [int(c) for c in rng.integers(1, s.A, size=2)]))
        tot_g = [torch.zeros_like(t) for t in base]
        tot_w = torch.zeros_like(w0)
        ref_tabs = [b_.clone().requires_grad_(True) for b_ in base]
        ref_w = w0.clone().requires_grad_(True)
        ref_sums, _ = ops.fused_contrastive_terms(ref_tabs, ref_w, dict(dd))
        # the forward sums of a shard are partial; replaying the all-reduce needs the totals: take them from the unsharded run via
        # the deterministic-replay helper of the test-suite is heavy -- here only the gradient additivity given FULL sums is checked
        # through the reduce hook: every shard is handed the totals of all shards (two rounds)
        totals = []
        for rnd in range(4):
            parts = []
            grads = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                st = {'n': 0}
                rec = []
                def reduce(t, st=st, rec=rec):
                    n = st['n']; st['n'] += 1
                    if n < len(totals): t.copy_(totals[n])
                    elif n == len(totals): rec.append(t.clone())
                tabs = [b_.clone().requires_grad_(True) for b_ in base]
                w = w0.clone().requires_grad_(True)
                sums, _ = ops.fused_contrastive_terms(tabs, w, dict(dd), shard=(lo, hi), reduce=reduce)
                (sums * cot).sum().backward()
                parts.append(rec[0] if rec else None)
                grads.append(([t.grad for t in tabs], w.grad, sums.detach()))
            if rnd < 3:
                totals.append(sum(p for p in parts))
        (ref_sums * cot).sum().backward()
        for m in range(M):
            gs = sum(gr[0][m] for gr in grads)
            e2 = float((gs - ref_tabs[m].grad).abs().max() / max(1e-6, float(ref_tabs[m].grad.abs().max())))
            assert e2 < 2e-3, ('shard', B, N, M, cuts, e2)
        n_shard += 1
    # ---- PointNet split vs unsplit
    Tn, P = int(rng.integers(1, 600)), int(rng.integers(33, 700))
    x = torch.randn(Tn, P, 3, device='cuda', generator=g)
    ws_ = [torch.randn(64, 3, device='cuda', generator=g), torch.randn(64, device='cuda', generator=g),
           torch.randn(128, 64, device='cuda', generator=g) * 0.2, torch.randn(128, device='cuda', generator=g),
           torch.randn(256, 128, device='cuda', generator=g) * 0.1, torch.randn(256, device='cuda', generator=g)]
    keep = ops.POINTNET_SPLIT_MAX_OBJECTS
    y1, a1 = ops.pointnet_forward(x, *ws_, True)
    ops.POINTNET_SPLIT_MAX_OBJECTS = 0
    y0, a0 = ops.pointnet_forward(x, *ws_, True)
    ops.POINTNET_SPLIT_MAX_OBJECTS = keep
    assert torch.equal(y0, y1) and torch.equal(a0, a1), ('pointnet', Tn, P)
    n_pn += 1
torch.cuda.synchronize()
print(f'fuzz ok: {n_loss} loss cases (worst rel. difference {worst:.2e}), {n_shard} shard partitions, {n_pn} PointNet split cases')
