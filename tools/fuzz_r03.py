"""Randomised cross-checks of the round-3 kernels on the GPU (implementation against implementation / against scipy):
  * wide16 (fp16-input loss on wide tables) vs the exact-fp32 wide path: terms within 1e-2, gradients within 1e-2 of their maximum,
    random (pairs, objects, width, raggedness, workspace size -> anchor-row blocks);
  * device hull vertices vs scipy/Qhull on random point sets (blobs, spheres, boxes, duplicates, float32 origins): wherever the
    device certifies, the vertex coordinates are identical; it must decline, never differ;
  * the PCT head's algebraic backward vs the chain of separate nodes, random (objects, points, gamma signs, train / eval);
  * the one-pass anchors x anchors mode vs the two-pass path: terms and every gradient, random (batch, M, stash blocks, upstream factor).
  python tools/fuzz_r03.py [seconds=300] [seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial import ConvexHull
from sgaligner_amd import ops, pct_ops as P
from sgaligner_amd.synthetic import make_batch
from sgaligner_amd.utils import point_cloud

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_w = n_h = n_hd = n_p = n_o = 0
worst_o_at = None
worst_w = worst_p = worst_o = 0.0
keep_stash = ops.STASH_BYTES
if os.environ.get('SGA_FUZZ_NOSYM'):
    ops.AA_SYMMETRIC = False
while time.time() < t_end:
    # ---- wide16 vs fp32 wide path
    B, N = int(rng.integers(1, 14)), int(rng.integers(6, 60))
    D = int(rng.choice([136, 200, 264, 520, 1024, 1032]))
    dd = make_batch(B, N, 1, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), anchors=('val', 'train')[int(rng.integers(2))])
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    base = [torch.randn(T, D, device='cuda', generator=g) + 0.3 * float(rng.random()) for _ in range(2)]
    cot = torch.rand(4, device='cuda', generator=g) + 0.5
    res = {}
    for mode in ('f32', 'f16'):
        tabs = [b.clone().requires_grad_(True) for b in base]
        old = ops.set_mfma_mode(mode)
        try:
            if mode == 'f16' and rng.random() < 0.5:
                s0 = ops.IndexSets.of(dd, 'cuda', T)
                ops.STASH_BYTES = 2 * 2 * max(s0.J1, s0.J2, 8) * (int(rng.choice([128, 256])) + 8) + 4096
            sums, _ = ops.contrastive_terms(tabs, dict(dd))
            (sums * cot).sum().backward()
            torch.cuda.synchronize()
        finally:
            ops.set_mfma_mode(old)
            ops.STASH_BYTES = keep_stash
        res[mode] = (sums.detach().double(), [t.grad.double() for t in tabs])
    rel = ((res['f16'][0] - res['f32'][0]).abs() / res['f32'][0].abs().clamp_min(1e-12)).max().item()
    assert rel < 1e-2, ('wide16 terms', B, N, D, rel)
    for a, b in zip(res['f16'][1], res['f32'][1]):
        e = (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
        worst_w = max(worst_w, e)
        assert e < 1e-2, ('wide16 grad', B, N, D, e)
    n_w += 1

    # ---- device hull vs Qhull
    sets = []
    for _ in range(24):
        n = int(rng.integers(4, 513))
        kind = int(rng.integers(5))
        if kind == 0:
            p = rng.standard_normal((n, 3)) * rng.uniform(0.05, 3.0, size=3) + rng.uniform(-50, 50, size=3)
        elif kind == 1:
            p = rng.standard_normal((n, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True); p *= rng.uniform(0.1, 5.0)
        elif kind == 2:
            p = (rng.random((n, 3)) - 0.5) * rng.uniform(0.01, 4.0, size=3)
        elif kind == 3:
            p = rng.standard_normal((n, 3)); p = np.concatenate([p, p[rng.integers(0, n, size=max(1, n // 5))]])[:512]
        else:
            p = (rng.standard_normal((n, 3)) * rng.uniform(0.01, 1.0) + rng.uniform(-2000, 2000, size=3)).astype(np.float32).astype(np.float64)
        sets.append(p)
    off = np.concatenate([[0], np.cumsum([len(x) for x in sets])])
    isv, status = point_cloud.hull_vertices_batch(np.concatenate(sets), off)
    for k, p in enumerate(sets):
        if status[k] != 0:
            n_hd += 1
            continue
        h = ConvexHull(p)
        ref = np.unique(h.points[h.vertices], axis=0)
        mine = np.unique(p[isv[off[k]:off[k + 1]]], axis=0)
        assert mine.shape == ref.shape and np.array_equal(mine, ref), ('hull', k, len(p), mine.shape, ref.shape)
        n_h += 1

    # ---- PCT head: algebraic backward vs separate nodes
    Tn, Np = int(rng.integers(2, 40)), int(rng.choice([32, 64, 96, 200]))
    K, C = 512, 1024
    training = bool(rng.integers(2))
    cat0 = torch.randn(Tn * Np, K, device='cuda', generator=g) * float(rng.uniform(0.2, 1.5))
    w0 = torch.randn(C, K, 1, device='cuda', generator=g) * 0.05
    cotp = torch.randn(Tn, C, device='cuda', generator=g)
    gam = (torch.randn(C, device='cuda', generator=g).abs() + 0.1) * torch.where(torch.rand(C, device='cuda', generator=g) < 0.2, -1.0, 1.0)
    bet = torch.randn(C, device='cuda', generator=g) * 0.3
    rm, rv = torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5
    out = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(C).cuda()
        with torch.no_grad():
            bn.weight.copy_(gam); bn.bias.copy_(bet); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        bn.train(training)
        cat = cat0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        gg = P.linear_bn_lrelu_max(cat, w, bn, Tn, Np) if fused else P.segment_max(P.batch_norm_act(P.rows_linear(cat, w, bn_stats=True), bn, act=2), Tn, Np)
        (gg * cotp).sum().backward()
        torch.cuda.synchronize()
        out.append((gg.detach(), cat.grad, w.grad, bn.weight.grad, bn.bias.grad))
    assert torch.equal(out[0][0], out[1][0])
    for k in range(1, 5):
        e = (out[0][k] - out[1][k]).abs().max().item() / max(1e-6, out[1][k].abs().max().item())
        worst_p = max(worst_p, e)
        assert e < 1e-3, ('pct head', Tn, Np, training, k, e)
    n_p += 1

    # ---- one-pass A x A (terms + gradients from one pass, ops.FusedContrastiveFn with coef_hint) vs the two-pass path:
    # random batch, M, stash size (-> several anchor-row blocks), upstream factor
    Mm = int(rng.integers(2, 5))
    Bp, Nn = int(rng.integers(14, 60)), int(rng.integers(30, 70))
    dd2 = make_batch(Bp, Nn, 1, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), anchors=('val', 'train')[int(rng.integers(2))])
    if len(dd2['e1i']) >= ops.ONEPASS_MIN_ANCHORS:
        T2 = int(dd2['tot_obj_count'].sum())
        tb = [torch.randn(T2, 100, device='cuda', generator=g) for _ in range(Mm)]
        wv = torch.randn(Mm, 1, device='cuda', generator=g) * 0.5
        hint = (torch.rand(3 * Mm + 1, device='cuda', generator=g) + 0.2) * 1e-4
        up = float(rng.uniform(0.2, 3.0))
        rr = []
        for h in (None, hint):
            tabs2 = [t.clone().requires_grad_(True) for t in tb]
            w2 = wv.clone().requires_grad_(True)
            try:
                if h is not None and rng.random() < 0.6:
                    ops.STASH_BYTES = 4 * len(dd2['e1i']) * Mm * int(rng.choice([32, 64, 160]))
                sm, _ = ops.fused_contrastive_terms(tabs2, w2, dd2, coef_hint=h)
                (sm.double() * hint.double() * up).sum().backward()
                torch.cuda.synchronize()
            finally:
                ops.STASH_BYTES = keep_stash
            rr.append((sm.detach().double(), [t.grad for t in tabs2] + [w2.grad]))
        ops.DEFERRED_CHECKS.flush()
        assert ((rr[0][0] - rr[1][0]).abs() / rr[0][0].abs().clamp_min(1e-12)).max().item() < 1e-6, 'one-pass terms'
        for ti, (a_, b_) in enumerate(zip(rr[1][1], rr[0][1])):
            e = (a_ - b_).abs().max().item() / max(1e-30, b_.abs().max().item())
            if e > worst_o:
                worst_o, worst_o_at = e, (Bp, Nn, Mm, 'fusion weights' if ti == Mm else f'table {ti}')
            assert e < 2e-4, ('one-pass grad', Bp, Nn, Mm, e)
        n_o += 1
print(f'fuzz_r03: {n_w} wide16 cases (worst gradient difference {worst_w:.2e} of the maximum), {n_h} hulls certified == Qhull '
      f'({n_hd} declined), {n_p} PCT-head cases (worst difference {worst_p:.2e}), {n_o} one-pass A x A cases (worst gradient difference '
      f'{worst_o:.2e} at {worst_o_at}); no mismatch')
