"""GPU: the one-pass anchors x anchors mode of the fused loss (ops.FUSED_AA_ONEPASS).  When OverallLoss can announce dL/d(terms) at forward
time (the standard head: it depends on the log_vars and constants only), the A x A similarities are computed ONCE -- the backward kernel
runs inside forward(), returns the term values from the same launches, and backward() only scales the saved gradients by the upstream
factor.  Must equal the two-pass path (forward kernel + backward kernel) in every returned value and every gradient; an upstream factor
is honoured; a gradient that is not a multiple of the announced one is refused loudly (deferred)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=0, pairs=24, nobj=40, mods=('point', 'gat', 'rel')):
    from sgaligner_amd.aligner import losses as L
    from sgaligner_amd.aligner.sg_aligner import MultiModalFusion
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(pairs, nobj, 1, seed=seed, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    torch.manual_seed(seed)
    base = {k: torch.randn(T, 100, device='cuda') for k in mods}
    w0 = torch.randn(len(mods), 1, device='cuda') * 0.5
    lv1, lv2 = 0.3 * torch.randn(len(mods), device='cuda'), 0.3 * torch.randn(len(mods), device='cuda')

    def run(onepass, upstream=1.0, key='loss', retain=False):
        from sgaligner_amd import ops
        e = {k: base[k].clone().requires_grad_(True) for k in mods}
        fus = MultiModalFusion(len(mods)).cuda()
        ial, icl = L.CustomMultiLossLayer(len(mods)).cuda(), L.CustomMultiLossLayer(len(mods)).cuda()
        with torch.no_grad():
            fus.weight.copy_(w0); ial.log_vars.copy_(lv1); icl.log_vars.copy_(lv2)
        out = dict(e)
        out['joint'] = fus([e[k] for k in mods])
        fn = L.OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': list(mods)})
        keep = ops.FUSED_AA_ONEPASS
        ops.FUSED_AA_ONEPASS = onepass
        try:
            res = fn(out, dd)
            (res[key] * upstream).backward(retain_graph=retain)
            if retain:
                (res[key] * upstream).backward()
            torch.cuda.synchronize()
        finally:
            ops.FUSED_AA_ONEPASS = keep
        grads = [e[k].grad for k in mods] + [fus.weight.grad, ial.log_vars.grad, icl.log_vars.grad]
        return {k: float(v) for k, v in res.items()}, grads
    return dd, run


@pytest.mark.parametrize('mods', [('point', 'gat', 'rel'), ('point', 'gat', 'rel', 'attr'), ('point', 'rel')])
def test_onepass_equals_two_pass(mods):
    from sgaligner_amd import ops
    dd, run = _setup(seed=len(mods), mods=mods)
    assert len(dd['e1i']) >= ops.ONEPASS_MIN_ANCHORS
    ops.DEFERRED_CHECKS.flush()
    r2, g2 = run(False)
    r1, g1 = run(True)
    ops.DEFERRED_CHECKS.flush()
    for k in r2:
        assert abs(r1[k] - r2[k]) <= 1e-6 * max(1.0, abs(r2[k])), (k, r1[k], r2[k])
    for a, b in zip(g1, g2):
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-12, b.abs().max().item()), ((a - b).abs().max().item(), b.abs().max().item())


def test_onepass_upstream_factor_and_retain_graph():
    """(loss * 0.37).backward() -- gradient accumulation divides the loss like this -- and backward twice on one graph."""
    from sgaligner_amd import ops
    dd, run = _setup(seed=7)
    _, g_ref = run(False, upstream=0.37)
    _, g_one = run(True, upstream=0.37)
    ops.DEFERRED_CHECKS.flush()
    for a, b in zip(g_one, g_ref):
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-12, b.abs().max().item())
    _, g_twice = run(True, upstream=1.0, retain=True)
    _, g_once = run(False, upstream=1.0)
    ops.DEFERRED_CHECKS.flush()
    for a, b in zip(g_twice, g_once):
        assert (a - 2 * b).abs().max().item() <= 4e-5 * max(1e-12, b.abs().max().item())


def test_onepass_refuses_a_gradient_it_did_not_announce():
    """Backward through one of the returned COMPONENTS alone is not a multiple of dL/d(terms) of `loss`: the saved A x A gradients cannot
    serve it, and the mode says so (at the next check) instead of returning wrong numbers; with the mode off the same call is fine."""
    from sgaligner_amd import ops
    dd, run = _setup(seed=9)
    ops.DEFERRED_CHECKS.flush()
    _, g_bad = run(True, key='icl_loss_unimodal')
    with pytest.raises(RuntimeError, match='FUSED_AA_ONEPASS'):
        ops.DEFERRED_CHECKS.flush()
    # ... and what it returned meanwhile cannot be consumed by an optimiser step: the table gradients are NaN (device-side poison,
    # independent of VALIDATE and of when the deferred error is polled)
    for g in g_bad[:3]:
        assert torch.isnan(g).all()
    ops.DEFERRED_CHECKS.flush()
    keep = ops.VALIDATE
    ops.VALIDATE = False
    try:
        _, g_bad = run(True, key='ial_loss')
    finally:
        ops.VALIDATE = keep
    for g in g_bad[:3]:
        assert torch.isnan(g).all()
    ops.DEFERRED_CHECKS.flush()
    _, g_ok = run(False, key='icl_loss_unimodal')
    ops.DEFERRED_CHECKS.flush()
    for g in g_ok[:3]:
        assert torch.isfinite(g).all()


def test_onepass_not_used_without_gradients():
    from sgaligner_amd.synthetic import make_batch, to_device
    from sgaligner_amd.trainer import AlignerSteps
    steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=3)
    dd = to_device(make_batch(24, 40, 16, seed=2), 'cuda')
    _, l_train = steps.train_step(0, 0, dd)
    with torch.no_grad():
        _, l_eval = steps.train_step(0, 0, dd)
    assert l_train['loss'].requires_grad and not l_eval['loss'].requires_grad
    assert abs(float(l_train['loss']) - float(l_eval['loss'])) <= 1e-6 * abs(float(l_eval['loss']))


@pytest.mark.parametrize('mods,stash_rows', [(('point', 'gat', 'rel'), None), (('point', 'rel'), 64), (('point', 'gat', 'rel'), 32),
                                             (('point', 'gat', 'rel', 'attr'), None), (('point', 'gat', 'rel', 'attr'), 32)])
def test_symmetric_walk_equals_ordered_walk(mods, stash_rows):
    """ops.AA_SYMMETRIC: every unordered anchor pair evaluated once (a block also produces the mirrored elements right of it) --
    same terms and gradients as the walk that visits both orders; several blocks of growing height with a small stash."""
    from sgaligner_amd import ops
    dd, run = _setup(seed=11 + len(mods), pairs=30, mods=mods)
    A = len(dd['e1i'])
    keep_s, keep_b = ops.AA_SYMMETRIC, ops.STASH_BYTES
    try:
        if stash_rows:
            ops.STASH_BYTES = 4 * len(mods) * 2 * A * stash_rows
            assert len(ops._sym_chunks(A, len(mods))) >= 3
        ops.AA_SYMMETRIC = False
        r0, g0 = run(True)
        ops.AA_SYMMETRIC = True
        r1, g1 = run(True)
        ops.DEFERRED_CHECKS.flush()
    finally:
        ops.AA_SYMMETRIC, ops.STASH_BYTES = keep_s, keep_b
    for k in r0:
        assert abs(r1[k] - r0[k]) <= 1e-6 * max(1.0, abs(r0[k])), (k, r1[k], r0[k])
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-12, b.abs().max().item()), ((a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize('A,rows,M', [(2100, 512, 3), (1000, 96, 2), (333, 160, 3), (1500, 480, 4), (333, 96, 4)])
def test_symmetric_kernel_walk_at_the_c_abi(A, rows, M):
    """sga_loss_anchor_multi_bwd_sym + sga_loss_stash_grad_sym over a whole walk == sga_loss_anchor_multi_bwd + sga_loss_stash_grad:
    terms, dL/d(sums), dL/dbeta and dZ; ragged last block, A not a multiple of 16, stashes poisoned with NaN beforehand."""
    from sgaligner_amd import _lib
    from sgaligner_amd.ops import _p, _ptr_array, _stream
    L = _lib.lib(); st = _stream(); dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(A)
    zs = []
    for m in range(M):
        z = torch.zeros(2 * A + 32, 104, device=dev)
        z[:2 * A, :100] = torch.nn.functional.normalize(torch.randn(2 * A, 100, device=dev, generator=g), dim=1)
        zs.append(z)
    nt, n_terms, slots = M + 1, M + 1 + 2 * M, 1 + L.sga_loss_slots()
    sums = torch.rand(nt, 8, device=dev, dtype=torch.float64, generator=g) * 1e3 + 1e3
    beta = torch.softmax(torch.randn(M, device=dev, generator=g), 0)
    coef = (torch.rand(3 * M + 1, device=dev, generator=g) + 0.5) * 1e-2
    zarr = _ptr_array(zs)

    def walk(sym):
        dz = [torch.zeros(2 * A + 32, 104, device=dev) for _ in range(M)]
        acc = [torch.zeros(n_terms, device=dev, dtype=torch.float64), torch.zeros(nt, 8, device=dev, dtype=torch.float64),
               torch.zeros(M, device=dev, dtype=torch.float64)]
        gsc = torch.empty(slots + 1, nt, 8, device=dev, dtype=torch.float64)
        gam2 = torch.empty(slots, M, device=dev, dtype=torch.float64)
        out = torch.empty(slots * n_terms, device=dev, dtype=torch.float64)
        for lo in range(0, A, rows):
            hi = min(lo + rows, A); ns = hi - lo
            m1 = [torch.full(((A - lo if sym else A) * ns,), float('nan'), device=dev) for _ in range(M)]
            if sym:
                m2 = [torch.full((max(1, (A - hi) * ns),), float('nan'), device=dev) for _ in range(M)]
                _lib.check(L.sga_loss_anchor_multi_bwd_sym(zarr, M, _p(beta), A, _p(sums), 0.5, 0.1, 1.0, _p(coef), _ptr_array(m1), _ptr_array(m2),
                                                           _p(gsc), _p(gam2), lo, hi, _p(out), st), 'sym')
                for k in range(M):
                    _lib.check(L.sga_loss_stash_grad_sym(_p(m1[k]), _p(m2[k]), _p(zs[k]), A, 104, _p(dz[k]), lo, hi, st), 'stash sym')
            else:
                _lib.check(L.sga_loss_anchor_multi_bwd(zarr, M, _p(beta), A, _p(sums), 0.5, 0.1, 1.0, _p(coef), _ptr_array(m1), _p(gsc), _p(gam2),
                                                       lo, hi, _p(out), st), 'ordered')
                for k in range(M):
                    _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zs[k]), A, 104, _p(dz[k]), lo, hi, st), 'stash')
            acc[0] += out[:n_terms]; acc[1] += gsc[0]; acc[2] += gam2[0]
        torch.cuda.synchronize()
        return acc + dz

    a, b = walk(False), walk(True)
    for x, y in zip(b, a):
        assert torch.isfinite(x).all()
        assert (x - y).abs().max().item() <= 5e-6 * y.abs().max().item(), ((x - y).abs().max().item(), y.abs().max().item())


def test_symmetric_entry_refuses_blocks_off_the_32_row_grid():
    from sgaligner_amd import _lib
    from sgaligner_amd.ops import _p, _ptr_array, _stream
    L = _lib.lib()
    z = [torch.zeros(2 * 64 + 32, 104, device='cuda') for _ in range(2)]
    d = torch.zeros(4096, device='cuda', dtype=torch.float64)
    f = torch.zeros(64 * 64, device='cuda')
    rc = L.sga_loss_anchor_multi_bwd_sym(_ptr_array(z), 2, _p(f), 64, _p(d), 0.5, 0.1, 1.0, _p(f), _ptr_array([f, f]), _ptr_array([f, f]),
                                         _p(d), _p(d), 8, 40, _p(d), _stream())
    assert rc != 0 and b'32-row' in L.sga_last_error()


@pytest.mark.parametrize('R,M,stash', [(2, 3, None), (3, 3, 1 << 19), (4, 2, 1 << 19), (5, 3, 1 << 18), (8, 3, None), (3, 4, 1 << 19)])
def test_symmetric_walk_sharded_over_ranks_equals_unsharded(R, M, stash):
    """The symmetric anchors x anchors walk on EVERY rank of an anchor-sharded job (ops._sym_jobs: own square + the rectangles against the
    next ranks' rows, cyclically; csrc sga_loss_anchor_multi_bwd_symx / sga_loss_stash_grad_symx), simulated on one GPU with a deterministic
    replay of the three all-reduces: every rank ends with the global term values, and the ranks' shares of dL/dE and dL/d(fusion weight)
    sum to the unsharded one-pass result.  32 anchors per pair, so every pair cut is on a 32-row boundary; a small stash bound forces
    several row blocks and the wrapped column ranges."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    pairs = 24
    dd = make_batch(pairs, 107, 1, seed=5)
    assert len(dd['e1i']) == 32 * pairs
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(R * 10 + M)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(M)]
    w0 = (torch.randn(M, 1, device='cuda', generator=g) * 0.5)
    lv1, lv2 = 0.3 * torch.randn(M, device='cuda', generator=g), 0.3 * torch.randn(M, device='cuda', generator=g)
    hint = ops.LossHeadFn.coef_hint(lv1, lv2, 32 * pairs, 0.1, 0.5, 0.1)
    keep, calls = ops.STASH_BYTES, []
    from sgaligner_amd import loss_ops          # (where FusedContrastiveFn looks _sym_jobs up)
    orig = loss_ops._sym_jobs
    loss_ops._sym_jobs = lambda cuts, rank, nt: (calls.append((len(cuts) - 1, rank)), orig(cuts, rank, nt))[1]
    if stash is not None:
        ops.STASH_BYTES = stash
    try:
        def run(shard, reduce):
            tabs = [b.clone().requires_grad_(True) for b in base]
            w = w0.clone().requires_grad_(True)
            sums, s = ops.fused_contrastive_terms(tabs, w, dict(dd), shard=shard, reduce=reduce, coef_hint=hint)
            (sums * hint).sum().backward()
            torch.cuda.synchronize()
            return sums.detach(), [t.grad for t in tabs], w.grad
        ref_sums, ref_g, ref_w = run(None, None)
        per = [pairs // R + (1 if r < pairs % R else 0) for r in range(R)]
        cuts = [32 * sum(per[:r]) for r in range(R + 1)]
        totals, results = [], None
        for rnd in range(4):
            partial, results = [None] * R, []
            for rank in range(R):
                state = {'n': 0}

                def reduce(t, rank=rank, state=state):
                    n = state['n']
                    state['n'] += 1
                    if n < len(totals):
                        t.copy_(totals[n])
                    elif n == len(totals):
                        partial[rank] = t.clone()
                results.append(run((cuts[rank], cuts[rank + 1], cuts, rank), reduce))
            if rnd < 3:
                assert all(p is not None for p in partial), rnd
                totals.append(sum(partial))
        assert any(c == (R, r) for c in calls for r in range(R)), calls            # the sharded symmetric plan really ran
        for sums_r, _, _ in results:
            assert torch.allclose(sums_r, ref_sums, rtol=2e-5, atol=1e-6)
        for m in range(M):
            tot = sum(res[1][m] for res in results)
            sc = ref_g[m].abs().max().item()
            assert (tot - ref_g[m]).abs().max().item() < 2e-5 * sc, (m, (tot - ref_g[m]).abs().max().item(), sc)
        totw = sum(res[2] for res in results)
        assert (totw - ref_w).abs().max().item() < 2e-4 * max(1e-3, ref_w.abs().max().item())
    finally:
        ops.STASH_BYTES = keep
        loss_ops._sym_jobs = orig
