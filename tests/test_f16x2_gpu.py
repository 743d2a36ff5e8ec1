"""Opt-in, fp32-FAITHFUL split-fp16 MFMA mode of the anchors x negatives loss sweeps (ops.set_mfma_mode('f16x2'), csrc/sweeph.hip; the default
stays exact fp32; reference arithmetic src/aligner/losses.py:5-15,43-97).  The accuracy GATE of the round-3 review: the same tolerances as the
exact-fp32 path everywhere, and an error against the fp64 oracle of at most 2x the exact-fp32 path's own error (the configs[2]-sized part of
the gate -- every parameter's error against 4x the fp32 rerun noise -- is tests/test_c3_gpu.py::test_c3_f16x2_gate)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def f16x2():
    from sgaligner_amd import ops
    old = ops.set_mfma_mode('f16x2')
    yield
    ops.set_mfma_mode(old)


def test_mode_switch_roundtrip():
    from sgaligner_amd import ops
    d = ops.get_mfma_mode()
    assert d == ops.DEFAULT_MFMA_MODE == 'bf16x6'            # the default: fp32 arithmetic on three exact bf16 planes (tests/test_bf16x6_gpu.py)
    assert ops.set_mfma_mode('f16x2') == d and ops.get_mfma_mode() == 'f16x2'
    assert ops.set_mfma_mode('f32') == 'f16x2' and ops.get_mfma_mode() == 'f32'
    assert ops.set_mfma_mode(d) == 'f32'


@pytest.mark.parametrize('coef_lo', [True, False])
@pytest.mark.parametrize('M,emb', [(3, 100), (2, 100), (3, 64), (4, 100)])
def test_sweeps_vs_fp32_sweeps_and_anchor_shards(f16x2, M, emb, coef_lo):
    """The split-fp16 loss sweeps against the exact-fp32 sweeps on the same tables (loss terms, dE, d fusion weight), unsharded and as
    the sum of 3 anchor shards with cuts that are NOT multiples of the 32-row blocks (what ranks of a multi-GPU job own)."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    from test_c3_gpu import _replay_sharded
    dd = make_batch(9, 30, 4, seed=40 + M, ragged=True, anchors='val')
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(M)
    base = [torch.randn(T, emb, device='cuda', generator=g) for _ in range(M)]
    w0 = torch.tensor([[0.3], [1.1], [-0.4], [0.6]], device='cuda')[:M].contiguous()
    cot = torch.randn(M + 1 + 2 * M, device='cuda', generator=g)
    keep = ops.F16X2_COEF_LO
    ops.F16X2_COEF_LO = coef_lo
    try:
        def run():
            tabs = [b.clone().requires_grad_(True) for b in base]
            w = w0.clone().requires_grad_(True)
            sums, s = ops.fused_contrastive_terms(tabs, w, dd)
            (sums * cot).sum().backward()
            torch.cuda.synchronize()
            return sums.detach(), [t.grad for t in tabs], w.grad, s
        sb, gb, wb, s = run()
        ops.set_mfma_mode('f32')
        sf, gf, wf, _ = run()
        ops.set_mfma_mode('f16x2')
        assert torch.allclose(sb, sf, rtol=2e-6, atol=1e-7), (sb, sf)
        tol = 5e-6 if coef_lo else 3e-4            # rounded coefficients: 2^-12 per pair over only ~100 pairs per row here
        for m in range(M):
            sc = gf[m].abs().max().item()
            assert (gb[m] - gf[m]).abs().max().item() < tol * sc, (m, (gb[m] - gf[m]).abs().max().item(), sc)
        assert (wb - wf).abs().max().item() < 10 * tol * max(1e-3, wf.abs().max().item())
        A = s.A
        cuts = [0, A // 3 + 5, 2 * A // 3 - 3, A]
        _, gs, gw, all_sums = _replay_sharded(base, w0, cot, dd, cuts)
        for sr in all_sums:
            assert torch.allclose(sr, sb, rtol=1e-5, atol=1e-6)
        for m in range(M):
            sc = gb[m].abs().max().item()
            assert (gs[m] - gb[m]).abs().max().item() < (2e-5 if coef_lo else 2e-4) * sc, m
        assert (gw - wb).abs().max().item() < 2e-4 * max(1e-3, wb.abs().max().item())
    finally:
        ops.F16X2_COEF_LO = keep


def _overall_vs_fp64(pairs, nobj, seed):
    """The product OverallLoss on fused tables in both modes and the fp64 oracle: errors of every gradient against the oracle."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    from test_fullsize_gpu import _loss_setup, _run_overall
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(pairs, nobj, mods, seed=seed)
    w0 = torch.tensor([[0.7], [1.2], [0.9]], device='cuda')
    lv1 = torch.tensor([0.1, -0.2, 0.05], device='cuda')
    lv2 = torch.tensor([-0.1, 0.15, 0.0], device='cuda')
    eo = {k: base[i].cpu().double().requires_grad_(True) for i, k in enumerate(mods)}
    wo = w0.cpu().double().requires_grad_(True)
    lo1, lo2 = lv1.cpu().double().requires_grad_(True), lv2.cpu().double().requires_grad_(True)
    out_o = dict(eo)
    out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
    ref = O.overall_loss(out_o, dd, mods, lo1, lo2)
    ref['loss'].backward()
    errs = {}
    for mode in ('f32', 'f16x2'):
        old = ops.set_mfma_mode(mode)
        try:
            lf, gf, gwf, g1f, g2f = _run_overall(base, dd, mods, w0, lv1, lv2, fused=True)
        finally:
            ops.set_mfma_mode(old)
        e = {'loss': abs(lf - ref['loss'].item()) / abs(ref['loss'].item())}
        for k in mods:
            gref = eo[k].grad
            e['dE_' + k] = (gf[k].cpu().double() - gref).abs().max().item() / gref.abs().max().item()
            # the column sums are what reaches the layers below (d bias): a cancellation over all rows
            e['colsum_' + k] = (gf[k].cpu().double().sum(0) - gref.sum(0)).abs().max().item() / gref.sum(0).abs().max().item()
        e['dw'] = (gwf.cpu().double() - wo.grad).abs().max().item() / max(1e-30, wo.grad.abs().max().item())
        e['dlv1'] = (g1f.cpu().double() - lo1.grad).abs().max().item() / lo1.grad.abs().max().item()
        e['dlv2'] = (g2f.cpu().double() - lo2.grad).abs().max().item() / lo2.grad.abs().max().item()
        errs[mode] = e
    return errs


@pytest.mark.parametrize('pairs,nobj,seed', [(64, 64, 23), (16, 40, 5), (3, 30, 9)])
def test_error_vs_fp64_oracle_at_most_twice_the_fp32_paths(pairs, nobj, seed):
    """GATE: against the fp64 oracle (the largest batch-global losses it finishes in seconds and two small ones) the split-fp16 mode's error in
    the loss, every table gradient (entries and column sums), d fusion weight and both d log_vars is at most 2x the exact-fp32 path's own
    error (+ a floor of 1e-6 relative: a few fp32 roundings of the result -- the centred split adds rowsum x zbar to every gradient row), and
    within the fp32 tests' tolerances (1e-4 loss, 1e-3 gradients)."""
    errs = _overall_vs_fp64(pairs, nobj, seed)
    a, b = errs['f32'], errs['f16x2']
    for k in a:
        assert b[k] <= 2.0 * a[k] + 1e-6, (k, a[k], b[k])
        assert b[k] < (1e-4 if k == 'loss' else 1e-3), (k, b[k])


@pytest.mark.parametrize('mods', [['point', 'gat', 'rel'], ['point', 'gat', 'rel', 'attr']])
def test_train_step_vs_oracle(f16x2, mods):
    """One training step in the mode against the oracle, for the three-module list of the headline and the reference's full module list
    (M = 4: sweeph_kernel<4, ...>, the symmetric A x A kernel with 16 anchor rows per workgroup)."""
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch, to_device
    from sgaligner_amd.trainer import AlignerSteps
    dd = make_batch(3, 20, 96, seed=8, ragged=True)
    steps = AlignerSteps(mods, device='cuda', seed=3)
    params = {k: v.detach().cpu().clone() for k, v in steps.model.state_dict().items() if 'num_batches' not in k}
    out_o, loss_o, grads_o = O.train_step(params, dd, mods)
    out, loss = steps.forward_backward(to_device(dd, 'cuda'))
    torch.cuda.synchronize()
    for k in out_o:
        assert (out[k].detach().cpu() - out_o[k].detach()).abs().max() < 1e-3, k
    assert abs(loss['loss'].item() - loss_o['loss'].item()) < 1e-3 * max(1, abs(loss_o['loss'].item()))
    for name, p in steps.model.named_parameters():
        if name in grads_o and p.grad is not None:
            ref = grads_o[name]
            assert (p.grad.cpu() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item()), name


def test_loss_scale_independence(f16x2):
    """The coefficient's power-of-two scale follows dL/d(sums): cotangents 1e-6 ... 1e+6 times larger give gradients exactly that much
    larger (to fp32 rounding) -- no fp16 overflow or underflow of the coefficient whatever the loss scale."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(6, 24, 4, seed=3, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(1)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(3)]
    cot = torch.rand(3 + 1 + 6, device='cuda', generator=g) + 0.5
    ref = None
    for scale in (1.0, 1e-6, 1e6, 3e-12):
        tabs = [b.clone().requires_grad_(True) for b in base]
        w = torch.ones(3, 1, device='cuda', requires_grad=True)
        sums, _ = ops.fused_contrastive_terms(tabs, w, dd)
        (sums * cot * scale).sum().backward()
        gs = [t.grad / scale for t in tabs]
        assert all(torch.isfinite(x).all() for x in gs)
        if ref is None:
            ref = gs
        else:
            for x, y in zip(gs, ref):
                assert (x - y).abs().max().item() < 1e-5 * y.abs().max().item(), scale


def test_nearly_identical_rows_vs_fp64_oracle(f16x2):
    """A table whose rows are nearly identical (what meta_embedding_rel makes of bag-of-words rows that are almost all alike): the loss gradient
    is the small tangential remainder of a large radial sum.  The planes are CENTRED for this (csrc/sweeph.hip): against the fp64 oracle the
    mode's error in the table gradient and in its column sums (= the bias gradient below it) stays within 2x the exact-fp32 path's own
    (un-centred, the 22-bit image of the rows missed the column sums by 8x the fp32 path's error at this size)."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    from test_fullsize_gpu import _run_overall
    mods = ['point', 'gat', 'rel']
    dd = make_batch(64, 64, 1, seed=3)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator().manual_seed(0)
    base = [torch.randn(T, 100, generator=g, dtype=torch.float64) for _ in mods]
    base[2] = torch.randn(1, 100, generator=g, dtype=torch.float64) + 1e-3 * torch.randn(T, 100, generator=g, dtype=torch.float64)
    base = [b.float().double() for b in base]
    w0 = torch.tensor([[0.7], [1.2], [0.9]], dtype=torch.float64)
    lv1 = torch.tensor([0.1, -0.2, 0.05], dtype=torch.float64)
    lv2 = torch.tensor([-0.1, 0.15, 0.0], dtype=torch.float64)
    eo = {k: base[i].clone().requires_grad_(True) for i, k in enumerate(mods)}
    wo = w0.clone().requires_grad_(True)
    out_o = dict(eo)
    out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
    O.overall_loss(out_o, dd, mods, lv1.clone().requires_grad_(True), lv2.clone().requires_grad_(True))['loss'].backward()
    err = {}
    for mode in ('f32', 'f16x2'):
        ops.set_mfma_mode(mode)
        _, gg, _, _, _ = _run_overall([b.float().cuda() for b in base], dd, mods, w0.float().cuda(), lv1.float().cuda(), lv2.float().cuda(), fused=True)
        gref = eo['rel'].grad
        got = gg['rel'].cpu().double()
        err[mode] = ((got - gref).abs().max().item() / gref.abs().max().item(),
                     (got.sum(0) - gref.sum(0)).abs().max().item() / gref.sum(0).abs().max().item())
    ops.set_mfma_mode('f16x2')
    assert err['f16x2'][0] <= 2.0 * err['f32'][0] + 1e-6, err
    assert err['f16x2'][1] <= 2.0 * err['f32'][1] + 1e-6, err


def test_pointnet_forward_same_argmax_points_as_fp32(f16x2):
    """'f16x2' training forward of the object encoder (more objects than the few-object form takes): fp16 hi + lo split, and every object in
    which some channel's two largest layer-3 values are distinct and within 2^-16 of each other -- or an operand left the fp16 range --
    re-run on the exact-fp32 kernel (csrc/pointnet.hip, TIE).  Result: the SAME arg-max point as the exact-fp32 mode for every
    (object, channel) -- the backward routes gradients through them (pointnet.py:140-161) --, values within fp32 rounding, the re-run objects
    bit-identical; an object whose coordinates overflow the split (millimetres instead of metres) is re-run, not NaN."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    p = O.init_params(['point'])
    g = torch.Generator(device='cuda').manual_seed(3)
    ws = [p['object_encoder.conv1.weight'].reshape(64, 3).contiguous().cuda(), 0.1 * torch.randn(64, device='cuda', generator=g),
          p['object_encoder.conv2.weight'].reshape(128, 64).contiguous().cuda(), 0.1 * torch.randn(128, device='cuda', generator=g),
          p['object_encoder.conv3.weight'].reshape(256, 128).contiguous().cuda(), 0.1 * torch.randn(256, device='cuda', generator=g)]
    T, P = 6000, 77                                     # ragged last tile: replicated points tie exactly and must NOT count as near ties
    x = torch.randn(T, P, 3, device='cuda', generator=g)
    x[17] *= 3.0e4
    y, am = ops.pointnet_forward(x, *ws, want_argmax=True)
    redo = ops.POINTNET_LAST_REDO.clone()
    ops.set_mfma_mode('f32')
    y0, am0 = ops.pointnet_forward(x, *ws, want_argmax=True)
    ops.set_mfma_mode('f16x2')
    n = int(redo[0])
    ids = redo[1:1 + n].long()
    assert 0 < n < T // 3, n
    assert 17 in ids.tolist() and torch.isfinite(y).all()
    assert torch.equal(y[ids], y0[ids]) and torch.equal(am[ids], am0[ids])
    assert (y - y0).abs().max().item() < 2e-6 * y0[:17].abs().max().item()
    assert torch.equal(am, am0), int((am != am0).sum())
    # inference (no arg-max wanted): the split alone, values within fp32 rounding
    yi, _ = ops.pointnet_forward(x[18:], *ws, want_argmax=False)
    assert (yi - y0[18:]).abs().max().item() < 2e-6 * y0[18:].abs().max().item()


@pytest.fixture
def f16x2p():
    from sgaligner_amd import ops
    old = ops.set_mfma_mode('f16x2p')
    yield
    ops.set_mfma_mode(old)


@pytest.mark.parametrize('tag', ['pointnet_small', 'pointnet_ragged'])
def test_pointnet_forward_vs_reference_golden(f16x2p, tag):
    """'f16x2p' = 'f16x2' + the object encoder's forward in the same split (fp16 hi + lo of scaled weights / activations, csrc/pointnet.hip;
    NOT part of the faithful mode: a point max that ties to fp32 rounding may pick the other point, which re-routes that channel's gradient --
    one such (object, channel) pair among 18 147 moved the conv weight gradients of a 116-object step by 1e-3).  Values: against the reference
    module's golden output at the exact-fp32 kernel's own tolerance, and within fp32 rounding of the exact-fp32 kernel (1e-6 relative; the
    bf16 split of the older mode: 2e-4), with the same arg-max points wherever two points do not tie to 1e-6."""
    from conftest import load_golden
    from sgaligner_amd import ops
    g = load_golden(tag)
    x = torch.from_numpy(np.ascontiguousarray(g['x'].transpose(0, 2, 1))).cuda()   # golden x is [T,3,P]; the kernel takes [T,P,3]
    w = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(g[k].shape[0], -1) if g[k].ndim > 1 else g[k])).cuda()
         for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')]
    y, am = ops.pointnet_forward(x, *w, want_argmax=True)
    assert np.abs(y.cpu().numpy() - g['y']).max() < 2e-5
    ops.set_mfma_mode('f32')
    y0, am0 = ops.pointnet_forward(x, *w, want_argmax=True)
    ops.set_mfma_mode('f16x2p')
    assert (y - y0).abs().max().item() < 1e-6 * max(1.0, y0.abs().max().item())
    assert (am == am0).float().mean().item() > 0.999


def test_pointnet_forward_large_values_are_loud_not_wrong(f16x2p):
    """Activations beyond the fp16 range of the split (|h| >= 8190: coordinates in millimetres instead of metres) give NaN for that object,
    never a finite wrong number; the exact-fp32 mode handles the same input."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    p = O.init_params(['point'])
    ws = [p['object_encoder.conv1.weight'].reshape(64, 3).contiguous().cuda(), torch.zeros(64).cuda(),
          p['object_encoder.conv2.weight'].reshape(128, 64).contiguous().cuda(), torch.zeros(128).cuda(),
          p['object_encoder.conv3.weight'].reshape(256, 128).contiguous().cuda(), torch.zeros(256).cuda()]
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(8, 64, 3, device='cuda', generator=g)
    x[3] *= 3.0e4
    y, _ = ops.pointnet_forward(x, *ws, want_argmax=True)
    ops.set_mfma_mode('f32')
    y0, _ = ops.pointnet_forward(x, *ws, want_argmax=True)
    ops.set_mfma_mode('f16x2p')
    ok = torch.isfinite(y).all(dim=1)
    assert ok[[0, 1, 2, 4, 5, 6, 7]].all() and torch.isfinite(y0).all()
    assert (y[ok] - y0[ok]).abs().max().item() < 1e-5 * y0[ok].abs().max().item()
    assert not ok[3] or (y[3] - y0[3]).abs().max().item() < 1e-5 * y0[3].abs().max().item()


def test_forward_sums_hi_only_products(f16x2):
    """The forward sums with the lo terms of the similarity dropped on the 96 main columns (what configs[2]-sized sums use: >= 2^24 terms each):
    at 64 pairs x 64 objects (A x J = 1.2 M x 2.8 M terms per sum ... small, so the averaging is weakest here) every loss term stays within
    3e-6 of the exact-fp32 path's, and a table of nearly identical rows -- where un-centred 11-bit rows would shift every similarity by the
    same 1e-5 -- within 3e-6 as well."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(64, 64, 1, seed=3)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(5)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(3)]
    base[2] = torch.randn(1, 100, device='cuda', generator=g) + 1e-2 * torch.randn(T, 100, device='cuda', generator=g)
    w = torch.tensor([[0.3], [1.1], [-0.4]], device='cuda')
    keep = ops.F16X2_SUMS_LO
    try:
        res = {}
        for tag, mode, lo in (('f32', 'f32', None), ('full', 'f16x2', True), ('hi', 'f16x2', False)):
            ops.set_mfma_mode(mode)
            ops.F16X2_SUMS_LO = lo
            with torch.no_grad():
                res[tag] = ops.fused_contrastive_terms(base, w, dd)[0].double()
        ops.set_mfma_mode('f16x2')
        for tag in ('full', 'hi'):
            rel = ((res[tag] - res['f32']).abs() / res['f32'].abs()).max().item()
            assert rel < (1e-6 if tag == 'full' else 3e-6), (tag, rel)
    finally:
        ops.F16X2_SUMS_LO = keep


@pytest.mark.parametrize('mode', ['f16x2', 'bf16x6'])
def test_tables_wider_than_100_columns_take_the_fp32_kernels(mode):
    """emb_dim 101..104 is accepted by the fused loss path, but columns 100, 101 of the split modes' planes carry the row centring's bookkeeping:
    such tables must run on the fp32 kernels EVERYWHERE -- sweeps, the symmetric A x A similarities, the stash products (round-4 advisor: the
    A x A planes of 'f16x2' used to drop columns 100..103 silently).  Terms and gradients equal the 'f32' mode's to fp32 summation noise, and
    columns 100..103 carry gradient."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(40, 40, 4, seed=11, ragged=True, anchors='val')       # enough anchors for the one-pass symmetric walk
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(4)
    base = [torch.randn(T, 104, device='cuda', generator=g) for _ in range(3)]
    for b in base:
        b[:, 100:] *= 3.0                                                  # make the last four columns matter
    w0 = torch.tensor([[0.3], [1.1], [-0.4]], device='cuda')
    res = {}
    for md in ('f32', mode):
        old = ops.set_mfma_mode(md)
        try:
            tabs = [b.clone().requires_grad_(True) for b in base]
            w = w0.clone().requires_grad_(True)
            hint = torch.linspace(0.5, 1.5, 3 + 1 + 6, device='cuda')
            sums, s = ops.fused_contrastive_terms(tabs, w, dd, coef_hint=hint)
            (sums * hint).sum().backward()
            torch.cuda.synchronize()
            res[md] = (sums.detach().double(), [t.grad.clone() for t in tabs], w.grad.clone())
        finally:
            ops.set_mfma_mode(old)
    a, b = res['f32'], res[mode]
    assert torch.allclose(a[0], b[0], rtol=1e-6, atol=1e-9)
    for x, y in zip(a[1], b[1]):
        assert (x - y).abs().max().item() < 2e-5 * x.abs().max().item()
        assert x[:, 100:].abs().max().item() > 1e-3 * x.abs().max().item()
    assert (a[2] - b[2]).abs().max().item() < 1e-4 * a[2].abs().max().item()
