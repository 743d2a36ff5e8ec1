"""The anchors x negatives loss sweeps with every fp32 operand split EXACTLY into three bf16 planes (ops.set_mfma_mode('bf16x6'),
csrc/sweep3.hip; reference arithmetic src/aligner/losses.py:5-15,43-97; SURVEY 7: "parity configs use fp32 MFMA or split-bf16 x3").
What makes the mode fp32 arithmetic and not a narrower one is tested first: h + m + l of every stored plane element IS the fp32 operand,
bit for bit.  Then the same tolerances as the exact-fp32 MFMA path everywhere, and an error against the fp64 oracle no larger than that
path's own."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16x6():
    from sgaligner_amd import ops
    old = ops.set_mfma_mode('bf16x6')
    yield
    ops.set_mfma_mode(old)


def _bf16_to_f64(u16):
    return (u16.to(torch.int32) << 16).view(torch.float32).double()


@pytest.mark.parametrize('parallel_rows', [False, True])
def test_planes_represent_every_fp32_operand_exactly(parallel_rows):
    """sga_loss_split3_tables: for every row and column of a packed table (values over 60 binades, incl. exact zeros) the three stored bf16
    terms sum to the fp32 operand EXACTLY, |m| <= 2^-8 |h|, |l| <= 2^-16 |h|.  The operand is the table entry z itself (zbar = 0) unless
    the table's rows are nearly parallel (|mean row|^2 >= 1/4), then it is the fp32 difference z - zbar and the bookkeeping columns hold
    b = zbar . z' + |zbar|^2 / 2 and 1; the K-tail image holds columns 96..103 in the documented k-group order (h, h, m, l); the column
    means are bitwise reproducible (fixed summation order)."""
    from sgaligner_amd import _lib, ops
    L = _lib.lib()
    A, J1, J2 = 37, 70, 45
    R = 2 * A + J1 + J2
    g = torch.Generator(device='cuda').manual_seed(0)
    z = torch.randn(R + 32, 104, device='cuda', generator=g) * torch.exp2(torch.randint(-40, 0, (R + 32, 104), device='cuda', generator=g).float())
    if parallel_rows:
        z = 1e-3 * torch.randn(R + 32, 104, device='cuda', generator=g) + 0.1 * torch.randn(1, 104, device='cuda', generator=g)
    z[3, 7] = 0.0
    z[:, 100:] = 0.0
    z[R:] = 0.0
    nb = L.sga_loss_split3_bytes(A, J1, J2)
    outs = []
    for _ in range(2):
        zb = torch.zeros((nb,), device='cuda', dtype=torch.uint8)
        zc = torch.full((2 * A, 104), float('nan'), device='cuda')
        _lib.check(L.sga_loss_split3_tables(z.data_ptr(), A, J1, J2, zb.data_ptr(), zc.data_ptr(), ops._stream()), 'sga_loss_split3_tables')
        torch.cuda.synchronize()
        outs.append(zb.clone())
    assert torch.equal(outs[0], outs[1])
    zb = outs[0]
    nblk = [(A + 31) // 32, (A + 31) // 32, (J1 + 31) // 32, (J2 + 31) // 32]
    seg0 = [0, A, 2 * A, 2 * A + J1]
    seglen = [A, A, J1, J2]
    BLOCK, PLANE, TAIL = 20480, 6144, 18432
    stat = zb[(sum(nblk) + 1) * BLOCK:(sum(nblk) + 1) * BLOCK + 4 * 105].view(torch.float32)
    zbar = stat[:104]
    ref_mean = (z[:R, :100].double().sum(0) / R).float()
    centred = float((ref_mean.double() ** 2).sum()) >= 0.25
    assert centred == parallel_rows
    if not centred:
        ref_mean = torch.zeros_like(ref_mean)
    assert torch.equal(zbar[:100], ref_mean) and float(zbar[100:].abs().max()) == 0.0
    cen = torch.zeros((R, 104), device='cuda')
    cen[:, :100] = z[:R, :100] - zbar[:100]
    assert centred or torch.equal(cen[:, :100], z[:R, :100])
    cen[:, 100] = ((zbar[:100].double() * cen[:, :100].double()).sum(1) + stat[104].double()).float()
    cen[:, 101] = 1.0
    # the fp32 copy of the anchor rows for the stash products: the same centred values, zero in column 100, one in column 101
    want_c = cen[:2 * A].clone()
    want_c[:, 100] = 0.0
    assert torch.equal(zc, want_c)
    u16 = zb.view(torch.int16).to(torch.int32) & 0xFFFF

    def slot(gk, i):
        return 16 * gk + (i ^ (12 * (gk & 1)))
    blk = 0
    worst_m, worst_l = 0.0, 0.0
    for sgm in range(4):
        for b in range(nblk[sgm]):
            base = blk * BLOCK // 2
            for w in range(32):
                row = 32 * b + w
                jh, i = (w >> 2) & 1, 4 * (w >> 3) + (w & 3)
                want = cen[seg0[sgm] + row] if row < seglen[sgm] else torch.zeros(104, device='cuda')
                planes = []
                for p in range(3):
                    cols = torch.zeros(104, device='cuda', dtype=torch.float64)
                    for q in range(3):
                        for gk in range(4):
                            o = base + (p * PLANE + ((q * 2 + jh) * 64 + slot(gk, i)) * 16) // 2
                            cols[32 * q + 8 * gk:32 * q + 8 * gk + 8] = _bf16_to_f64(u16[o:o + 8])
                    planes.append(cols)
                # the tail image: k groups (h, h, m, l)
                t0 = [_bf16_to_f64(u16[base + (TAIL + (jh * 64 + slot(gk, i)) * 16) // 2:][:8]) for gk in range(4)]
                planes[0][96:104], planes[1][96:104], planes[2][96:104] = t0[0], t0[2], t0[3]
                assert torch.equal(t0[1], t0[0])
                tot = planes[0] + planes[1] + planes[2]
                assert torch.equal(tot, want.double()), (sgm, b, w, (tot - want.double()).abs().max().item())
                nz = planes[0] != 0
                if nz.any():
                    worst_m = max(worst_m, float((planes[1][nz] / planes[0][nz]).abs().max()))
                    worst_l = max(worst_l, float((planes[2][nz] / planes[0][nz]).abs().max()))
            blk += 1
    assert worst_m <= 2.0 ** -8 * 1.01 and worst_l <= 2.0 ** -16 * 1.01, (worst_m, worst_l)


@pytest.mark.parametrize('M,emb', [(3, 100), (2, 100), (3, 64), (4, 100), (4, 37)])
def test_sweeps_vs_fp32_sweeps_and_anchor_shards(bf16x6, M, emb):
    """The three-plane sweeps against the exact-fp32 MFMA sweeps on the same tables (loss terms, dE, d fusion weight), unsharded and as the
    sum of 3 anchor shards with cuts that are NOT multiples of the 32-row blocks (what ranks of a multi-GPU job own)."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    from test_c3_gpu import _replay_sharded
    dd = make_batch(9, 30, 4, seed=40 + M, ragged=True, anchors='val')
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(M)
    base = [torch.randn(T, emb, device='cuda', generator=g) for _ in range(M)]
    w0 = torch.tensor([[0.3], [1.1], [-0.4], [0.6]], device='cuda')[:M].contiguous()
    cot = torch.randn(M + 1 + 2 * M, device='cuda', generator=g)

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
    ops.set_mfma_mode('bf16x6')
    assert torch.allclose(sb, sf, rtol=2e-6, atol=1e-7), (sb, sf)
    for m in range(M):
        sc = gf[m].abs().max().item()
        assert (gb[m] - gf[m]).abs().max().item() < 5e-6 * sc, (m, (gb[m] - gf[m]).abs().max().item(), sc)
    assert (wb - wf).abs().max().item() < 5e-5 * max(1e-3, wf.abs().max().item())
    A = s.A
    cuts = [0, A // 3 + 5, 2 * A // 3 - 3, A]
    _, gs, gw, all_sums = _replay_sharded(base, w0, cot, dd, cuts)
    for sr in all_sums:
        assert torch.allclose(sr, sb, rtol=1e-5, atol=1e-6)
    for m in range(M):
        sc = gb[m].abs().max().item()
        assert (gs[m] - gb[m]).abs().max().item() < 2e-5 * sc, m
    assert (gw - wb).abs().max().item() < 2e-4 * max(1e-3, wb.abs().max().item())


def _overall_vs_fp64(pairs, nobj, seed, modes=('f32', 'bf16x6'), mods=('point', 'gat', 'rel')):
    """The product OverallLoss on fused tables in both modes and the fp64 oracle: errors of every gradient against the oracle."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    from test_fullsize_gpu import _loss_setup, _run_overall
    mods = list(mods)
    M = len(mods)
    dd, T, base = _loss_setup(pairs, nobj, mods, seed=seed)
    w0 = torch.tensor([[0.7], [1.2], [0.9], [1.05]], device='cuda')[:M].contiguous()
    lv1 = torch.tensor([0.1, -0.2, 0.05, 0.12], device='cuda')[:M].contiguous()
    lv2 = torch.tensor([-0.1, 0.15, 0.0, -0.07], device='cuda')[:M].contiguous()
    eo = {k: base[i].cpu().double().requires_grad_(True) for i, k in enumerate(mods)}
    wo = w0.cpu().double().requires_grad_(True)
    lo1, lo2 = lv1.cpu().double().requires_grad_(True), lv2.cpu().double().requires_grad_(True)
    out_o = dict(eo)
    out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
    ref = O.overall_loss(out_o, dd, mods, lo1, lo2)
    ref['loss'].backward()
    errs = {}
    for mode in modes:
        old = ops.set_mfma_mode(mode)
        try:
            lf, gf, gwf, g1f, g2f = _run_overall(base, dd, mods, w0, lv1, lv2, fused=True)
        finally:
            ops.set_mfma_mode(old)
        e = {'loss': abs(lf - ref['loss'].item()) / abs(ref['loss'].item())}
        for k in mods:
            gref = eo[k].grad
            e['dE_' + k] = (gf[k].cpu().double() - gref).abs().max().item() / gref.abs().max().item()
            e['colsum_' + k] = (gf[k].cpu().double().sum(0) - gref.sum(0)).abs().max().item() / gref.sum(0).abs().max().item()
        e['dw'] = (gwf.cpu().double() - wo.grad).abs().max().item() / max(1e-30, wo.grad.abs().max().item())
        e['dlv1'] = (g1f.cpu().double() - lo1.grad).abs().max().item() / lo1.grad.abs().max().item()
        e['dlv2'] = (g2f.cpu().double() - lo2.grad).abs().max().item() / lo2.grad.abs().max().item()
        errs[mode] = e
    return errs


@pytest.mark.parametrize('pairs,nobj,seed', [(64, 64, 23), (16, 40, 5), (3, 30, 9)])
def test_error_vs_fp64_oracle_no_larger_than_the_fp32_mfma_paths(pairs, nobj, seed):
    """Against the fp64 oracle (the largest batch-global losses it finishes in seconds and two small ones): the mode's error in the loss,
    every table gradient (entries and column sums), d fusion weight and both d log_vars is that of fp32 arithmetic -- at most 1.25 x the
    exact-fp32 MFMA path's own error + 5e-7 relative (two different fp32 summation orders differ by that much among themselves), and
    within the fp32 tests' tolerances (1e-4 loss, 1e-3 gradients)."""
    errs = _overall_vs_fp64(pairs, nobj, seed)
    a, b = errs['f32'], errs['bf16x6']
    for k in a:
        assert b[k] <= 1.25 * a[k] + 5e-7, (k, a[k], b[k])
        assert b[k] < (1e-4 if k == 'loss' else 1e-3), (k, b[k])


@pytest.mark.parametrize('pairs,nobj,seed', [(64, 64, 31), (16, 40, 6)])
def test_four_tables_one_launch_error_vs_fp64_oracle(pairs, nobj, seed):
    """M = 4 (point + gat + rel + attr, the module list of every yaml the reference ships): the gradient sweep forms all four tables' owner
    gradients in ONE launch with a shared set of small-product accumulators that is folded into the large ones by fp32 adds every tile
    (sweep3_kernel's FOLD form).  Against the fp64 oracle its errors -- entries AND column sums of every table gradient, where a one-sided
    fold would show -- stay those of fp32 arithmetic: at most 1.25 x the fp32-MFMA paired-wave kernels' own error + 5e-7."""
    errs = _overall_vs_fp64(pairs, nobj, seed, mods=('point', 'gat', 'rel', 'attr'))
    a, b = errs['f32'], errs['bf16x6']
    for k in a:
        assert b[k] <= 1.25 * a[k] + 5e-7, (k, a[k], b[k])
        assert b[k] < (1e-4 if k == 'loss' else 1e-3), (k, b[k])


def test_train_step_vs_oracle(bf16x6):
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch, to_device
    from sgaligner_amd.trainer import AlignerSteps
    mods = ['point', 'gat', 'rel']
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


def test_loss_scale_independence(bf16x6):
    """bf16 has fp32's exponent range: cotangents 1e-30 ... 1e+20 times larger give gradients exactly that much larger (to fp32 rounding) --
    no scaling logic, no overflow or underflow of a coefficient plane whatever the loss scale."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(6, 24, 4, seed=3, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(1)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(3)]
    cot = torch.rand(3 + 1 + 6, device='cuda', generator=g) + 0.5
    ref = None
    for scale in (1.0, 1e-6, 1e6, 1e-24, 1e20):
        tabs = [b.clone().requires_grad_(True) for b in base]
        w = torch.ones(3, 1, device='cuda', requires_grad=True)
        sums, _ = ops.fused_contrastive_terms(tabs, w, dd)
        (sums * cot * scale).sum().backward()
        gs = [t.grad.double() / scale for t in tabs]
        assert all(torch.isfinite(x).all() for x in gs)
        if ref is None:
            ref = gs
        else:
            for x, y in zip(gs, ref):
                assert (x - y).abs().max().item() < 1e-5 * y.abs().max().item(), scale


def test_nearly_identical_rows_vs_fp64_oracle(bf16x6):
    """A table whose rows are nearly identical (what meta_embedding_rel makes of bag-of-words rows that are almost all alike): the loss gradient
    is the small tangential remainder of a large radial sum.  With exact operands AND centred rows the mode is at least as accurate as the
    exact-fp32 MFMA sweep against the fp64 oracle, in the table gradient and in its column sums (= the bias gradient below it)."""
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
    for mode in ('f32', 'bf16x6'):
        ops.set_mfma_mode(mode)
        _, gg, _, _, _ = _run_overall([b.float().cuda() for b in base], dd, mods, w0.float().cuda(), lv1.float().cuda(), lv2.float().cuda(), fused=True)
        gref = eo['rel'].grad
        got = gg['rel'].cpu().double()
        err[mode] = ((got - gref).abs().max().item() / gref.abs().max().item(),
                     (got.sum(0) - gref.sum(0)).abs().max().item() / gref.sum(0).abs().max().item())
    ops.set_mfma_mode('bf16x6')
    print('nearly identical rows, (max entry, max column sum) error vs fp64:', err)
    assert err['bf16x6'][0] <= 1.25 * err['f32'][0] + 5e-7, err
    assert err['bf16x6'][1] <= 1.25 * err['f32'][1] + 5e-7, err


@pytest.mark.parametrize('sym', [True, False])
def test_stash_products_on_three_planes_equal_the_fp32_gemms(bf16x6, sym):
    """The anchors x anchors stash products on the sweeps' planes (stash3_kernel: fp32 coefficients split in registers, six bf16 MFMAs per
    product) against the four fp32-MFMA GEMMs they replace (ops.BF16X6_STASH = False), one-pass mode, symmetric and ordered walks, with a
    stash bound that forces several anchor-row blocks and ragged block edges: every table gradient and d fusion weight to fp32 rounding."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(40, 40, 4, seed=12, ragged=True, anchors='val')
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(9)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(3)]
    base[2] = torch.randn(1, 100, device='cuda', generator=g) + 1e-2 * torch.randn(T, 100, device='cuda', generator=g)   # nearly parallel rows: centred planes
    w0 = torch.tensor([[0.3], [1.1], [-0.4]], device='cuda')
    hint = torch.linspace(0.5, 1.5, 3 + 1 + 6, device='cuda')
    keep = (ops.BF16X6_STASH, ops.AA_SYMMETRIC, ops.STASH_BYTES)
    res = {}
    try:
        ops.AA_SYMMETRIC = sym
        s0 = ops.IndexSets.of(dd, 'cuda', T)
        ops.STASH_BYTES = 3 * 4 * s0.A * 160 * 2                      # ~160-row blocks
        for flag in (True, False):
            ops.BF16X6_STASH = flag
            tabs = [b.clone().requires_grad_(True) for b in base]
            w = w0.clone().requires_grad_(True)
            sums, s = ops.fused_contrastive_terms(tabs, w, dd, coef_hint=hint)
            (sums * hint).sum().backward()
            torch.cuda.synchronize()
            res[flag] = (sums.detach().double(), [t.grad.clone() for t in tabs], w.grad.clone())
    finally:
        ops.BF16X6_STASH, ops.AA_SYMMETRIC, ops.STASH_BYTES = keep
    a, b = res[False], res[True]
    assert torch.allclose(a[0], b[0], rtol=1e-7, atol=0)
    for k, (x, y) in enumerate(zip(a[1], b[1])):
        tol = 2e-5 if k < 2 else 1e-3          # (the centred table: the fp32 GEMM form carries the larger error, section 3g of DESIGN.md)
        assert (x - y).abs().max().item() < tol * x.abs().max().item(), (k, (x - y).abs().max().item(), x.abs().max().item())
    assert (a[2] - b[2]).abs().max().item() < 1e-4 * a[2].abs().max().item()


def test_mode_switch_roundtrip():
    from sgaligner_amd import ops
    d = ops.get_mfma_mode()
    assert d == ops.DEFAULT_MFMA_MODE == 'bf16x6'            # the default: fp32 arithmetic on three exact bf16 planes
    assert ops.MFMA_MODES == ('f32', 'bf16x6', 'f16')
    assert ops.set_mfma_mode('f32') == d and ops.get_mfma_mode() == 'f32'
    assert ops.set_mfma_mode('f16') == 'f32' and ops.get_mfma_mode() == 'f16'
    assert ops.set_mfma_mode(d) == 'f16'
    with pytest.raises(ValueError):
        ops.set_mfma_mode('f16x2')                           # the two-plane modes of rounds 2-5 are gone


def test_tables_wider_than_100_columns_take_the_fp32_kernels():
    """emb_dim 101..104 is accepted by the fused loss path, but columns 100, 101 of the three-plane blocks (and of the 'f32' mode's centred
    tables) carry the row centring's bookkeeping: such tables must run on the plain fp32 kernels EVERYWHERE -- sweeps, the stash products.
    Terms and gradients of the default mode equal the 'f32' mode's to fp32 summation noise, and columns 100..103 carry gradient."""
    mode = 'bf16x6'
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


@pytest.mark.parametrize('M', [3, 2, 4])
def test_lite_forward_sums_move_the_global_sums_by_less_than_1e6_and_are_used_only_for_large_batches(M):
    """ops.BF16X6_SUMS_LITE (csrc/sweep3.hip, LITE): the forward sums from the h and m planes only -- every similarity with an unbiased 2^-17
    rounding, which averages out over the terms of a global sum.  At 64 pairs x 64 objects (1e7 terms per sum) every loss term moves by < 1e-6
    relative and the gradients (which always multiply all six products) by < 1e-5 of their maximum; the DEFAULT policy takes the lite form only
    when the smallest global sum has >= 2^24 terms (reference arithmetic: src/aligner/losses.py:5-15).  M = 4 runs the one-wave-per-SIMD form of the
    kernel, whose operand requests ride on the MFMA loop (a skipped product must not skip its neighbour's request: round-6 fuzz finding)."""
    from sgaligner_amd import loss_ops, ops
    from sgaligner_amd.synthetic import make_batch
    assert ops.BF16X6_SUMS_LITE is None and ops.BF16X6_SUMS_LITE_MIN_TERMS == 1 << 24
    assert not loss_ops._sums_lite(2432, 4000, 4100) and loss_ops._sums_lite(9728, 46080, 46080) and loss_ops._sums_lite(19456, 368640, 368640)
    dd = make_batch(64, 64, 4, seed=31, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(8)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in range(M)]
    base[M - 1] = base[M - 1] * 0.05 + torch.randn(1, 100, device='cuda', generator=g)    # a table of nearly parallel rows (centred planes)
    w0 = torch.tensor([[0.4], [1.0], [-0.3], [0.7]], device='cuda')[:M].contiguous()
    hint = torch.linspace(0.5, 1.5, M + 1 + 2 * M, device='cuda')
    res = {}
    keep = ops.BF16X6_SUMS_LITE
    try:
        for lite in (False, True):
            ops.BF16X6_SUMS_LITE = lite
            ops.KERNEL_EVENTS = {}
            tabs = [b.clone().requires_grad_(True) for b in base]
            w = w0.clone().requires_grad_(True)
            sums, s = ops.fused_contrastive_terms(tabs, w, dd, coef_hint=hint)
            (sums * hint).sum().backward()
            torch.cuda.synchronize()
            res[lite] = (sums.detach().double(), [t.grad.clone() for t in tabs], w.grad.clone())
            assert ops.KERNEL_EVENTS['loss_multi_sums_bf16x6'][0][2][5] is lite                 # (the form asked for is the one that ran)
    finally:
        ops.BF16X6_SUMS_LITE = keep
        ops.KERNEL_EVENTS = None
    a, b = res[False], res[True]
    rel = ((a[0] - b[0]).abs() / a[0].abs().clamp_min(1e-300)).max().item()
    assert rel < 1e-6, rel
    for x, y in zip(a[1], b[1]):
        assert (x - y).abs().max().item() < 1e-5 * x.abs().max().item()
    assert (a[2] - b[2]).abs().max().item() < 1e-5 * a[2].abs().max().item()
