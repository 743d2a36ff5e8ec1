"""GPU parity: contrastive loss kernels (through the C-ABI) vs reference golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _overall(out, dd, mods, lv_ial, lv_icl):
    """The PRODUCT class on arbitrary tables: sgaligner_amd.aligner.losses.OverallLoss with the given log_vars.  The joint table of
    the goldens is an independent random tensor (not a fusion output), so this is the general per-table kernel path."""
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    m = len(mods)
    ial, icl = CustomMultiLossLayer(m).cuda(), CustomMultiLossLayer(m).cuda()
    if lv_ial is not None:
        ial.log_vars = torch.nn.Parameter(lv_ial)           # the test owns the leaves it reads gradients from
        icl.log_vars = torch.nn.Parameter(lv_icl)
    fn = OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': list(mods)})
    res = fn(out, dd)
    res['_lv'] = (ial.log_vars, icl.log_vars)
    return res


@pytest.mark.parametrize('tag', ['b1', 'b2', 'b4'])
def test_loss_golden(tag):
    g = load_golden('losses_' + tag)
    mods = [str(s) for s in g['modules']]
    out = {k: torch.from_numpy(g['emb_' + k]).cuda().requires_grad_(True) for k in mods + ['joint']}
    dd = {k: g[k] for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    lv_ial = torch.from_numpy(g['lv_ial']).cuda().requires_grad_(True)
    lv_icl = torch.from_numpy(g['lv_icl']).cuda().requires_grad_(True)
    res = _overall(out, dd, mods, lv_ial, lv_icl)
    res['loss'].backward()
    torch.cuda.synchronize()
    lv_ial, lv_icl = res['_lv']
    assert abs(res['loss'].item() - float(g['loss'])) < 1e-3 * max(1, abs(float(g['loss'])))
    assert abs(res['ial_loss'].item() - float(g['ial'])) < 1e-4 * max(1, abs(float(g['ial'])))
    assert abs(res['icl_loss_unimodal'].item() - float(g['icl_uni'])) < 1e-4 * max(1, abs(float(g['icl_uni'])))
    assert abs(res['icl_loss_multimodal'].item() - float(g['icl_multi'])) < 1e-4
    assert np.abs(lv_ial.grad.cpu().numpy() - g['g_lv_ial']).max() < 1e-4
    assert np.abs(lv_icl.grad.cpu().numpy() - g['g_lv_icl']).max() < 1e-4
    for k in mods + ['joint']:
        err = np.abs(out[k].grad.cpu().numpy() - g['g_' + k]).max()
        assert err < 1e-4, (k, err)


def test_loss_single_module_golden():
    g = load_golden('losses_m1')
    emb = torch.from_numpy(g['emb_point']).cuda().requires_grad_(True)
    dd = {k: g[k] for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    res = _overall({'point': emb}, dd, ['point'], None, None)
    res['loss'].backward()
    assert abs(res['loss'].item() - float(g['loss'])) < 1e-4
    assert np.abs(emb.grad.cpu().numpy() - g['g_point']).max() < 1e-4


@pytest.mark.parametrize('B,N,mods', [(3, 40, ['point', 'gat', 'rel']), (6, 64, ['point', 'rel']), (2, 150, ['point'])])
def test_loss_vs_oracle_fp64(B, N, mods):
    """multi-tile sizes (A, J not multiples of the 128/64 tiles) against the fp64 oracle"""
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(B, N, 8, seed=B * 100 + N, ragged=True, anchors='val')
    T = int(dd['tot_obj_count'].sum())
    torch.manual_seed(B + N)
    m = len(mods)
    out64 = {k: torch.randn(T, 100, dtype=torch.float64).requires_grad_(True) for k in mods}
    if m > 1:
        out64['joint'] = (0.5 * torch.randn(T, 100 * m, dtype=torch.float64)).requires_grad_(True)
    lv1 = (0.2 * torch.randn(m, dtype=torch.float64)).requires_grad_(True)
    lv2 = (0.2 * torch.randn(m, dtype=torch.float64)).requires_grad_(True)
    ref = O.overall_loss(out64, dd, mods, lv1, lv2)
    ref['loss'].backward()
    out = {k: v.detach().float().cuda().requires_grad_(True) for k, v in out64.items()}
    l1 = lv1.detach().float().cuda().requires_grad_(True)
    l2 = lv2.detach().float().cuda().requires_grad_(True)
    res = _overall(out, dd, mods, l1, l2)
    res['loss'].backward()
    torch.cuda.synchronize()
    l1, l2 = res['_lv']
    if m > 1:
        assert torch.allclose(l1.grad.cpu().double(), lv1.grad, rtol=1e-3, atol=1e-5) and torch.allclose(l2.grad.cpu().double(), lv2.grad, rtol=1e-3, atol=1e-5)
    assert abs(res['loss'].item() - ref['loss'].item()) < 1e-3 * max(1, abs(ref['loss'].item()))
    for k in out:
        gref = out64[k].grad
        err = (out[k].grad.cpu().double() - gref).abs().max().item()
        assert err < 1e-3 * max(1e-3, gref.abs().max().item()) + 1e-6, (k, err, gref.abs().max().item())


@pytest.mark.parametrize('m', [2, 3, 4])
def test_fused_loss_head_equals_torch_arithmetic(m):
    """ops.LossHeadFn (one launch forward, one backward) against the same arithmetic written as torch ops (losses.FUSED_HEAD =
    False): the four returned values and the gradients of every table, the fusion weight and both log_vars vectors; also with a
    cotangent on the logged values, not only on `loss`."""
    from sgaligner_amd.aligner import losses as L
    from sgaligner_amd.aligner.sg_aligner import MultiModalFusion
    from sgaligner_amd.synthetic import make_batch
    mods = ['point', 'gat', 'rel', 'attr'][:m]
    dd = make_batch(3, 17, 8, seed=5, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    torch.manual_seed(2)
    base = {k: torch.randn(T, 100, device='cuda') for k in mods}
    lv1, lv2 = 0.3 * torch.randn(m, device='cuda'), 0.3 * torch.randn(m, device='cuda')
    cot = torch.tensor([1.0, 0.3, -0.2, 0.7], device='cuda', dtype=torch.float64)
    res = {}
    for fused in (True, False):
        L.FUSED_HEAD = fused
        try:
            e = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            fus = MultiModalFusion(m).cuda()
            ial, icl = L.CustomMultiLossLayer(m).cuda(), L.CustomMultiLossLayer(m).cuda()
            with torch.no_grad():
                ial.log_vars.copy_(lv1); icl.log_vars.copy_(lv2)
            out = dict(e)
            out['joint'] = fus([e[k] for k in mods])
            fn = L.OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
            r = fn(out, dd)
            vals = torch.stack([r['loss'], r['icl_loss_unimodal'], r['icl_loss_multimodal'], r['ial_loss']]).double()
            (vals * cot).sum().backward()
            torch.cuda.synchronize()
            res[fused] = (vals.detach(), [e[k].grad for k in mods], fus.weight.grad, ial.log_vars.grad, icl.log_vars.grad)
        finally:
            L.FUSED_HEAD = True
    a, b = res[True], res[False]
    assert torch.allclose(a[0], b[0], rtol=1e-6, atol=1e-9), (a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert (x - y).abs().max() <= 1e-5 * max(1.0, y.abs().max().item())
    for x, y in zip(a[2:], b[2:]):
        assert (x - y).abs().max() <= 1e-5 * max(1.0, y.abs().max().item()), (x, y)
