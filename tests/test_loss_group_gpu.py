"""loss_group = b: the batch is cut into groups of b consecutive pairs and the reference's loss is evaluated
on each group independently -- what the reference computes when its trainer feeds b pairs per iteration
(configs/scan3r/scan3r_ground_truth.yaml:27, src/engine/epoch_based_trainer.py:91-93, src/aligner/losses.py:114-152) --
compared NUMBER FOR NUMBER with the oracle run b pairs at a time (loss terms and every gradient summed over the groups)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_groups(base, w0, lv1, lv2, dd, mods, b):
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    eo = {k: base[k].clone().requires_grad_(True) for k in mods}
    wo = w0.clone().requires_grad_(True) if len(mods) > 1 else None
    l1, l2 = lv1.clone().requires_grad_(True), lv2.clone().requires_grad_(True)
    tot = None
    for gd in ops.group_data_dicts(dd, b):
        out_o = dict(eo)
        if len(mods) > 1:
            out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
        r = O.overall_loss(out_o, gd, mods, l1, l2)
        tot = r if tot is None else {k: tot[k] + r[k] for k in r}
    tot['loss'].backward()
    return tot, eo, wo, l1, l2


def _hip_groups(base, w0, lv1, lv2, dd, mods, b, fused=True):
    from sgaligner_amd.aligner import losses as L
    from sgaligner_amd.aligner.sg_aligner import MultiModalFusion
    m = len(mods)
    L.FUSED_JOINT = fused
    try:
        e = {k: base[k].float().cuda().requires_grad_(True) for k in mods}
        ial, icl = L.CustomMultiLossLayer(m).cuda(), L.CustomMultiLossLayer(m).cuda()
        with torch.no_grad():
            ial.log_vars.copy_(lv1.float()); icl.log_vars.copy_(lv2.float())
        out = dict(e)
        fus = None
        if m > 1:
            fus = MultiModalFusion(m).cuda()
            with torch.no_grad():
                fus.weight.copy_(w0.float())
            out['joint'] = fus([e[k] for k in mods])
        fn = L.OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods,
                                              'loss_group': b})
        res = fn(out, dd)
        res['loss'].backward()
        torch.cuda.synchronize()
        return res, e, fus, ial, icl
    finally:
        L.FUSED_JOINT = True


@pytest.mark.parametrize('mods,B,b', [(['point', 'gat', 'rel'], 8, 2), (['point', 'gat', 'rel'], 8, 4), (['point', 'gat', 'rel', 'attr'], 7, 3),
                                      (['point', 'rel'], 6, 2), (['point'], 6, 2), (['point', 'gat', 'rel'], 5, 1)])
def test_grouped_loss_vs_oracle_b_pairs_at_a_time(mods, B, b):
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(B, 17, 4, seed=B * 10 + b, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    m = len(mods)
    torch.manual_seed(B + b)
    base = {k: torch.randn(T, 100, dtype=torch.float64) for k in mods}
    w0 = torch.tensor([[0.4], [1.3], [-0.2], [0.8]], dtype=torch.float64)[:m]
    lv1, lv2 = 0.2 * torch.randn(m, dtype=torch.float64), 0.2 * torch.randn(m, dtype=torch.float64)
    ref, eo, wo, l1, l2 = _oracle_groups(base, w0, lv1, lv2, dd, mods, b)
    for fused in ((True, False) if m > 1 else (True,)):           # grouped kernels / the per-group loop over the general path
        res, e, fus, ial, icl = _hip_groups(base, w0, lv1, lv2, dd, mods, b, fused=fused)
        for key in ('loss', 'icl_loss_unimodal', 'icl_loss_multimodal', 'ial_loss'):
            r, g = float(ref[key]), float(res[key])
            assert abs(g - r) < 1e-4 * max(1.0, abs(r)), (fused, key, g, r)
        for k in mods:
            gref = eo[k].grad
            err = (e[k].grad.cpu().double() - gref).abs().max().item()
            assert err < 1e-3 * max(1e-6, gref.abs().max().item()), (fused, k, err, gref.abs().max().item())
        if m > 1:
            assert (fus.weight.grad.cpu().double() - wo.grad).abs().max().item() < 1e-3 * max(1e-3, wo.grad.abs().max().item()), fused
            assert torch.allclose(ial.log_vars.grad.cpu().double(), l1.grad, rtol=1e-3, atol=1e-6), fused
            assert torch.allclose(icl.log_vars.grad.cpu().double(), l2.grad, rtol=1e-3, atol=1e-6), fused


def test_group_covering_the_whole_batch_is_the_global_loss():
    from sgaligner_amd.synthetic import make_batch
    mods = ['point', 'gat', 'rel']
    dd = make_batch(6, 20, 4, seed=3, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    torch.manual_seed(0)
    base = {k: torch.randn(T, 100, dtype=torch.float64) for k in mods}
    w0 = torch.tensor([[0.4], [1.3], [-0.2]], dtype=torch.float64)
    lv = torch.zeros(3, dtype=torch.float64)
    rg, eg, fg, _, _ = _hip_groups(base, w0, lv, lv, dd, mods, 6)
    rglob, e2, f2, _, _ = _hip_groups(base, w0, lv, lv, dd, mods, 'global')
    assert abs(float(rg['loss']) - float(rglob['loss'])) < 1e-5 * abs(float(rglob['loss']))
    for k in mods:
        sc = e2[k].grad.abs().max().item()
        assert (eg[k].grad - e2[k].grad).abs().max().item() < 1e-4 * sc, k
    assert (fg.weight.grad - f2.weight.grad).abs().max().item() < 1e-4 * max(1e-3, f2.weight.grad.abs().max().item())


@pytest.mark.parametrize('b', [2, 4])
def test_train_step_b64_loss_group_vs_reference_sized_oracle_steps(b):
    """B = 64 pairs on the device in ONE step with loss_group = b  ==  the oracle (the pinned restatement of the reference)
    run on 64/b consecutive b-pair batches with gradients summed: loss and every parameter gradient, number for number."""
    from oracle import sga_oracle as O
    from sgaligner_amd import dist as sdist
    from sgaligner_amd.synthetic import make_batch, to_device
    from sgaligner_amd.trainer import AlignerSteps
    mods = ['point', 'gat', 'rel']
    B = 64
    dd = make_batch(B, 12, 32, seed=77, ragged=True)
    steps = AlignerSteps(mods, device='cuda', seed=5, loss_group=b)
    with torch.no_grad():
        steps.multi_loss_layer_ial.log_vars.copy_(torch.tensor([0.1, -0.2, 0.05]))
        steps.multi_loss_layer_icl.log_vars.copy_(torch.tensor([-0.1, 0.15, 0.0]))
    params = {k: v.detach().cpu().clone() for k, v in steps.model.state_dict().items() if 'num_batches' not in k}
    lv_ial = steps.multi_loss_layer_ial.log_vars.detach().cpu().clone()
    lv_icl = steps.multi_loss_layer_icl.log_vars.detach().cpu().clone()
    tot_loss, grads, g_ial, g_icl = 0.0, {}, torch.zeros(3), torch.zeros(3)
    for lo in range(0, B, b):
        sub = sdist.shard_data_dict(dd, lo, min(B, lo + b))
        _, loss_o, g_o = O.train_step(params, sub, mods, log_vars_ial=lv_ial, log_vars_icl=lv_icl)
        tot_loss += float(loss_o['loss'])
        for k, v in g_o.items():
            if k == 'log_vars_ial':
                g_ial += v
            elif k == 'log_vars_icl':
                g_icl += v
            else:
                grads[k] = grads.get(k, 0) + v
    out, loss = steps.forward_backward(to_device(dd, 'cuda'))
    torch.cuda.synchronize()
    assert abs(float(loss['loss']) - tot_loss) < 1e-4 * max(1.0, abs(tot_loss)), (float(loss['loss']), tot_loss)
    seen = 0
    for name, p in steps.model.named_parameters():
        if name in grads and p.grad is not None:
            ref = grads[name]
            err = (p.grad.cpu() - ref).abs().max().item()
            assert err < 1e-3 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())
            seen += 1
    assert seen >= 15
    assert torch.allclose(steps.multi_loss_layer_ial.log_vars.grad.cpu(), g_ial, rtol=1e-3, atol=1e-5)
    assert torch.allclose(steps.multi_loss_layer_icl.log_vars.grad.cpu(), g_icl, rtol=1e-3, atol=1e-5)
