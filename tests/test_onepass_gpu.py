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
    run(True, key='icl_loss_unimodal')
    with pytest.raises(RuntimeError, match='FUSED_AA_ONEPASS'):
        ops.DEFERRED_CHECKS.flush()
    ops.DEFERRED_CHECKS.flush()
    run(False, key='icl_loss_unimodal')
    ops.DEFERRED_CHECKS.flush()


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
