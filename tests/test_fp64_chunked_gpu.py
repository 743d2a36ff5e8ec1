"""Oracle-level evidence for the loss GRADIENT at the headline size (round-3 review item 3; reference src/aligner/losses.py:5-15,43-97):
tests/fp64_chunked.py -- a chunked fp64 evaluation of OverallLoss and its gradient in plain torch ops, independent of the library's loss
kernels -- is pinned on the oracle where the oracle runs, and then checks the DEFAULT product path (one-pass + symmetric anchors x anchors
walk + the three-plane bf16 sweeps) at BASELINE configs[2], 4096 pairs x 128 objects x 512 points on one GPU: the four loss terms, dL/dE of sampled anchor /
negative rows of every table at 1e-3 of the row maximum, and every row in aggregate -- for the default step AND for the same step with the
sweeps on the fp32 MFMA ('f32'), every parameter included (both arithmetics deliver the gradient in two parts: no exemption for
meta_embedding_rel.*).  The 1024-pair case also carries the symmetric-vs-ordered anchors x anchors walk comparison."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('B,N,seed', [(3, 30, 1), (8, 40, 2)])
def test_chunked_fp64_equals_oracle(B, N, seed):
    from fp64_chunked import overall_loss_fp64
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch
    mods = ['point', 'gat', 'rel']
    dd = make_batch(B, N, 1, seed=seed, ragged=True)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator().manual_seed(seed)
    base = [torch.randn(T, 100, generator=g, dtype=torch.float64) for _ in mods]
    w0 = torch.tensor([[0.7], [1.2], [0.9]], dtype=torch.float64)
    lv1 = torch.tensor([0.1, -0.2, 0.05], dtype=torch.float64)
    lv2 = torch.tensor([-0.1, 0.15, 0.0], dtype=torch.float64)
    eo = {k: base[i].clone().requires_grad_(True) for i, k in enumerate(mods)}
    wo, lo1, lo2 = w0.clone().requires_grad_(True), lv1.clone().requires_grad_(True), lv2.clone().requires_grad_(True)
    out_o = dict(eo)
    out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
    ref = O.overall_loss(out_o, dd, mods, lo1, lo2)
    ref['loss'].backward()
    # chunk sizes that do / do not divide the anchor count; the anchors x anchors part under autograd (literal formulas) and with the hand-derived
    # gradients the headline-size test below uses
    for rows, rows_aa, closed in ((1024, 128, False), (7, 5, False), (1024, 128, True), (7, 5, True)):
        r = overall_loss_fp64([b.cuda() for b in base], w0, lv1, lv2, dd, rows=rows, rows_aa=rows_aa, closed_form=closed)
        assert abs(r['loss'] - ref['loss'].item()) < 1e-10 * abs(ref['loss'].item())
        assert abs(r['ial'] - ref['ial_loss'].item()) < 1e-10 * abs(ref['ial_loss'].item())
        assert abs(r['icl_uni'] - ref['icl_loss_unimodal'].item()) < 1e-10 * abs(ref['icl_loss_unimodal'].item())
        assert abs(r['icl_multi'] - ref['icl_loss_multimodal'].item()) < 1e-10 * abs(ref['icl_loss_multimodal'].item())
        for i, k in enumerate(mods):
            assert (r['dE'][i].cpu() - eo[k].grad).abs().max().item() < 1e-9 * eo[k].grad.abs().max().item(), k
        assert (r['dw'].cpu() - wo.grad).abs().max().item() < 1e-9 * wo.grad.abs().max().item()
        assert torch.allclose(r['dlv_ial'].cpu(), lo1.grad, rtol=1e-10) and torch.allclose(r['dlv_icl'].cpu(), lo2.grad, rtol=1e-10)


def _step(steps, dd, mods):
    steps.zero_grad()
    out, loss = steps.train_step(0, 0, dd)
    for m in mods:
        out[m].retain_grad()
    loss['loss'].backward()
    torch.cuda.synchronize()
    res = {'loss': {k: float(v.detach()) for k, v in loss.items()}, 'dE': [out[m].grad.detach().clone() for m in mods],
           'tables': [out[m].detach().clone() for m in mods],
           'params': {n: p.grad.detach().clone() for n, p in steps.model.named_parameters() if p.grad is not None},
           'lv': [steps.multi_loss_layer_ial.log_vars.grad.detach().clone(), steps.multi_loss_layer_icl.log_vars.grad.detach().clone()]}
    del out, loss
    return res


def _row_err(g, ref, rows):
    """max over the sampled rows of (max |g - ref| / max |ref|) per row"""
    d = (g[rows].double() - ref[rows]).abs().amax(dim=1)
    return float((d / ref[rows].abs().amax(dim=1).clamp_min(1e-300)).max())


@pytest.mark.parametrize('pairs', [1024, 4096])
def test_headline_loss_gradient_vs_fp64(pairs):
    """pairs = 4096 IS BASELINE configs[2] (SGA_TEST_C3_FP64=0 skips it); pairs = 1024: same 128 objects x 512 points, A = 38 912, a
    sixteenth of the pair work.  Checked against the fp64 evaluation, on the same batch and weights:
      * the DEFAULT step (ops.DEFAULT_MFMA_MODE = 'bf16x6': the sweeps on three exact bf16 planes; one-pass + symmetric A x A walk) -- the
        four loss terms to 2e-6, dL/dE of sampled rows and of every row to 1e-3 of the maximum, d fusion weight, both d log_vars;
      * the same step with the sweeps on the fp32 MFMA ('f32', centred tables) to the same bars;
      * meta_embedding_rel.{weight, bias} -- the gradient of a table of nearly parallel rows, a 1e-4-sized tangential remainder of 10^6-term
        sums -- within 1e-3 of its own maximum against fp64 in BOTH arithmetics (reference src/aligner/losses.py:43-58 through F.normalize's
        backward; until round 6 the 'f32' step formed the radial part first and was 1.3 % off);
      * the default's table-gradient error is no larger than the fp32-MFMA step's, up to the two steps' run-to-run differences, and EVERY
        parameter of the default step lies within 4 x the fp32 step's rerun difference of it (floor 2e-4 of the parameter's maximum);
      * (1024 pairs only) the ordered A x A walk -- what N > 1 ranks run -- against fp64 as well."""
    if pairs == 4096 and os.environ.get('SGA_TEST_C3_FP64', '1') == '0':
        pytest.skip('SGA_TEST_C3_FP64=0')
    from fp64_chunked import overall_loss_fp64
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch_fast
    from sgaligner_amd.trainer import AlignerSteps
    mods = ['point', 'gat', 'rel']
    torch.cuda.empty_cache()
    dd = make_batch_fast(pairs, 128, 512, seed=44, device='cuda')
    steps = AlignerSteps(mods, device='cuda', seed=42)
    assert ops.get_mfma_mode() == ops.DEFAULT_MFMA_MODE == 'bf16x6' and ops.FUSED_AA_ONEPASS and ops.AA_SYMMETRIC and ops.CENTRED_F32
    ops.DEFERRED_CHECKS.flush()
    res = {'bf16x6': _step(steps, dd, mods)}                            # the default path: one-pass + symmetric walk + sweep3
    res['bf16x6_rerun'] = _step(steps, dd, mods)
    old = ops.set_mfma_mode('f32')
    # (this test is about the LOSS arithmetic: the object encoder keeps the default's forward -- three exact bf16 planes -- in both steps, so
    #  that both see the same tables and the same max-pool arg-maxes; the encoder's own parity: tests/test_pointnet_gpu.py)
    pn_f32 = ops._POINTNET_MODE['f32']
    ops._POINTNET_MODE['f32'] = ops._POINTNET_MODE['bf16x6']
    try:
        res['f32'] = _step(steps, dd, mods)                             # the same step with sweep16 (fp32 MFMA) over centred tables
        res['f32_rerun'] = _step(steps, dd, mods)
    finally:
        ops._POINTNET_MODE['f32'] = pn_f32
        ops.set_mfma_mode(old)
    ref32 = res['f32']
    truth = overall_loss_fp64(ref32['tables'], steps.model.fusion.weight, steps.multi_loss_layer_ial.log_vars, steps.multi_loss_layer_icl.log_vars, dd,
                              rows=4096 if pairs >= 4096 else 1024, timings=(timings := {}), closed_form=True, rows_aa=512 if pairs >= 4096 else 256)
    for i in range(len(mods)):
        assert torch.equal(res['bf16x6']['tables'][i], ref32['tables'][i])        # the encoder is the same arithmetic in both steps
    gen = torch.Generator().manual_seed(7)
    samp = {}
    for key in ('e1i', 'e2i', 'e1j', 'e2j'):
        ix = np.asarray(dd[key])
        samp[key] = torch.as_tensor(ix[torch.randperm(len(ix), generator=gen)[:32].numpy()], dtype=torch.long, device='cuda')
    rows = torch.cat([samp[k] for k in samp])
    report = {'pairs': pairs, 'fp64_evaluation_seconds': timings, 'sampled_rows_per_set': 32, 'default_mode': ops.DEFAULT_MFMA_MODE, 'tables': {m: {} for m in mods}}
    x_rel = dd['tot_bow_vec_object_edge_feats'].double()
    tw, tb = truth['dE'][2].t() @ x_rel, truth['dE'][2].sum(0)

    def rel_err(r):
        return {'weight': float((r['params']['meta_embedding_rel.weight'].double() - tw).abs().max() / tw.abs().max()),
                'bias': float((r['params']['meta_embedding_rel.bias'].double() - tb).abs().max() / tb.abs().max())}

    def rerun_diff(a, b, n):
        return float((a['params'][n] - b['params'][n]).abs().max() / a['params'][n].abs().max())
    report['meta_embedding_rel_err_vs_fp64_rel_to_own_max'] = {k: rel_err(v) for k, v in res.items()}
    report['meta_embedding_rel_rerun_diff_rel_to_own_max'] = {
        md: {n: rerun_diff(res[md], res[md + '_rerun'], 'meta_embedding_rel.' + n) for n in ('weight', 'bias')} for md in ('bf16x6', 'f32')}

    def save():
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', f'gradient_vs_fp64_{pairs}.json'), 'w') as f:
            json.dump(report, f, indent=1)
    for md in ('bf16x6', 'f32'):
        r = res[md]
        for i, m in enumerate(mods):
            tr = truth['dE'][i]
            e_row = _row_err(r['dE'][i], tr, rows)
            e_all = float((r['dE'][i].double() - tr).abs().max() / tr.abs().max())
            e_col = float((r['dE'][i].double().sum(0) - tr.sum(0)).abs().max() / tr.sum(0).abs().max())
            dcol = (r['dE'][i].double().sum(0) - tr.sum(0)) / tr.sum(0).abs().max()
            report['tables'][m].update({md + '_sampled_row_err': e_row, md + '_max_err_rel_to_max': e_all, md + '_column_sum_err_rel_to_max': e_col,
                                        md + '_column_sum_err_worst_columns': [int(c) for c in dcol.abs().topk(5).indices.tolist()],
                                        md + '_column_sum_err_signed_mean': float(dcol.mean()), md + '_column_sum_err_rms': float(dcol.pow(2).mean().sqrt())})
    save()                                  # (the evidence survives a failing gate)
    for md in ('bf16x6', 'f32'):
        r = res[md]
        # ---- the four loss terms
        for k_t, k_p in (('loss', 'loss'), ('ial', 'ial_loss'), ('icl_uni', 'icl_loss_unimodal'), ('icl_multi', 'icl_loss_multimodal')):
            assert abs(r['loss'][k_p] - truth[k_t]) <= 2e-6 * abs(truth[k_t]), (md, k_t, r['loss'][k_p], truth[k_t])
        # ---- dL/dE of sampled anchor rows and sampled negative rows of every table, 1e-3 of the row maximum (and far better in aggregate)
        for m in mods:
            t = report['tables'][m]
            assert t[md + '_sampled_row_err'] < 1e-3 and t[md + '_max_err_rel_to_max'] < 1e-3, (md, m, t)
        assert torch.allclose(r['lv'][0].double(), truth['dlv_ial'], rtol=1e-4) and torch.allclose(r['lv'][1].double(), truth['dlv_icl'], rtol=1e-4), md
        assert (r['params']['fusion.weight'].double() - truth['dw']).abs().max() <= 1e-3 * truth['dw'].abs().max(), md
        # ---- the parameter whose gradient is the tangential remainder of an almost radial table gradient: 1e-3 of its own maximum, both arithmetics
        e = report['meta_embedding_rel_err_vs_fp64_rel_to_own_max'][md]
        assert e['weight'] <= 1e-3 and e['bias'] <= 1e-3, (md, e)
    # ---- GATE of the default arithmetic: no less accurate than the fp32 MFMA, up to the run-to-run differences of the two steps
    for m in mods:
        t = report['tables'][m]
        assert t['bf16x6_max_err_rel_to_max'] <= 1.1 * t['f32_max_err_rel_to_max'] + 1e-7, (m, t)
        # column sums (what the bias gradients of the layers below collect over 10^6 rows): a random-walk statistic of ~1e-5 of the largest
        # column sum in either arithmetic, the same worst columns in both; measured ratios default / fp32-MFMA over round 5's runs: 0.4 .. 1.0
        # at 1024 pairs, 0.97 .. 1.6 at 4096 (2.9e-5 vs 1.8e-5 for `point`) -- bounded at 2 x
        assert t['bf16x6_column_sum_err_rel_to_max'] <= 2.0 * t['f32_column_sum_err_rel_to_max'] + 1e-6, (m, t)
    worst = {}
    for n, gref in ref32['params'].items():
        own = float(gref.abs().max())
        noise = float((res['f32_rerun']['params'][n] - gref).abs().max()) / max(1e-30, own)
        err = float((res['bf16x6']['params'][n] - gref).abs().max()) / max(1e-30, own)
        worst[n] = (err, noise)
    report['bf16x6_param_diff_vs_f32_rel_to_own_max'] = {n: {'diff': e, 'f32_rerun': z} for n, (e, z) in worst.items()}
    save()
    for n, (e, z) in worst.items():
        assert e <= max(4.0 * z, 2e-4 if not n.startswith('meta_embedding_rel') else 1e-3), (n, e, z)
    if pairs <= 1024:
        # ---- symmetric vs ordered anchors x anchors walk (ordered = what N > 1 ranks run), default arithmetic
        keep = ops.AA_SYMMETRIC
        ops.AA_SYMMETRIC = False
        try:
            rord = _step(steps, dd, mods)
        finally:
            ops.AA_SYMMETRIC = keep
        assert abs(rord['loss']['loss'] - res['bf16x6']['loss']['loss']) <= 1e-9 * abs(res['bf16x6']['loss']['loss'])
        for i, m in enumerate(mods):
            tr = truth['dE'][i]
            eo_all = float((rord['dE'][i].double() - tr).abs().max() / tr.abs().max())
            report['tables'][m]['ordered_walk_max_err_rel_to_max'] = eo_all
            assert eo_all < 1e-3, (m, eo_all)
    ops.DEFERRED_CHECKS.flush()
    save()
    print(json.dumps(report, indent=1))
