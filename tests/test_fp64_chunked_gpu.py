"""Oracle-level evidence for the loss GRADIENT at the headline size (round-3 review item 3; reference src/aligner/losses.py:5-15,43-97):
tests/fp64_chunked.py -- a chunked fp64 evaluation of OverallLoss and its gradient in plain torch ops, independent of the library's loss
kernels -- is pinned on the oracle where the oracle runs, and then checks the DEFAULT product path (one-pass + symmetric anchors x anchors
walk + sweep16) at BASELINE configs[2], 4096 pairs x 128 objects x 512 points on one GPU: the four loss terms, dL/dE of sampled anchor /
negative rows of every table at 1e-3 of the row maximum, and every row in aggregate.  The same batch then carries the configs[2] part of the
f16x2 accuracy gate (errors against fp64, next to the exact-fp32 path's own) and the symmetric-vs-ordered walk comparison that used to live
in tools/dbg/c3_sym_vs_ordered.py."""
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
    for rows, rows_aa in ((1024, 128), (7, 5)):            # chunk sizes that do / do not divide the anchor count
        r = overall_loss_fp64([b.cuda() for b in base], w0, lv1, lv2, dd, rows=rows, rows_aa=rows_aa)
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
    """pairs = 4096 IS BASELINE configs[2] (the fp64 pass over its 2.3e11 anchor-negative pairs x 4 tables and 2.4e10 anchor pairs x 8 takes
    ~6 minutes of plain torch fp64 ops on the MI355X): it runs when SGA_TEST_C3_FP64=1 and its report is committed as
    profiles/r04_c3_gradient_vs_fp64.json; pairs = 1024 (same 128 objects x 512 points, A = 38 912, a sixteenth of the pair work) runs always."""
    if pairs == 4096 and os.environ.get('SGA_TEST_C3_FP64') != '1':
        pytest.skip('set SGA_TEST_C3_FP64=1 (about 7 GPU-minutes); last report: profiles/r04_c3_gradient_vs_fp64.json')
    from fp64_chunked import overall_loss_fp64
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch_fast
    from sgaligner_amd.trainer import AlignerSteps
    mods = ['point', 'gat', 'rel']
    torch.cuda.empty_cache()
    dd = make_batch_fast(pairs, 128, 512, seed=44, device='cuda')
    steps = AlignerSteps(mods, device='cuda', seed=42)
    assert ops.get_mfma_mode() == 'f32' and ops.FUSED_AA_ONEPASS and ops.AA_SYMMETRIC
    ops.DEFERRED_CHECKS.flush()
    ref32 = _step(steps, dd, mods)                                   # the default path: one-pass + symmetric walk + sweep16
    rerun = _step(steps, dd, mods)
    truth = overall_loss_fp64(ref32['tables'], steps.model.fusion.weight, steps.multi_loss_layer_ial.log_vars, steps.multi_loss_layer_icl.log_vars, dd)
    # ---- the four loss terms
    for k_t, k_p in (('loss', 'loss'), ('ial', 'ial_loss'), ('icl_uni', 'icl_loss_unimodal'), ('icl_multi', 'icl_loss_multimodal')):
        assert abs(ref32['loss'][k_p] - truth[k_t]) <= 2e-6 * abs(truth[k_t]), (k_t, ref32['loss'][k_p], truth[k_t])
    # ---- dL/dE of sampled anchor rows and sampled negative rows of every table, 1e-3 of the row maximum (and far better in aggregate)
    gen = torch.Generator().manual_seed(7)
    samp = {}
    for key in ('e1i', 'e2i', 'e1j', 'e2j'):
        ix = np.asarray(dd[key])
        samp[key] = torch.as_tensor(ix[torch.randperm(len(ix), generator=gen)[:32].numpy()], dtype=torch.long, device='cuda')
    rows = torch.cat([samp[k] for k in samp])
    report = {'sampled_rows_per_set': 32, 'tables': {}}
    for i, m in enumerate(mods):
        tr = truth['dE'][i]
        e_row = _row_err(ref32['dE'][i], tr, rows)
        e_all = float((ref32['dE'][i].double() - tr).abs().max() / tr.abs().max())
        assert e_row < 1e-3, (m, e_row)
        assert e_all < 1e-3, (m, e_all)
        report['tables'][m] = {'f32_sampled_row_err': e_row, 'f32_max_err_rel_to_max': e_all}
    assert torch.allclose(ref32['lv'][0].double(), truth['dlv_ial'], rtol=1e-4) and torch.allclose(ref32['lv'][1].double(), truth['dlv_icl'], rtol=1e-4)
    dw = steps.model.fusion.weight.grad
    assert (ref32['params']['fusion.weight'].double() - truth['dw']).abs().max() <= 1e-3 * truth['dw'].abs().max()
    # ---- parameters fed by nearly identical rows: meta_embedding_rel sees bag-of-words rows that are almost all alike, its gradient is a 1e-4-sized
    # remainder of 10^6-term sums.  fp64 truth of dW = dE_rel^T x, db = column sums, next to what each arithmetic makes of it.
    x_rel = dd['tot_bow_vec_object_edge_feats'].double()
    tw, tb = truth['dE'][2].t() @ x_rel, truth['dE'][2].sum(0)

    def rel_err(res):
        return {'weight': float((res['params']['meta_embedding_rel.weight'].double() - tw).abs().max() / tw.abs().max()),
                'bias': float((res['params']['meta_embedding_rel.bias'].double() - tb).abs().max() / tb.abs().max())}
    report['meta_embedding_rel_err_vs_fp64_rel_to_own_max'] = {'f32': rel_err(ref32), 'f32_rerun': rel_err(rerun)}
    report['meta_embedding_rel_f32_rerun_diff_rel_to_own_max'] = float(
        (rerun['params']['meta_embedding_rel.weight'] - ref32['params']['meta_embedding_rel.weight']).abs().max() / ref32['params']['meta_embedding_rel.weight'].abs().max())
    # ---- the split-fp16 mode on the same batch: errors against fp64 beside the exact-fp32 path's own (the configs[2] part of its gate)
    old = ops.set_mfma_mode('f16x2')
    try:
        r16 = _step(steps, dd, mods)
    finally:
        ops.set_mfma_mode(old)
    report['meta_embedding_rel_err_vs_fp64_rel_to_own_max']['f16x2'] = rel_err(r16)
    assert abs(r16['loss']['loss'] - truth['loss']) <= 2e-6 * abs(truth['loss'])
    for i, m in enumerate(mods):
        tr = truth['dE'][i]
        e16_row, e16_all = _row_err(r16['dE'][i], tr, rows), float((r16['dE'][i].double() - tr).abs().max() / tr.abs().max())
        e32_row, e32_all = report['tables'][m]['f32_sampled_row_err'], report['tables'][m]['f32_max_err_rel_to_max']
        report['tables'][m].update(f16x2_sampled_row_err=e16_row, f16x2_max_err_rel_to_max=e16_all)
        assert e16_row < 1e-3 and e16_all < 1e-3, (m, e16_row, e16_all)
        assert e16_all <= 2.0 * e32_all + 2e-7, (m, e16_all, e32_all)          # GATE: at most twice the exact-fp32 path's own error
    # every parameter: error against the exact-fp32 step, in units of that parameter's fp32 rerun difference
    worst = {}
    for n, gref in ref32['params'].items():
        own = float(gref.abs().max())
        noise = float((rerun['params'][n] - gref).abs().max()) / max(1e-30, own)
        err = float((r16['params'][n] - gref).abs().max()) / max(1e-30, own)
        worst[n] = (err, noise)
    report['f16x2_param_err_vs_f32_rel_to_own_max'] = {n: {'err': e, 'f32_rerun': z} for n, (e, z) in worst.items()}
    for n, (e, z) in worst.items():
        if n.startswith('meta_embedding_rel'):
            continue            # judged against fp64 above: the exact-fp32 path's own error there is far above its rerun difference
        assert e <= max(4.0 * z, 2e-4), (n, e, z)      # the tables' dL/dE (the only thing the mode changes) are held to 2x the fp32 error above
    e16, e32 = report['meta_embedding_rel_err_vs_fp64_rel_to_own_max']['f16x2'], report['meta_embedding_rel_err_vs_fp64_rel_to_own_max']['f32']
    assert e16['weight'] <= 4.0 * max(e32['weight'], 1e-3) and e16['bias'] <= 4.0 * max(e32['bias'], 1e-3), (e16, e32)
    # ---- symmetric vs ordered anchors x anchors walk (ordered = what N > 1 ranks run)
    keep = ops.AA_SYMMETRIC
    ops.AA_SYMMETRIC = False
    try:
        rord = _step(steps, dd, mods)
    finally:
        ops.AA_SYMMETRIC = keep
    assert abs(rord['loss']['loss'] - ref32['loss']['loss']) <= 1e-9 * abs(ref32['loss']['loss'])
    for i, m in enumerate(mods):
        tr = truth['dE'][i]
        eo_all = float((rord['dE'][i].double() - tr).abs().max() / tr.abs().max())
        report['tables'][m]['ordered_walk_max_err_rel_to_max'] = eo_all
        assert eo_all < 1e-3, (m, eo_all)
    ops.DEFERRED_CHECKS.flush()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    report['pairs'] = pairs
    with open(os.path.join(ROOT, 'gpurun_out', f'gradient_vs_fp64_{pairs}.json'), 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
