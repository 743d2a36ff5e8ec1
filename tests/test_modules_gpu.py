"""GPU parity at module level: the drop-in aligner API (MultiModalEncoder / OverallLoss / alignment metrics)
against the reference-generated golden vectors and the oracle on synthetic batches."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _load_sd(model, g, prefix='sd__'):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    model.load_state_dict(sd, strict=True)          # strict, like engine/base_tester.py:61


def test_example_pair_c1_point_only():
    """BASELINE.json configs[0] on the HIP path: embeddings, loss, grads, Hits@K / MRR / SGAR vs the reference."""
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    from sgaligner_amd.utils import alignment
    g = load_golden('example_pair_point')
    ns, nr = [int(v) for v in g['counts']]
    model = MultiModalEncoder(modules=['point'], rel_dim=41, attr_dim=164).cuda()
    _load_sd(model, g)
    dd = {'tot_obj_pts': torch.from_numpy(g['pts']).cuda(), 'batch_size': 1, 'e1i': g['e1i'], 'e2i': g['e2i'],
          'e1j': g['e1j'], 'e2j': g['e2j'], 'tot_obj_count': np.array([ns + nr]), 'e1i_count': np.array([len(g['e1i'])]),
          'graph_per_obj_count': np.array([[ns, nr]])}
    out = model(dd)
    loss_fn = OverallLoss(CustomMultiLossLayer(1), CustomMultiLossLayer(1), 'cuda',
                          {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': ['point']})
    res = loss_fn(out, dd)
    res['loss'].backward()
    torch.cuda.synchronize()
    assert np.abs(out['point'].detach().cpu().numpy() - g['emb']).max() < TOL
    assert abs(res['loss'].item() - float(g['loss'])) < TOL
    for name, p in model.named_parameters():
        key = 'grad__' + name
        if key in g:
            ref = g[key]
            assert np.abs(p.grad.cpu().numpy() - ref).max() < TOL * max(1.0, np.abs(ref).max()), name
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    m = alignment.evaluate_batch(out['point'].detach(), dd, reg_k=2)
    assert np.allclose(m['mrr'], g['mrr'])
    assert [m[k]['correct'] for k in (1, 2, 3, 4, 5)] == [int(v) for v in g['hits']]
    assert [m['sgar'][k][0] for k in ('2', '50', '100')] == [float(v) for v in g['sgar']]
    assert m['node_corrs'][0] == [tuple(int(x) for x in c) for c in g['node_corrs']]


def test_full_multimodal_golden():
    """P+S+R+A through the drop-in classes vs the reference's own orchestration (GAT layer GAT-UNPINNED)."""
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    g = load_golden('full_multimodal_gat_unpinned')
    mods = ['point', 'gat', 'rel', 'attr']
    model = MultiModalEncoder(modules=mods, rel_dim=41, attr_dim=164).cuda()
    assert set(model.state_dict().keys()) == set(str(s) for s in g['sd_keys'])
    _load_sd(model, g)
    dd = {}
    for k, v in g.items():
        if k.startswith('dd__'):
            name = k[4:]
            dd[name] = torch.from_numpy(v).cuda() if (name.startswith('tot_') and name != 'tot_obj_count') or name == 'edges' else v
    dd['batch_size'] = 2
    out = model(dd)
    ial, icl = CustomMultiLossLayer(4).cuda(), CustomMultiLossLayer(4).cuda()
    loss_fn = OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
    res = loss_fn(out, dd)
    res['loss'].backward()
    torch.cuda.synchronize()
    for k in mods + ['joint']:
        assert np.abs(out[k].detach().cpu().numpy() - g['out__' + k]).max() < TOL, k
    for key, ref in (('loss', 'loss'), ('icl_loss_unimodal', 'icl_uni'), ('icl_loss_multimodal', 'icl_multi'), ('ial_loss', 'ial')):
        assert abs(res[key].item() - float(g[ref])) < TOL * max(1.0, abs(float(g[ref]))), key
    assert np.abs(ial.log_vars.grad.cpu().numpy() - g['g_lv_ial']).max() < TOL
    assert np.abs(icl.log_vars.grad.cpu().numpy() - g['g_lv_icl']).max() < TOL
    seen = 0
    for name, p in model.named_parameters():
        key = 'grad__' + name
        if key in g:
            ref = g[key]
            err = np.abs(p.grad.cpu().numpy() - ref).max()
            assert err < TOL * max(1.0, np.abs(ref).max()), (name, err)
            seen += 1
    assert seen >= 20


@pytest.mark.parametrize('B,N,P,mods', [(3, 20, 64, ['point', 'gat', 'rel']), (2, 33, 40, ['point', 'gat', 'rel', 'attr'])])
def test_train_step_vs_oracle(B, N, P, mods):
    from oracle import sga_oracle as O
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    from sgaligner_amd.synthetic import make_batch, to_device
    from sgaligner_amd.utils import alignment
    dd = make_batch(B, N, P, seed=B + N, ragged=True)
    torch.manual_seed(1)
    model = MultiModalEncoder(modules=mods, rel_dim=41, attr_dim=164)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if 'num_batches' not in k}
    out_o, loss_o, grads_o = O.train_step(params, dd, mods)
    model = model.cuda()
    ddd = to_device(dd, 'cuda')
    m = len(mods)
    loss_fn = OverallLoss(CustomMultiLossLayer(m).cuda(), CustomMultiLossLayer(m).cuda(), 'cuda',
                          {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
    out = model(ddd)
    res = loss_fn(out, ddd)
    res['loss'].backward()
    torch.cuda.synchronize()
    for k in out_o:
        assert (out[k].detach().cpu() - out_o[k].detach()).abs().max() < TOL, k
    assert abs(res['loss'].item() - loss_o['loss'].item()) < TOL * max(1, abs(loss_o['loss'].item()))
    for name, p in model.named_parameters():
        if name in grads_o and p.grad is not None:
            ref = grads_o[name]
            err = (p.grad.cpu() - ref).abs().max().item()
            assert err < TOL * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())
    # Hits@K / MRR parity on the joint embedding (val-style: every common object is an anchor)
    ddv = make_batch(B, N, P, seed=B + N, ragged=True, anchors='val')
    mo = O.evaluate_batch(out_o['joint'].detach(), ddv)
    mg = alignment.evaluate_batch(out['joint'].detach(), ddv)
    assert [mg[k]['correct'] for k in (1, 2, 3, 4, 5)] == [mo['hits'][k][0] for k in (1, 2, 3, 4, 5)]
    assert np.allclose(mg['mrr'], mo['mrr'])
