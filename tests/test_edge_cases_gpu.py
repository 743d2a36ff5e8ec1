"""Degenerate batches through the whole HIP path (encoder -> fused loss -> backward -> Hits@K) against the oracle:
pairs without a single anchor, three-object scenes, a one-pair batch, one point per object."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

def drop_anchors_of_pair(dd, b):
    """pair b keeps no anchors: its anchor objects become negatives (a pair without common objects)"""
    e1c, e2c = dd['e1i_count'].copy(), dd['e2i_count'].copy()
    o1 = int(e1c[:b].sum()); n1 = int(e1c[b])
    j1o = int(dd['e1j_count'][:b + 1].sum()); j2o = int(dd['e2j_count'][:b + 1].sum())
    mv1, mv2 = dd['e1i'][o1:o1 + n1], dd['e2i'][o1:o1 + n1]
    dd['e1j'] = np.concatenate([dd['e1j'][:j1o], mv1, dd['e1j'][j1o:]]).astype(np.int32)
    dd['e2j'] = np.concatenate([dd['e2j'][:j2o], mv2, dd['e2j'][j2o:]]).astype(np.int32)
    dd['e1i'] = np.delete(dd['e1i'], np.s_[o1:o1 + n1]); dd['e2i'] = np.delete(dd['e2i'], np.s_[o1:o1 + n1])
    dd['e1j_count'] = dd['e1j_count'].copy(); dd['e2j_count'] = dd['e2j_count'].copy()
    dd['e1j_count'][b] += n1; dd['e2j_count'][b] += n1
    e1c[b] = 0; e2c[b] = 0
    dd['e1i_count'], dd['e2i_count'] = e1c, e2c
    return dd


def _cases():
    from sgaligner_amd.synthetic import make_batch
    return {
        'pair_without_anchors': lambda: drop_anchors_of_pair(make_batch(4, 9, 16, seed=1, ragged=True), 1),
        'first_pair_without_anchors': lambda: drop_anchors_of_pair(make_batch(3, 7, 16, seed=2, ragged=True), 0),
        'three_object_scenes': lambda: make_batch(5, 3, 16, seed=3),
        'single_pair': lambda: make_batch(1, 20, 16, seed=4),
        'one_point_per_object': lambda: make_batch(3, 10, 1, seed=5),
    }


@pytest.mark.parametrize('name', ['pair_without_anchors', 'first_pair_without_anchors', 'three_object_scenes', 'single_pair',
                                  'one_point_per_object'])
def test_degenerate_batches_vs_oracle(name):
    from oracle import sga_oracle as O
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    from sgaligner_amd.synthetic import to_device
    from sgaligner_amd.utils import alignment
    mods = ['point', 'gat', 'rel']
    dd = _cases()[name]()
    torch.manual_seed(0)
    model = MultiModalEncoder(modules=mods, rel_dim=41, attr_dim=164)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if 'num_batches' not in k}
    out_o, loss_o, grads_o = O.train_step(params, dd, mods)
    model = model.cuda()
    ddd = to_device(dd, 'cuda')
    lf = OverallLoss(CustomMultiLossLayer(3).cuda(), CustomMultiLossLayer(3).cuda(), 'cuda',
                     {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
    out = model(ddd)
    res = lf(out, ddd)
    res['loss'].backward()
    torch.cuda.synchronize()
    assert np.isfinite(res['loss'].item())
    assert abs(res['loss'].item() - loss_o['loss'].item()) < 1e-3 * max(1, abs(loss_o['loss'].item()))
    for n, p in model.named_parameters():
        if n in grads_o and p.grad is not None:
            assert (p.grad.cpu() - grads_o[n]).abs().max().item() < 1e-3 * max(1.0, grads_o[n].abs().max().item()), n
    mo = O.evaluate_batch(out_o['joint'].detach(), dd)
    mg = alignment.evaluate_batch(out['joint'].detach(), dd)
    assert [mg[k]['correct'] for k in (1, 2, 3, 4, 5)] == [mo['hits'][k][0] for k in (1, 2, 3, 4, 5)]


def test_edge_list_range_check_is_deferred_but_raises():
    """Edge endpoints outside every graph (global instead of graph-local node ids) must raise -- without a device read-back
    stalling every step: the check is queued, and fires at the next batch or at ops.DEFERRED_CHECKS.flush()."""
    import pytest
    import torch
    from sgaligner_amd import ops
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    from sgaligner_amd.synthetic import make_batch, to_device
    ops.DEFERRED_CHECKS.flush()
    dd = to_device(make_batch(2, 9, 16, seed=3), 'cuda')
    model = MultiModalEncoder(modules=['point', 'gat'], rel_dim=41, attr_dim=164).cuda()
    model(dd)
    ops.DEFERRED_CHECKS.flush()                                   # a good batch passes
    bad = dict(dd)
    bad['edges'] = dd['edges'].clone()
    bad['edges'][0, 0] = 1000
    model(bad)                                                    # the kernels drop the endpoint; the check is pending
    with pytest.raises(RuntimeError, match='graph-local'):
        ops.DEFERRED_CHECKS.flush()
    ops.DEFERRED_CHECKS.flush()                                   # reported once
