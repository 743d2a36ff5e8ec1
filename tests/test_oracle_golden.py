"""CPU: the oracle (oracle/sga_oracle.py) against vectors produced by the REFERENCE ITSELF
(oracle/make_golden.py imported /root/reference; the vectors travel, the reference does not)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import sga_oracle as O

T = torch.from_numpy
TOL = 2e-6


def close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, f'max abs err {err} (scale {scale})'


@pytest.mark.parametrize('tag', ['small', 'ragged'])
def test_pointnet_forward_backward_bn(tag):
    g = load_golden('pointnet_' + tag)
    ws = [T(g[k]).clone().requires_grad_(True) for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')]
    x = T(g['x'])
    y = O.pointnet_feat(x, *ws)
    close(y, g['y'])
    close(y, g['y_eval'])          # BN output is discarded -> train == eval (pointnet.py:141-142)
    (y * T(g['cot'])).sum().backward()
    for w, k in zip(ws, ('gw1', 'gb1', 'gw2', 'gb2', 'gw3', 'gb3')):
        close(w.grad, g[k], 1e-5)
    assert bool(g['bn_w_grad_is_none'][0])
    stats = O.pointnet_bn_batch_stats(x, *[w.detach() for w in ws])
    for i, (m, v) in enumerate(stats):
        close(0.1 * m, g[f'rm{i+1}'], 1e-5)
        close(0.9 + 0.1 * v, g[f'rv{i+1}'], 1e-5)


@pytest.mark.parametrize('m', [2, 3, 4])
def test_fusion(m):
    g = load_golden(f'fusion_m{m}')
    embs = [T(g[f'emb{i}']).clone().requires_grad_(True) for i in range(m)]
    w = T(g['weight']).clone().requires_grad_(True)
    j = O.fusion(embs, w)
    close(j, g['joint'])
    (j * T(g['cot'])).sum().backward()
    close(w.grad, g['gweight'], 1e-5)
    for i, e in enumerate(embs):
        close(e.grad, g[f'gemb{i}'], 1e-5)


@pytest.mark.parametrize('tag', ['b1', 'b2', 'b4'])
def test_losses(tag):
    g = load_golden('losses_' + tag)
    mods = [str(s) for s in g['modules']]
    out = {k: T(g['emb_' + k]).clone().requires_grad_(True) for k in mods + ['joint']}
    dd = {k: g[k] for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    lv_ial = T(g['lv_ial']).clone().requires_grad_(True)
    lv_icl = T(g['lv_icl']).clone().requires_grad_(True)
    res = O.overall_loss(out, dd, mods, lv_ial, lv_icl)
    close(res['loss'], g['loss'], 1e-5)
    close(res['icl_loss_unimodal'], g['icl_uni'], 1e-5)
    close(res['icl_loss_multimodal'], g['icl_multi'], 1e-5)
    close(res['ial_loss'], g['ial'], 1e-5)
    res['loss'].backward()
    close(lv_ial.grad, g['g_lv_ial'], 1e-5)
    close(lv_icl.grad, g['g_lv_icl'], 1e-5)
    for k in mods + ['joint']:
        close(out[k].grad, g['g_' + k], 2e-5)
    e = torch.nn.functional.normalize(out[mods[0]].detach(), dim=1)
    ix = {k: torch.as_tensor(v, dtype=torch.long) for k, v in dd.items()}
    q = O.calculate_prob_dist(e[ix['e1i']], e[ix['e2i']], e[ix['e1j']], e[ix['e2j']], 0.1)
    close(q, g['q_first_t01'], 1e-6)
    close(O.icl_loss(out[mods[0]].detach(), dd), g['icl_first'], 1e-5)
    close(O.ial_loss(out[mods[0]].detach(), out['joint'].detach(), dd), g['ial_first'], 1e-5)


def test_losses_fp64_agrees():
    """fp64 oracle vs the reference's fp32 numbers: bounds the reference's own rounding."""
    g = load_golden('losses_b2')
    mods = [str(s) for s in g['modules']]
    out = {k: T(g['emb_' + k]).double() for k in mods + ['joint']}
    dd = {k: g[k] for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    res = O.overall_loss(out, dd, mods, T(g['lv_ial']).double(), T(g['lv_icl']).double())
    close(res['loss'], g['loss'].astype(np.float64), 1e-5)


def test_losses_single_module():
    g = load_golden('losses_m1')
    emb = T(g['emb_point']).clone().requires_grad_(True)
    dd = {k: g[k] for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    res = O.overall_loss({'point': emb}, dd, ['point'])
    close(res['loss'], g['loss'], 1e-5)
    res['loss'].backward()
    close(emb.grad, g['g_point'], 2e-5)


def _params_from(g, prefix='sd__'):
    return {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}


def test_example_pair_c1():
    """BASELINE.json configs[0]: example_data scene_1/scene_2, 256 pts/object, ['point'], CPU."""
    g = load_golden('example_pair_point')
    params = _params_from(g)
    ns, nr = [int(v) for v in g['counts']]
    dd = {'tot_obj_pts': T(g['pts']), 'batch_size': 1, 'e1i': g['e1i'], 'e2i': g['e2i'], 'e1j': g['e1j'],
          'e2j': g['e2j'], 'tot_obj_count': np.array([ns + nr]), 'e1i_count': np.array([len(g['e1i'])]),
          'tot_bow_vec_object_attr_feats': torch.zeros(ns + nr, 164), 'tot_rel_pose': torch.zeros(ns + nr, 3),
          'tot_bow_vec_object_edge_feats': torch.zeros(ns + nr, 41)}
    out, loss, grads = O.train_step(params, dd, ['point'])
    close(out['point'], g['emb'], 1e-5)
    close(loss['loss'], g['loss'], 1e-5)
    for k, v in g.items():
        if k.startswith('grad__'):
            close(grads[k[6:]], v, 5e-5)
    sim, _ = O.pair_similarity(out['point'].detach())
    close(sim, g['sim'], 1e-5)
    res = O.evaluate_batch(out['point'].detach(), dd)
    close(np.array(res['mrr']), g['mrr'], 1e-12)
    assert [res['hits'][k][0] for k in (1, 2, 3, 4, 5)] == [int(v) for v in g['hits']]
    assert [res['sgar'][m][0] for m in ('2', '50', '100')] == [float(v) for v in g['sgar']]
    assert [tuple(c) for c in O.node_corrs(sim, ns, 2)] == [tuple(int(x) for x in c) for c in g['node_corrs']]


def test_alignment_handmade():
    g = load_golden('alignment_handmade')
    m = O.alignment_metrics(T(g['sim']), g['e1i'], g['e2i'])
    close(np.array(m['mrr']), g['mrr'], 1e-12)
    assert [m['hits'][k][0] for k in (1, 2, 3, 4, 5)] == [int(v) for v in g['hits']]
    assert [m['sgar'][k] for k in ('2', '50', '100')] == [float(v) for v in g['sgar']]
    ns = int(g['src_count'])
    assert [tuple(c) for c in O.node_corrs(T(g['sim']), ns, 3)] == [tuple(int(x) for x in c) for c in g['node_corrs']]


def test_full_multimodal_gat_unpinned():
    """Reference orchestration (sg_aligner.py:71-137 + losses.py:114-152) end to end; the GAT layer in
    the golden is the oracle's own restatement (GAT-UNPINNED), everything else is the reference's."""
    g = load_golden('full_multimodal_gat_unpinned')
    params = _params_from(g)
    dd = {k[4:]: (T(v) if k[4:].startswith('tot_') and k[4:] != 'tot_obj_count' or k[4:] == 'edges' else v)
          for k, v in g.items() if k.startswith('dd__')}
    dd['batch_size'] = 2
    mods = ['point', 'gat', 'rel', 'attr']
    out, loss, grads = O.train_step(params, dd, mods)
    for k in mods + ['joint']:
        close(out[k], g['out__' + k], 1e-5)
    close(loss['loss'], g['loss'], 1e-5)
    close(loss['ial_loss'], g['ial'], 1e-5)
    for k, v in g.items():
        if k.startswith('grad__'):
            name = k[6:]
            if name.endswith('lin_dst.weight'):
                continue
            close(grads[name], v, 1e-4)
    expected = {k[4:] for k in g if k.startswith('sd__')}
    assert set(str(s) for s in g['sd_keys']) == expected


def test_gat_edge_vs_dense():
    """Independent cross-check of the un-pinned GATConv restatement: edge-list vs dense-count forms,
    with duplicate edges, explicit self loops in the input, and an isolated node."""
    torch.manual_seed(0)
    n = 9
    x = torch.randn(n, 3, dtype=torch.float64)
    src = torch.tensor([0, 1, 2, 3, 3, 3, 4, 5, 5, 6, 2, 2, 7, 0])
    dst = torch.tensor([1, 0, 1, 1, 1, 3, 4, 6, 6, 5, 0, 0, 2, 7])      # dup edges, self loops, node 8 isolated
    ei = torch.stack([src, dst])
    p = O.init_params(['point', 'gat'], dtype=torch.float64)
    layers = O._gat_layers(p)
    for l in layers:
        l['bias'] = torch.randn_like(l['bias']) * 0.1
    a = O.multi_gat(x, ei, layers, conv=O.gat_conv)
    b = O.multi_gat(x, ei, layers, conv=O.gat_conv_dense)
    close(a, b, 1e-12)


def test_gat_oracle_vs_hand_derived_case():
    """Row G's oracle against numbers derived by hand from the published GATConv definition (tests/gat_handcase.py: 3 nodes,
    2 heads, a duplicate edge, an explicit self loop in the input) -- an implementation-independent pin of the restatement."""
    import gat_handcase as G
    h, a_s, a_d, b = G.inputs()
    # x = one-hot node features and lin_w[:, j] = h_j make  x @ lin_w^T == h  exactly
    x = torch.eye(3, dtype=torch.float64)
    lin_w = torch.from_numpy(h.T.copy())
    ei = torch.from_numpy(G.EDGES.T.copy())
    args = (lin_w, torch.from_numpy(a_s).view(1, G.H, G.C), torch.from_numpy(a_d).view(1, G.H, G.C), torch.from_numpy(b))
    for conv in (O.gat_conv, O.gat_conv_dense):
        out = conv(x, ei, *args)
        assert np.allclose(out.numpy(), G.expected(), rtol=0, atol=1e-13), conv.__name__


def test_gat_gradcheck_fp64():
    torch.manual_seed(1)
    n = 5
    x = torch.randn(n, 3, dtype=torch.float64)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
    m = ii != jj
    ei = torch.from_numpy(np.stack([ii[m], jj[m]]))
    w = torch.randn(8, 3, dtype=torch.float64, requires_grad=True)
    a_s = torch.randn(1, 2, 4, dtype=torch.float64, requires_grad=True)
    a_d = torch.randn(1, 2, 4, dtype=torch.float64, requires_grad=True)
    b = torch.randn(8, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: O.gat_conv(x, ei, *a), (w, a_s, a_d, b))
