"""Opt-in split-bf16 x3 MFMA mode (ops.set_mfma_mode('bf16x3'); the default stays exact fp32): PointNet object encoder forward on
v_mfma_f32_32x32x16_bf16 with every fp32 operand split into bf16 hi + lo.  Bar: embeddings, loss and
gradients within 1e-3 of the reference goldens / the oracle and identical Hits@K."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16x3():
    from sgaligner_amd import ops
    old = ops.set_mfma_mode('bf16x3')
    yield
    ops.set_mfma_mode(old)


def test_mode_switch_roundtrip():
    from sgaligner_amd import ops
    d = ops.get_mfma_mode()
    assert d == ops.DEFAULT_MFMA_MODE == 'bf16x6'            # the default: fp32 arithmetic on three exact bf16 planes (tests/test_bf16x6_gpu.py)
    assert ops.set_mfma_mode('bf16x3') == d and ops.get_mfma_mode() == 'bf16x3'
    assert ops.set_mfma_mode('f32') == 'bf16x3' and ops.get_mfma_mode() == 'f32'
    assert ops.set_mfma_mode(d) == 'f32'
    with pytest.raises(ValueError):
        ops.set_mfma_mode('fp8')


@pytest.mark.parametrize('tag', ['pointnet_small', 'pointnet_ragged'])
def test_pointnet_forward_vs_reference_golden(bf16x3, tag):
    from sgaligner_amd import ops
    g = load_golden(tag)
    x = torch.from_numpy(np.ascontiguousarray(g['x'].transpose(0, 2, 1))).cuda()   # golden x is [T,3,P]; the kernel takes [T,P,3]
    w = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(g[k].shape[0], -1) if g[k].ndim > 1 else g[k])).cuda()
         for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')]
    y, am = ops.pointnet_forward(x, *w, want_argmax=True)
    err = np.abs(y.cpu().numpy() - g['y']).max()
    assert err < 1e-3 and err > 0, err                          # within the bar, and really the other arithmetic
    y0 = None
    ops.set_mfma_mode('f32')
    y0, _ = ops.pointnet_forward(x, *w, want_argmax=True)
    ops.set_mfma_mode('bf16x3')
    assert (y - y0).abs().max().item() < 2e-4 * max(1.0, y0.abs().max().item())


def test_example_pair_embeddings_loss_grads_and_hits(bf16x3):
    """BASELINE configs[0] (the reference's own example pair, PointNet-only encoder) in bf16x3 mode against the reference golden."""
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    from sgaligner_amd.utils import alignment
    g = load_golden('example_pair_point')
    ns, nr = [int(v) for v in g['counts']]
    model = MultiModalEncoder(modules=['point'], rel_dim=41, attr_dim=164).cuda()
    model.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd__')}, strict=True)
    dd = {'tot_obj_pts': torch.from_numpy(g['pts']).cuda(), 'batch_size': 1, 'e1i': g['e1i'], 'e2i': g['e2i'],
          'e1j': g['e1j'], 'e2j': g['e2j'], 'tot_obj_count': np.array([ns + nr]), 'e1i_count': np.array([len(g['e1i'])]),
          'graph_per_obj_count': np.array([[ns, nr]])}
    out = model(dd)
    res = OverallLoss(CustomMultiLossLayer(1), CustomMultiLossLayer(1), 'cuda',
                      {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': ['point']})(out, dd)
    res['loss'].backward()
    torch.cuda.synchronize()
    assert np.abs(out['point'].detach().cpu().numpy() - g['emb']).max() < 1e-3
    assert abs(res['loss'].item() - float(g['loss'])) < 1e-3
    for name, p in model.named_parameters():
        key = 'grad__' + name
        if key in g:
            assert np.abs(p.grad.cpu().numpy() - g[key]).max() < 1e-3 * max(1.0, np.abs(g[key]).max()), name
    m = alignment.evaluate_batch(out['point'].detach(), dd)
    assert [m[k]['correct'] for k in (1, 2, 3, 4, 5)] == [int(v) for v in g['hits']]
    assert np.allclose(m['mrr'], g['mrr'])


def test_train_step_vs_oracle(bf16x3):
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
    ddv = make_batch(3, 20, 96, seed=8, ragged=True, anchors='val')
    mo = O.evaluate_batch(out_o['joint'].detach(), ddv)
    mg = steps.eval_step(0, ddv, out)
    assert [mg[k]['correct'] for k in (1, 2, 3, 4, 5)] == [mo['hits'][k][0] for k in (1, 2, 3, 4, 5)]


@pytest.mark.parametrize('M,emb', [(3, 100), (2, 100), (3, 64)])
def test_sweeps_vs_fp32_sweeps_and_anchor_shards(bf16x3, M, emb):
    """The bf16x3 loss sweeps against the exact-fp32 sweeps on the same tables (loss terms, dE, d fusion weight), unsharded and as
    the sum of 3 anchor shards with cuts that are NOT multiples of the 32-row blocks (what ranks of a multi-GPU job own)."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    from test_c3_gpu import _replay_sharded
    dd = make_batch(9, 30, 4, seed=40 + M, ragged=True, anchors='val')
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(M)
    base = [torch.randn(T, emb, device='cuda', generator=g) for _ in range(M)]
    w0 = torch.tensor([[0.3], [1.1], [-0.4]], device='cuda')[:M].contiguous()
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
    ops.set_mfma_mode('bf16x3')
    assert torch.allclose(sb, sf, rtol=2e-5, atol=1e-6)
    for m in range(M):
        sc = gf[m].abs().max().item()
        assert (gb[m] - gf[m]).abs().max().item() < 1e-4 * sc, (m, (gb[m] - gf[m]).abs().max().item(), sc)
    assert (wb - wf).abs().max().item() < 1e-4 * max(1e-3, wf.abs().max().item())
    A = s.A
    cuts = [0, A // 3 + 5, 2 * A // 3 - 3, A]
    _, gs, gw, all_sums = _replay_sharded(base, w0, cot, dd, cuts)
    for sr in all_sums:
        assert torch.allclose(sr, sb, rtol=1e-4, atol=1e-5)
    for m in range(M):
        sc = gb[m].abs().max().item()
        assert (gs[m] - gb[m]).abs().max().item() < 2e-4 * sc, m
    assert (gw - wb).abs().max().item() < 2e-4 * max(1e-3, wb.abs().max().item())
