"""Dataset -> collate -> DeviceBatch -> hot path on the GPU, against the oracle fed with the same collated batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3


def test_dataset_batch_through_the_hot_path(tmp_path):
    from oracle import sga_oracle as O
    from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
    from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
    from sgaligner_amd.datasets import DeviceBatch, Scan3RDataset, synthetic_scan3r as S
    from sgaligner_amd.utils import alignment
    root = str(tmp_path)
    S.write_dataset(root, n_pairs=5, seed=9, resolutions=(64,))
    mods = ['point', 'gat', 'rel', 'attr']
    ds = Scan3RDataset(S.make_cfg(root, pc_res=64), 'train')
    np.random.seed(4)
    dd = ds.collate_fn([ds[i] for i in range(len(ds))])
    torch.manual_seed(2)
    model = MultiModalEncoder(modules=mods, rel_dim=41, attr_dim=164)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if 'num_batches' not in k}
    out_o, loss_o, grads_o = O.train_step(params, dd, mods)
    model = model.cuda()
    ddd = DeviceBatch(dd)
    assert ddd['tot_obj_pts'].is_cuda and isinstance(ddd['e1i'], np.ndarray)
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
            assert (p.grad.cpu() - ref).abs().max().item() < TOL * max(1.0, ref.abs().max().item()), name
    # validation split of the same files: Hits@K / MRR of the HIP ranking kernel == oracle
    dv = Scan3RDataset(S.make_cfg(root, pc_res=64), 'val')
    ddv = dv.collate_fn([dv[i] for i in range(len(dv))])
    with torch.no_grad():
        ev = model(DeviceBatch(ddv))['joint']
    p2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if 'num_batches' not in k}
    ev_o = O.encoder_forward(p2, ddv, mods)['joint']
    mo = O.evaluate_batch(ev_o.detach(), ddv)
    mg = alignment.evaluate_batch(ev, ddv)
    assert [mg[k]['correct'] for k in (1, 2, 3, 4, 5)] == [mo['hits'][k][0] for k in (1, 2, 3, 4, 5)]
    assert np.allclose(mg['mrr'], mo['mrr'])


def test_device_prefetcher_matches_device_batch():
    """DevicePrefetcher (upload of batch i+1 on a second stream under the step of batch i) hands out the same batches, in
    order, as DeviceBatch; the step results are identical; `prepare` sees the host batch."""
    import torch
    from sgaligner_amd.datasets import DeviceBatch, DevicePrefetcher
    from sgaligner_amd.synthetic import make_batch
    from sgaligner_amd.trainer import AlignerSteps
    host = [make_batch(3, 10, 32, seed=70 + i) for i in range(4)]
    steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
    want = []
    for dd in host:
        _, l = steps.forward_backward(DeviceBatch(dd))
        want.append(float(l['loss'].item()))
    seen = []
    got = []
    for i, dd in enumerate(DevicePrefetcher(host, 'cuda', prepare=lambda d: (seen.append(id(d)), d)[1])):
        assert all(v.is_cuda for v in dd.values() if isinstance(v, torch.Tensor))
        assert torch.equal(dd['tot_obj_pts'].cpu(), host[i]['tot_obj_pts'])
        _, l = steps.forward_backward(dd)
        got.append(float(l['loss'].item()))
    assert seen == [id(d) for d in host]
    assert got == want
    assert len(DevicePrefetcher(host, 'cuda')) == 4
