"""GPU parity: HIP PointNet (through the C-ABI) vs the CPU oracle and the reference golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-3        # north_star tolerance (fp32 absolute); observed error is ~1e-6


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('tag', ['small', 'ragged'])
def test_pointnet_fwd_golden(tag):
    from sgaligner_amd import ops
    g = load_golden('pointnet_' + tag)
    x = _dev(g['x'].transpose(0, 2, 1))           # golden x is [T,3,P]; the kernel takes [T,P,3]
    w = [_dev(g[k].reshape(g[k].shape[0], -1)) for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')]
    y, am = ops.pointnet_forward(x, *w, want_argmax=True)
    torch.cuda.synchronize()
    err = np.abs(y.cpu().numpy() - g['y']).max()
    assert err < TOL, err
    assert err < 2e-5, err
    y2, _ = ops.pointnet_forward(x, *w, want_argmax=False)
    assert torch.equal(y, y2)


@pytest.mark.parametrize('T,P', [(1, 1), (3, 31), (17, 32), (40, 512), (9, 100), (2100, 64)])
def test_pointnet_fwd_oracle(T, P):
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    torch.manual_seed(T * 1000 + P)
    p = O.init_params(['point'])
    ws = [p['object_encoder.conv1.weight'].reshape(64, 3), torch.randn(64) * 0.1,
          p['object_encoder.conv2.weight'].reshape(128, 64), torch.randn(128) * 0.1,
          p['object_encoder.conv3.weight'].reshape(256, 128), torch.randn(256) * 0.1]
    x = torch.randn(T, P, 3)
    yo, io = O.pointnet_feat(x.permute(0, 2, 1), *ws, return_argmax=True)
    y, am = ops.pointnet_forward(x.cuda(), *[w.contiguous().cuda() for w in ws], want_argmax=True)
    torch.cuda.synchronize()
    assert (y.cpu() - yo).abs().max() < 2e-5
    # argmax parity wherever the max is positive and the top-2 gap is not a rounding tie
    am = am.cpu().long()
    assert am.min() >= 0 and am.max() < P
    agree = (am == io) | (yo <= 0)
    assert agree.float().mean() > 0.999


@pytest.mark.parametrize('mode', ['bf16x6', 'f32'])
@pytest.mark.parametrize('C3,T,P', [(64, 1100, 40), (128, 1100, 40), (128, 30, 70), (256, 1500, 33), (64, 9, 512)])
def test_pointnet_fwd_out_sizes_and_modes_vs_oracle(mode, C3, T, P):
    """Every out_size the kernels are built for (64, 128: all of W3's three planes in LDS; 256: channel halves over workgroup pairs), both
    launch forms, in the default arithmetic (three exact bf16 planes) and on the fp32 MFMA: values within 2e-5 of the fp64 oracle, arg-max
    parity off rounding ties, and the BatchNorm sums of the fused forward equal to the oracle's batch statistics."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    torch.manual_seed(C3 + T + P)
    ws = [torch.randn(64, 3) * 0.3, torch.randn(64) * 0.1, torch.randn(128, 64) * 0.15, torch.randn(128) * 0.1,
          torch.randn(C3, 128) * 0.1, torch.randn(C3) * 0.1]
    x = torch.randn(T, P, 3) + torch.tensor([0.5, -1.0, 0.25])
    yo, io = O.pointnet_feat(x.double().permute(0, 2, 1), *[w.double() for w in ws], return_argmax=True)
    ref = O.pointnet_bn_batch_stats(x.double().permute(0, 2, 1), *[w.double() for w in ws])
    wd = [w.contiguous().cuda() for w in ws]
    old = ops.set_mfma_mode(mode)
    try:
        y, am = ops.pointnet_forward(x.cuda(), *wd, want_argmax=True)
        sums = torch.empty(265 + 2 * C3, device='cuda', dtype=torch.float64)
        y2, am2 = ops.pointnet_forward(x.cuda(), *wd, want_argmax=True, bn_sums=sums)
        torch.cuda.synchronize()
    finally:
        ops.set_mfma_mode(old)
    assert torch.equal(y, y2) and torch.equal(am, am2)
    assert (y.cpu().double() - yo).abs().max() < 2e-5
    agree = (am.cpu().long() == io) | (yo <= 0)
    assert agree.float().mean() > 0.999
    n = T * P
    s = sums.cpu()
    mean2, var2 = s[9:137] / n, (s[137:265] / n - (s[9:137] / n) ** 2) * n / (n - 1)
    u = s[265:265 + C3] / n
    mean3, var3 = u + ws[5].double(), (s[265 + C3:] / n - u * u) * n / (n - 1)
    for got, want in ((mean2, ref[1][0]), (var2, ref[1][1]), (mean3, ref[2][0]), (var3, ref[2][1])):
        assert (got - want).abs().max() < 2e-6 * max(1.0, want.abs().max()), (mode, C3)
    m = s[0:3] / n
    assert (m - x.double().reshape(-1, 3).mean(0)).abs().max() < 1e-9


@pytest.mark.parametrize('tag', ['small', 'ragged'])
def test_pointnet_bwd_golden(tag):
    from sgaligner_amd import ops
    g = load_golden('pointnet_' + tag)
    x = _dev(g['x'].transpose(0, 2, 1))
    w = [_dev(g[k]).requires_grad_(True) for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')]
    y = ops.pointnet(x, *w)
    (y * _dev(g['cot'])).sum().backward()
    torch.cuda.synchronize()
    assert np.abs(y.detach().cpu().numpy() - g['y']).max() < 2e-5
    for t, k in zip(w, ('gw1', 'gb1', 'gw2', 'gb2', 'gw3', 'gb3')):
        ref = g[k]
        err = np.abs(t.grad.cpu().numpy() - ref).max()
        assert err < 1e-4 * max(1.0, np.abs(ref).max()), (k, err, np.abs(ref).max())


@pytest.mark.parametrize('T,P', [(1, 5), (300, 64), (40, 512), (1000, 33)])
def test_pointnet_bwd_oracle(T, P):
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    torch.manual_seed(T + P)
    p = O.init_params(['point'])
    ws = [p['object_encoder.conv1.weight'].reshape(64, 3), torch.randn(64) * 0.1,
          p['object_encoder.conv2.weight'].reshape(128, 64), torch.randn(128) * 0.1,
          p['object_encoder.conv3.weight'].reshape(256, 128), torch.randn(256) * 0.1]
    x = torch.randn(T, P, 3)
    cot = torch.randn(T, 256)
    wr = [w.clone().double().requires_grad_(True) for w in ws]
    yo = O.pointnet_feat(x.double().permute(0, 2, 1), *wr)
    (yo * cot.double()).sum().backward()
    wd = [w.clone().contiguous().cuda().requires_grad_(True) for w in ws]
    y = ops.pointnet(x.cuda(), *wd)
    (y * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    for a, b, name in zip(wd, wr, ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')):
        ref = b.grad
        err = (a.grad.cpu().double() - ref).abs().max().item()
        assert err < 2e-4 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())


@pytest.mark.parametrize('tag', ['small', 'ragged'])
def test_pointnet_bn_running_stats_side_effect(tag):
    """The reference's discarded BatchNorm calls update their running statistics in train mode (pointnet.py:141-142,
    154-155,158-159).  The HIP path reproduces the buffers of the reference run BY DEFAULT, from sums taken inside the forward kernel
    (sga_pointnet_fwd_bn); the separate chunked pass (what the opt-in arithmetic modes use) gives the same; the switch turns it off."""
    from sgaligner_amd.aligner.networks.pointnet import PointNetfeat
    g = load_golden('pointnet_' + tag)

    def fresh():
        net = PointNetfeat(global_feat=True, batch_norm=True, point_size=3, input_transform=False, feature_transform=False, out_size=256).cuda()
        with torch.no_grad():
            for conv, w, b in ((net.conv1, 'w1', 'b1'), (net.conv2, 'w2', 'b2'), (net.conv3, 'w3', 'b3')):
                conv.weight.copy_(_dev(g[w])); conv.bias.copy_(_dev(g[b]))
        return net.train()

    def check(net):
        for bn, rm, rv in ((net.bn1, 'rm1', 'rv1'), (net.bn2, 'rm2', 'rv2'), (net.bn3, 'rm3', 'rv3')):
            assert np.abs(bn.running_mean.cpu().numpy() - g[rm]).max() < 1e-5 * max(1.0, np.abs(g[rm]).max()), rm
            assert np.abs(bn.running_var.cpu().numpy() - g[rv]).max() < 1e-5 * max(1.0, np.abs(g[rv]).max()), rv
            assert int(bn.num_batches_tracked) == 1

    x = _dev(g['x'])                                   # [T,3,P] as the reference passes it
    net = fresh()
    assert net.update_bn_running_stats                 # the default
    y = net(x)
    torch.cuda.synchronize()
    assert np.abs(y.detach().cpu().numpy() - g['y']).max() < 2e-5
    check(net)
    net.eval()
    net(x)
    assert int(net.bn1.num_batches_tracked) == 1       # eval: no update
    # the separate pass
    net2 = fresh()
    with torch.no_grad():
        net2._update_bn_running_stats(x.permute(0, 2, 1).contiguous())
    check(net2)
    # switched off: buffers untouched
    net3 = fresh()
    net3.update_bn_running_stats = False
    net3(x)
    assert float(net3.bn1.running_mean.abs().max()) == 0.0 and int(net3.bn1.num_batches_tracked) == 0


@pytest.mark.parametrize('T,P', [(3, 5), (37, 512), (1100, 40), (2500, 64)])
def test_pointnet_fused_bn_sums_vs_oracle(T, P):
    """sga_pointnet_fwd_bn in both launch forms (objects split over a workgroup's waves below 4 x CUs objects, one wave per object above),
    whole and ragged 32-point tiles, with and without arg-max: the forward's outputs are bit-identical to the plain forward's, and
    the statistics it delivers give the oracle's batch means / unbiased variances (fp64 evaluation of pointnet.py:141-159)."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    from sgaligner_amd.aligner.networks.pointnet import PointNetfeat
    torch.manual_seed(T * 7 + P)
    net = PointNetfeat(global_feat=True, batch_norm=True, point_size=3, input_transform=False, feature_transform=False, out_size=256)
    with torch.no_grad():
        for c in (net.conv1, net.conv2, net.conv3):
            c.bias.normal_(0, 0.2)
    x = torch.randn(T, 3, P) * 0.7 + torch.tensor([1.5, -0.5, 0.25])[None, :, None]     # off-centre points: the moment route must not cancel
    ws = [net.conv1.weight.detach().reshape(64, 3), net.conv1.bias.detach(), net.conv2.weight.detach().reshape(128, 64), net.conv2.bias.detach(),
          net.conv3.weight.detach().reshape(256, 128), net.conv3.bias.detach()]
    ref = O.pointnet_bn_batch_stats(x.double(), *[w.double() for w in ws])
    net = net.cuda().train()
    xd = x.cuda()
    wd = [w.cuda().contiguous() for w in ws]
    y_plain, am_plain = ops.pointnet_forward(xd.permute(0, 2, 1).contiguous(), *wd, want_argmax=True)
    for want_am in (True, False):
        for bn in (net.bn1, net.bn2, net.bn3):
            bn.reset_running_stats()
        if want_am:
            y = net(xd)
        else:
            with torch.no_grad():
                y = net(xd)
        torch.cuda.synchronize()
        assert torch.equal(y.detach(), y_plain)
        for bn, (mean, var) in zip((net.bn1, net.bn2, net.bn3), ref):
            rm = 0.1 * mean
            rv = 0.9 + 0.1 * var
            assert (bn.running_mean.cpu().double() - rm).abs().max() < 2e-6 * max(1.0, rm.abs().max()), (want_am, bn.num_features)
            assert (bn.running_var.cpu().double() - rv).abs().max() < 2e-6 * max(1.0, rv.abs().max()), (want_am, bn.num_features)
            assert int(bn.num_batches_tracked) == 1
