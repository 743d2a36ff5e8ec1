"""GPU parity: generic MFMA GEMM / Linear and the fusion kernels vs torch-CPU oracle + golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('ta,tb,m,n,k', [(0, 1, 300, 100, 256), (0, 0, 300, 256, 100), (1, 0, 100, 256, 9000),
                                         (0, 1, 77, 100, 41), (1, 0, 100, 41, 5000), (0, 1, 1, 1, 1),
                                         (0, 0, 129, 130, 33), (1, 1, 65, 200, 70), (0, 1, 500, 304, 304)])
def test_gemm_variants(ta, tb, m, n, k):
    from sgaligner_amd import ops
    torch.manual_seed(m + n + k)
    a = torch.randn((k, m) if ta else (m, k), dtype=torch.float64)
    b = torch.randn((n, k) if tb else (k, n), dtype=torch.float64)
    bias = torch.randn(n, dtype=torch.float64)
    ref = (a.t() if ta else a) @ (b.t() if tb else b) + bias
    out = ops.gemm(a.float().cuda(), b.float().cuda(), bool(ta), bool(tb), m, n, k, bias=bias.float().cuda())
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, k ** 0.5), err
    out2 = ops.gemm(a.float().cuda(), b.float().cuda(), bool(ta), bool(tb), m, n, k, out=out.clone(), accumulate=True)
    torch.cuda.synchronize()
    err2 = (out2.cpu().double() - (2 * ref - bias)).abs().max().item()
    assert err2 < 2e-4 * max(1.0, k ** 0.5), err2


@pytest.mark.parametrize('t,k,f64', [(333, 256, False), (1000, 41, True), (257, 164, True)])
def test_linear_fwd_bwd(t, k, f64):
    from sgaligner_amd import ops
    torch.manual_seed(t)
    x = torch.randn(t, k, dtype=torch.float64 if f64 else torch.float32)
    w = (torch.randn(100, k) * 0.1).requires_grad_(True)
    b = (torch.randn(100) * 0.1).requires_grad_(True)
    cot = torch.randn(t, 100)
    xr = x.float().requires_grad_(not f64)
    y_ref = torch.nn.functional.linear(xr, w, b)
    (y_ref * cot).sum().backward()
    xd = x.cuda().detach().requires_grad_(not f64)
    wd = w.detach().cuda().requires_grad_(True)
    bd = b.detach().cuda().requires_grad_(True)
    y = ops.linear(xd, wd, bd)
    (y * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert (y.detach().cpu() - y_ref.detach()).abs().max() < 1e-4
    assert (wd.grad.cpu() - w.grad).abs().max() < 1e-3 * max(1.0, w.grad.abs().max().item())
    assert (bd.grad.cpu() - b.grad).abs().max() < 1e-3 * max(1.0, b.grad.abs().max().item())
    if not f64:
        assert (xd.grad.cpu() - xr.grad).abs().max() < 1e-4


@pytest.mark.parametrize('m', [2, 3, 4])
def test_fusion_golden(m):
    from sgaligner_amd import ops
    g = load_golden(f'fusion_m{m}')
    w = torch.from_numpy(g['weight']).cuda().requires_grad_(True)
    embs = [torch.from_numpy(g[f'emb{i}']).cuda().requires_grad_(True) for i in range(m)]
    j = ops.fusion(w, embs)
    (j * torch.from_numpy(g['cot']).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert np.abs(j.detach().cpu().numpy() - g['joint']).max() < 1e-5
    assert np.abs(w.grad.cpu().numpy() - g['gweight']).max() < 1e-4
    for i, e in enumerate(embs):
        assert np.abs(e.grad.cpu().numpy() - g[f'gemb{i}']).max() < 1e-4


def test_fusion_zero_row_and_large():
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    torch.manual_seed(0)
    embs = [torch.randn(5000, 100) for _ in range(3)]
    embs[1][17] = 0.0                                    # F.normalize eps branch
    w = torch.tensor([[0.3], [1.2], [-0.4]])
    er = [e.clone().requires_grad_(True) for e in embs]
    wr = w.clone().requires_grad_(True)
    cot = torch.randn(5000, 300)
    (O.fusion(er, wr) * cot).sum().backward()
    ed = [e.cuda().requires_grad_(True) for e in embs]
    wd = w.cuda().requires_grad_(True)
    j = ops.fusion(wd, ed)
    (j * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert (j.detach().cpu() - O.fusion(embs, w)).abs().max() < 1e-5
    assert (wd.grad.cpu() - wr.grad).abs().max() < 1e-3 * wr.grad.abs().max().clamp_min(1)
    for a, b in zip(ed, er):
        # the all-zero row takes F.normalize's eps branch: gradient = w * g / 1e-12 (huge) -> relative check
        err = (a.grad.cpu() - b.grad).abs() / b.grad.abs().clamp_min(1.0)
        assert err.max() < 1e-4
