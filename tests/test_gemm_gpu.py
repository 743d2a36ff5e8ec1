"""The exact-fp32 MFMA GEMM entry points (sga_gemm / sga_gemm_ex) against torch fp64: every kernel variant (generic,
NT with register prefetch, row-major TN and NN with split-K) and the shapes that fall back to the generic one."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(c, ref, tol=2e-5):
    err = (c.double() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize('m,n,k', [(1000, 100, 256), (65, 33, 7), (128, 128, 32), (300, 104, 9728), (5000, 128, 3), (257, 260, 132), (128, 3, 40000), (70, 8, 333),
                                   (2000, 512, 1024)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False)])
def test_gemm_variants_vs_fp64(m, n, k, ta, tb):
    from sgaligner_amd import ops
    torch.manual_seed(m + n + k)
    a = torch.randn((k, m) if ta else (m, k), device='cuda')
    b = torch.randn((n, k) if tb else (k, n), device='cuda')
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    c = ops.gemm(a, b, ta, tb, m, n, k)
    _check(c, ref, 1e-5 * max(1.0, k ** 0.5))
    # accumulate on top of an existing C (split-K atomics must not clear it)
    c0 = torch.randn(m, n, device='cuda')
    c1 = ops.gemm(a, b, ta, tb, m, n, k, out=c0.clone(), accumulate=True)
    _check(c1, ref + c0.double(), 1e-5 * max(1.0, k ** 0.5))


def test_gemm_strided_views_and_bias():
    from sgaligner_amd import ops
    torch.manual_seed(0)
    big_a = torch.randn(700, 512, device='cuda')
    a = big_a[:, 128:256]                                   # lda = 512, 16-byte aligned column slice
    w = torch.randn(96, 128, device='cuda')
    bias = torch.randn(96, device='cuda')
    out_big = torch.zeros(700, 300, device='cuda')
    out = out_big[:, 100:196]
    ops.gemm(a, w, False, True, 700, 96, 128, bias=bias, out=out)
    _check(out, a.double() @ w.double().t() + bias.double())
    assert out_big[:, :100].abs().max() == 0 and out_big[:, 196:].abs().max() == 0


@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('m,n,k,with_resid', [(1000, 128, 128, True), (77, 1024, 512, False), (300, 64, 3, False)])
def test_gemm_ex_epilogue(act, m, n, k, with_resid):
    from sgaligner_amd.aligner.networks.pct import _gemm_ex
    torch.manual_seed(act * 7 + m)
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') * 0.3
    bias = torch.randn(n, device='cuda')
    resid = torch.randn(m, n, device='cuda') if with_resid else None
    y = _gemm_ex(a, w, bias, act=act, resid=resid)
    z = a.double() @ w.double().t() + bias.double()
    ref = z if act == 0 else (z.clamp_min(0) if act == 1 else torch.where(z > 0, z, 0.2 * z))
    if with_resid:
        ref = ref + resid.double()
    _check(y, ref, 2e-5)
