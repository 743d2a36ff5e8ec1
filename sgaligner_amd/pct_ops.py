"""Autograd wrappers of the PCT object-encoder kernels (csrc/bn.hip, csrc/pct.hip, csrc/gemm.hip): every forward and
backward below is a HIP launch through the C ABI; torch only owns the buffers and the tiny per-channel vectors."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _p, _stream


def _chk(rc, what):
    _lib.check(rc, what)


class BatchNormActFn(torch.autograd.Function):
    """y = act(BatchNorm1d(x)) (+ resid) over rows of x [R, C] (any row stride).  Train mode uses the batch statistics
    and updates running_mean / running_var / num_batches_tracked exactly like nn.BatchNorm1d (momentum 0.1, unbiased
    running variance); eval mode uses the running statistics.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, act, resid, out, pre_sums=None):
        R, C = x.shape
        dev = x.device
        L = _lib.lib()
        st = _stream()
        sums = None
        if training:
            if pre_sums is not None and pre_sums.numel() == 2 * C:       # from the producing GEMM's epilogue (rows_linear(bn_stats=True))
                sums = pre_sums
            else:
                sums = torch.empty((2 * C,), device=dev, dtype=torch.float64)
                _chk(L.sga_bn_stats(_p(x), x.stride(0), R, C, _p(sums), st), 'sga_bn_stats')
        # scale | shift | mean | rstd and the running-statistics update: one launch (csrc/bn.hip, bn_finalize_kernel)
        fin = torch.empty((4, C), device=dev, dtype=torch.float32)
        nbt = num_batches_tracked if (num_batches_tracked is not None and num_batches_tracked.dtype == torch.int64) else None
        g32 = gamma.detach() if gamma.dtype == torch.float32 else gamma.detach().float()
        b32 = beta.detach() if beta.dtype == torch.float32 else beta.detach().float()
        _chk(L.sga_bn_finalize(_p(sums), R, C, _p(g32.contiguous()), _p(b32.contiguous()), _p(running_mean), _p(running_var), _p(nbt),
                               float(momentum), float(eps), int(bool(training)), _p(fin), st), 'sga_bn_finalize')
        if training and num_batches_tracked is not None and nbt is None:
            num_batches_tracked.add_(1)
        scale, shift, mean, rstd = fin[0], fin[1], fin[2], fin[3]
        y = out if out is not None else torch.empty((R, C), device=dev, dtype=torch.float32)
        _chk(L.sga_bn_apply(_p(x), x.stride(0), R, C, _p(scale), _p(shift), act, _p(resid), resid.stride(0) if resid is not None else 0,
                            _p(y), y.stride(0), st), 'sga_bn_apply')
        ctx.save_for_backward(x, fin)
        ctx.act, ctx.training, ctx.has_resid = act, training, resid is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, fin = ctx.saved_tensors
        scale, shift, mean, rstd = fin[0], fin[1], fin[2], fin[3]
        R, C = x.shape
        L = _lib.lib()
        st = _stream()
        dyc = dy if dy.stride(1) == 1 else dy.contiguous()
        sums = torch.empty((2 * C,), device=x.device, dtype=torch.float64)
        _chk(L.sga_bn_bwd_stats(_p(x), x.stride(0), _p(dyc), dyc.stride(0), R, C, _p(scale), _p(shift), _p(mean), _p(rstd), ctx.act,
                                _p(sums), st), 'sga_bn_bwd_stats')
        bf = torch.empty((4, C), device=x.device, dtype=torch.float32)          # dbeta | dgamma | mean_g | mean_gx: one launch
        _chk(L.sga_bn_bwd_finalize(_p(sums), R, C, _p(bf), st), 'sga_bn_bwd_finalize')
        dbeta, dgamma = bf[0], bf[1]
        mg, mgx = (bf[2], bf[3]) if ctx.training else (None, None)
        dx = torch.empty((R, C), device=x.device, dtype=torch.float32)
        _chk(L.sga_bn_bwd_apply(_p(x), x.stride(0), _p(dyc), dyc.stride(0), R, C, _p(scale), _p(shift), _p(mean), _p(rstd), _p(mg), _p(mgx),
                                ctx.act, _p(dx), dx.stride(0), st), 'sga_bn_bwd_apply')
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, (dy if ctx.has_resid else None), None, None


def batch_norm_act(x, bn: torch.nn.BatchNorm1d, act=0, resid=None, out=None):
    """nn.BatchNorm1d `bn` applied to rows of x, then the activation, then `+ resid`."""
    mom = 0.1 if bn.momentum is None else bn.momentum
    return BatchNormActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training, mom,
                                bn.eps, act, resid, out, getattr(x, '_sga_bn_sums', None) if bn.training else None)


class PCTAttentionFn(torch.autograd.Function):
    """xs[j] = sum_i softmax_row_i(q q^T / sqrt(32))[i, j] v[i] per object (pct.py:211-222 with the shared q/k weight);
    q [T*N, 32], v [T*N, 128] point-major (column slices allowed), every object has n_pts points."""

    @staticmethod
    def forward(ctx, q, v, n_obj, n_pts):
        dev = q.device
        stats = torch.empty((2 * n_obj * n_pts,), device=dev, dtype=torch.float32)
        xs = torch.empty((n_obj * n_pts, 128), device=dev, dtype=torch.float32)
        _chk(_lib.lib().sga_pct_attention(_p(q), q.stride(0), _p(v), v.stride(0), n_obj, n_pts, _p(stats), _p(xs), xs.stride(0), _stream()),
             'sga_pct_attention')
        ctx.save_for_backward(q, v, stats)
        ctx.dims = (n_obj, n_pts)
        return xs

    @staticmethod
    def backward(ctx, dxs):
        q, v, stats = ctx.saved_tensors
        n_obj, n_pts = ctx.dims
        dxs = dxs.contiguous()
        dev = q.device
        work = torch.empty((n_obj * n_pts,), device=dev, dtype=torch.float32)
        dq = torch.empty((n_obj * n_pts, 32), device=dev, dtype=torch.float32)
        dv = torch.empty((n_obj * n_pts, 128), device=dev, dtype=torch.float32)
        _chk(_lib.lib().sga_pct_attention_bwd(_p(q), q.stride(0), _p(v), v.stride(0), _p(dxs), dxs.stride(0), n_obj, n_pts, _p(stats),
                                             _p(work), _p(dq), dq.stride(0), _p(dv), dv.stride(0), _stream()), 'sga_pct_attention_bwd')
        return dq, dv, None, None


class PCTAttentionQVFn(torch.autograd.Function):
    """PCTAttentionFn on ONE [T*N, 160] tensor holding q (columns 0..31) and v (32..159) side by side -- the output of a single
    N = 160 GEMM over x (q_conv and v_conv read the same rows): x is read once instead of twice, the two dX GEMMs and the add of
    their results become one, and the gradient leaves as one [T*N, 160] tensor (no concatenation)."""

    @staticmethod
    def forward(ctx, qv, n_obj, n_pts):
        dev = qv.device
        qv = qv if qv.is_contiguous() else qv.contiguous()
        q, v = qv[:, :32], qv[:, 32:]
        stats = torch.empty((2 * n_obj * n_pts,), device=dev, dtype=torch.float32)
        xs = torch.empty((n_obj * n_pts, 128), device=dev, dtype=torch.float32)
        _chk(_lib.lib().sga_pct_attention(_p(q), q.stride(0), _p(v), v.stride(0), n_obj, n_pts, _p(stats), _p(xs), xs.stride(0), _stream()),
             'sga_pct_attention')
        ctx.save_for_backward(qv, stats)
        ctx.dims = (n_obj, n_pts)
        return xs

    @staticmethod
    def backward(ctx, dxs):
        qv, stats = ctx.saved_tensors
        n_obj, n_pts = ctx.dims
        q, v = qv[:, :32], qv[:, 32:]
        dxs = dxs.contiguous()
        dev = qv.device
        work = torch.empty((n_obj * n_pts,), device=dev, dtype=torch.float32)
        dqv = torch.empty_like(qv)
        dq, dv = dqv[:, :32], dqv[:, 32:]
        _chk(_lib.lib().sga_pct_attention_bwd(_p(q), q.stride(0), _p(v), v.stride(0), _p(dxs), dxs.stride(0), n_obj, n_pts, _p(stats),
                                             _p(work), _p(dq), dq.stride(0), _p(dv), dv.stride(0), _stream()), 'sga_pct_attention_bwd')
        return dqv, None, None


def pct_attention_qv(qv, n_obj, n_pts):
    return PCTAttentionQVFn.apply(qv, n_obj, n_pts)


def pct_attention(q, v, n_obj, n_pts):
    return PCTAttentionFn.apply(q, v, n_obj, n_pts)


class SegmentMaxFn(torch.autograd.Function):
    """g[t, c] = max over the n_pts rows of object t (pct.py:308); backward routes to the first arg-max row."""

    @staticmethod
    def forward(ctx, y, n_obj, n_pts):
        C = y.shape[1]
        g = torch.empty((n_obj, C), device=y.device, dtype=torch.float32)
        am = torch.empty((n_obj, C), device=y.device, dtype=torch.int32)
        _chk(_lib.lib().sga_segment_max(_p(y), y.stride(0), n_obj, n_pts, C, _p(g), _p(am), _stream()), 'sga_segment_max')
        ctx.save_for_backward(am)
        ctx.dims = (n_obj, n_pts, C)
        return g

    @staticmethod
    def backward(ctx, dg):
        (am,) = ctx.saved_tensors
        n_obj, n_pts, C = ctx.dims
        dg = dg.contiguous()
        dy = torch.empty((n_obj * n_pts, C), device=dg.device, dtype=torch.float32)
        _chk(_lib.lib().sga_segment_max_bwd(_p(dg), _p(am), n_obj, n_pts, C, _p(dy), dy.stride(0), _stream()), 'sga_segment_max_bwd')
        return dy, None, None


def segment_max(y, n_obj, n_pts):
    return SegmentMaxFn.apply(y, n_obj, n_pts)


class RowsLinearFn(torch.autograd.Function):
    """y = x W^T (+ b) over point-major rows: Conv1d(k=1) / nn.Linear of the PCT encoder.  W may be [out, in] or the
    Conv1d layout [out, in, 1]; the bias is optional (pct.py uses bias=False in front of every BatchNorm)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stats_holder=None):
        from . import ops
        x = x if x.is_contiguous() else x.contiguous()
        w = weight.reshape(weight.shape[0], -1)
        r, k = x.shape
        n = w.shape[0]
        y = None
        if stats_holder is not None and k % 4 == 0 and r > 0 and w.is_contiguous():
            # the BatchNorm that follows needs sum / sum of squares of y: produced by the GEMM's own epilogue (no second pass over y)
            y = torch.empty((r, n), device=x.device, dtype=torch.float32)
            sums = torch.empty((2 * n,), device=x.device, dtype=torch.float64)
            rc = _lib.lib().sga_gemm_bnstats(r, n, k, _p(x), x.stride(0), _p(w), w.stride(0), _p(y), y.stride(0), _p(bias), _p(sums), _stream())
            if rc == 0:
                stats_holder.append(sums)
            else:
                y = None
        if y is None:
            y = ops.gemm(x, w, False, True, r, n, k, bias=bias)
        ctx.save_for_backward(x, w)
        ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x, w = ctx.saved_tensors
        gy = gy if gy.is_contiguous() else gy.contiguous()
        r, k = x.shape
        n = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = w.t().contiguous()                                           # [k, n]: dX = dY (W^T)^T runs on the NT kernel
            gx = ops.gemm(gy, wt, False, True, r, k, n)
        if ctx.needs_input_grad[1]:
            gw = ops.gemm(gy, x, True, False, n, k, r).reshape(ctx.wshape)    # dW = dY^T X (split-K over the rows)
        if ctx.needs_input_grad[2]:
            gb = ops.colsum(gy)
        return gx, gw, gb, None


def rows_linear(x, weight, bias=None, bn_stats=False):
    """bn_stats=True: the output carries the BatchNorm batch statistics of itself (`_sga_bn_sums`, from the GEMM's epilogue) for the
    `batch_norm_act` call that consumes it."""
    if not bn_stats:
        return RowsLinearFn.apply(x, weight, bias, None)
    holder = []
    y = RowsLinearFn.apply(x, weight, bias, holder)
    if holder:
        y._sga_bn_sums = holder[0]
    return y


class LinearBNActMaxFn(torch.autograd.Function):
    """g[t, :] = max over object t's points of LeakyReLU_0.2(BatchNorm1d(cat W^T))  -- the encoder's widest stage (pct.py:282-286 + :308:
    Conv1d(512 -> 1024, bias=False), BatchNorm1d(1024), LeakyReLU, torch.max over points) as ONE autograd node.
    Forward: GEMM, batch statistics, then BatchNorm-apply + LeakyReLU folded into the arg-max pool (y is read twice and never rewritten);
    nothing of size [T*N, 1024] is kept for the backward.  Backward: only the arg-max rows carry dL/dz, and the batch-statistic terms of the BatchNorm
    backward are affine in y = cat W^T, so dW and dcat follow from a 512 x 512 Gram matrix and one [T*N, 512] x [512, 512] product plus
    two sparse passes (csrc/pct.hip head_*_kernel): half the GEMM FLOPs of dY W / dY^T cat, none of the [T*N, 1024] gradient traffic."""

    SLOPE = 0.2

    @staticmethod
    def forward(ctx, cat, weight, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, n_obj, n_pts):
        from . import ops
        L = _lib.lib()
        st = _stream()
        cat = cat if cat.is_contiguous() else cat.contiguous()
        w = weight.reshape(weight.shape[0], -1).contiguous()
        R, K = cat.shape
        C = w.shape[0]
        dev = cat.device
        y = sums = None
        if training and K % 4 == 0 and R > 0:
            y = torch.empty((R, C), device=dev, dtype=torch.float32)
            sums = torch.empty((2 * C,), device=dev, dtype=torch.float64)
            if L.sga_gemm_bnstats(R, C, K, _p(cat), cat.stride(0), _p(w), w.stride(0), _p(y), y.stride(0), None, _p(sums), st) != 0:
                y = sums = None
        if y is None:
            y = ops.gemm(cat, w, False, True, R, C, K)
            if training:
                sums = torch.empty((2 * C,), device=dev, dtype=torch.float64)
                _chk(L.sga_bn_stats(_p(y), y.stride(0), R, C, _p(sums), st), 'sga_bn_stats')
        fin = torch.empty((4, C), device=dev, dtype=torch.float32)
        nbt = num_batches_tracked if (num_batches_tracked is not None and num_batches_tracked.dtype == torch.int64) else None
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _chk(L.sga_bn_finalize(_p(sums), R, C, _p(g32), _p(b32), _p(running_mean), _p(running_var), _p(nbt), float(momentum), float(eps),
                               int(bool(training)), _p(fin), st), 'sga_bn_finalize')
        if training and num_batches_tracked is not None and nbt is None:
            num_batches_tracked.add_(1)
        g = torch.empty((n_obj, C), device=dev, dtype=torch.float32)
        am = torch.empty((n_obj, C), device=dev, dtype=torch.int32)
        # BatchNorm-apply + LeakyReLU folded into the point max: y is read once and never rewritten
        _chk(L.sga_segment_max_affine(_p(y), y.stride(0), n_obj, n_pts, C, _p(fin[0]), _p(fin[1]), LinearBNActMaxFn.SLOPE, _p(g), _p(am), st),
             'sga_segment_max_affine')
        del y
        ctx.save_for_backward(cat, w, fin, g, am, g32, b32)
        ctx.meta = (n_obj, n_pts, bool(training), tuple(weight.shape))
        return g

    @staticmethod
    def backward(ctx, dg):
        from . import ops
        cat, w, fin, g, am, gamma, beta = ctx.saved_tensors
        n_obj, n_pts, training, wshape = ctx.meta
        L = _lib.lib()
        st = _stream()
        R, K = cat.shape
        C = w.shape[0]
        dev = cat.device
        dg = dg.contiguous().float()
        coef = torch.empty((n_obj, C), device=dev, dtype=torch.float32)
        ab = torch.empty((4, C), device=dev, dtype=torch.float32)                  # a | b | dgamma | dbeta
        _chk(L.sga_pct_head_prep(_p(dg), _p(g), _p(gamma), _p(beta), _p(fin), n_obj, C, R, int(training), LinearBNActMaxFn.SLOPE, _p(coef),
                                 _p(ab), st), 'sga_pct_head_prep')
        gram = ops.gemm(cat, cat, True, False, K, K, R)                            # cat^T cat  [K, K]  (split over the rows)
        cs = ops.colsum(cat)                                                       # [K]
        wg = ops.gemm(w, gram, False, True, C, K, K)                               # W (cat^T cat)  [C, K]  (the Gram matrix is symmetric: NT kernel)
        dw = torch.empty((C, K), device=dev, dtype=torch.float32)
        wb = torch.empty((C, K), device=dev, dtype=torch.float32)
        a0 = torch.zeros((K,), device=dev, dtype=torch.float32)                    # accumulated atomically by head_dw_kernel
        _chk(L.sga_pct_head_dw(_p(wg), _p(w), _p(ab), _p(cs), _p(coef), _p(am), _p(cat), cat.stride(0), n_obj, n_pts, C, K, _p(dw), _p(wb),
                               _p(a0), st), 'sga_pct_head_dw')
        dcat = None
        if ctx.needs_input_grad[0]:
            m = ops.gemm(w, wb, True, False, K, K, C)                              # W^T diag(b) W  [K, K]
            dcat = ops.gemm(cat, m, False, True, R, K, K, bias=a0)                 # cat M + 1 (x) a^T W  (M is symmetric: NT kernel)
            _chk(L.sga_pct_head_scatter(_p(coef), _p(am), _p(w), n_obj, n_pts, C, K, _p(dcat), dcat.stride(0), st), 'sga_pct_head_scatter')
        return dcat, dw.reshape(wshape), ab[2].clone(), ab[3].clone(), None, None, None, None, None, None, None, None


def linear_bn_lrelu_max(cat, weight, bn: torch.nn.BatchNorm1d, n_obj, n_pts):
    mom = 0.1 if bn.momentum is None else bn.momentum
    return LinearBNActMaxFn.apply(cat, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training, mom,
                                  bn.eps, n_obj, n_pts)
