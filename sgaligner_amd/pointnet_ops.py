"""PointNetfeat forward / backward (reference src/aligner/networks/pointnet.py:120-175) on csrc/pointnet.hip.

Part of the autograd layer over the C-ABI HIP kernels (see ops.py, which re-exports everything here: `sgaligner_amd.ops.<name>` keeps
working).  The run-time switches live in ops.py and are read through the module at call time (`_o.FLAG`), so `ops.FLAG = value` set by a
caller or a test takes effect here."""
from __future__ import annotations

import ctypes as _ct

import numpy as _np
import torch

from . import _lib
from . import ops as _o
from .ops import (_SmallCache, _ev_start, _ev_stop, _fingerprint, _h2d, _p, _ptr_array, _req, _stream, get_mfma_mode, DEFERRED_CHECKS, IndexSets,
                  _POINTNET_MODE, _stash_bytes, cast_f32, colsum, gemm)

# ------------------------------------------------------------------------------------------ PointNet
def pointnet_bn_fusable() -> bool:
    """Both forward kernels (exact fp32, three exact bf16 planes) deliver the BatchNorm batch statistics of the reference's training forward
    from inside the kernel (sga_pointnet_fwd_bn)."""
    return True


def pointnet_forward(x_tp3, w1, b1, w2, b2, w3, b3, want_argmax: bool, bn_sums=None):
    """x_tp3 [T,P,3] (point-major, as in data_dict['tot_obj_pts']).  Returns (y [T,C3], argmax|None).
    bn_sums: a float64 tensor of 265 + 2 C3 elements to receive the batch-statistic sums of the three pre-activations (layout:
    include/sgaligner_hip.h, sga_pointnet_fwd_bn) -- exact-fp32 forward only."""
    T, P, _ = x_tp3.shape
    C3 = w3.shape[0]
    y = torch.empty((T, C3), device=x_tp3.device, dtype=torch.float32)
    am = torch.empty((T, C3), device=x_tp3.device, dtype=torch.int32) if want_argmax else None
    ev = None
    if _o.KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    L = _lib.lib()
    ws, ws_bytes = None, 0
    if 0 < T <= _o.POINTNET_SPLIT_MAX_OBJECTS:      # few objects: split every object over a workgroup's 8 waves (needs a partials buffer)
        ws_bytes = int(L.sga_pointnet_fwd_ws_bytes(T, C3))
        ws = torch.empty((ws_bytes,), device=x_tp3.device, dtype=torch.uint8)
    elif T > 0 and _POINTNET_MODE[get_mfma_mode()] == 4 and C3 == 256:
        # many objects on three planes: 80 KiB of scratch for the l planes of W2 / W3 in operand order (one workgroup then serves whole objects)
        ws_bytes = 81920
        ws = torch.empty((ws_bytes,), device=x_tp3.device, dtype=torch.uint8)
    if bn_sums is not None:
        if bn_sums.dtype != torch.float64 or bn_sums.numel() != 265 + 2 * C3 or not bn_sums.is_contiguous() or bn_sums.device != x_tp3.device:
            raise RuntimeError('sgaligner_amd.pointnet_forward: bn_sums must be a contiguous float64 tensor of 265 + 2 C3 elements on the input device')
        bws_bytes = int(L.sga_pointnet_fwd_bn_ws_bytes(T, C3))
        bws = torch.empty((max(bws_bytes, 8),), device=x_tp3.device, dtype=torch.uint8)
        rc = L.sga_pointnet_fwd_bn(_p(x_tp3), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(y), _p(am),
                                   T, P, C3, _p(ws), ws_bytes, _p(bws), bws_bytes, _p(bn_sums), _POINTNET_MODE[get_mfma_mode()], _stream())
        _lib.check(rc, 'sga_pointnet_fwd_bn')
    else:
        rc = L.sga_pointnet_fwd_ws(_p(x_tp3), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(y), _p(am),
                                   T, P, C3, _p(ws), ws_bytes, _POINTNET_MODE[get_mfma_mode()], _stream())
        _lib.check(rc, 'sga_pointnet_fwd')
    if ev is not None:
        ev[1].record()
        pm = _POINTNET_MODE[get_mfma_mode()]
        _o.KERNEL_EVENTS.setdefault('pointnet_fwd_kernel', []).append(ev + ((T, P, w1.shape[0], w2.shape[0], C3, 'bf16x6' if pm == 4 else 'f32',
                                                                          bn_sums is not None),))
    return y, am


class PointNetFn(torch.autograd.Function):
    """PointNetfeat.forward (reference pointnet.py:120-175) with the sparse max-pool backward."""

    @staticmethod
    def forward(ctx, x_tp3, w1, b1, w2, b2, w3, b3, bn_sums=None):
        x = _req(x_tp3.contiguous(), 'tot_obj_pts')
        ws = [_req(w1.reshape(w1.shape[0], -1).contiguous(), 'conv1.weight'), _req(b1.contiguous(), 'conv1.bias'),
              _req(w2.reshape(w2.shape[0], -1).contiguous(), 'conv2.weight'), _req(b2.contiguous(), 'conv2.bias'),
              _req(w3.reshape(w3.shape[0], -1).contiguous(), 'conv3.weight'), _req(b3.contiguous(), 'conv3.bias')]
        need = any(ctx.needs_input_grad[1:])
        y, am = pointnet_forward(x, *ws, want_argmax=need, bn_sums=bn_sums)
        if need:
            ctx.save_for_backward(x, am, y, *ws)
            ctx.wshapes = (tuple(w1.shape), tuple(w2.shape), tuple(w3.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, am, y, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        T, P, _ = x.shape
        C3 = w3.shape[0]
        gy = gy.contiguous()
        ps = (w1, b1, w2, b2, w3, b3)
        flat = torch.empty((sum(t.numel() for t in ps),), device=x.device, dtype=torch.float32)   # adjacent: the library zeroes it in one launch
        g, o = [], 0
        for t in ps:
            g.append(flat[o:o + t.numel()].view(t.shape))
            o += t.numel()
        rc = _lib.lib().sga_pointnet_bwd(_p(x), _p(am), _p(y), _p(gy), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3),
                                         _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]), _p(g[4]), _p(g[5]), T, P, C3, _POINTNET_MODE[get_mfma_mode()], _stream())
        _lib.check(rc, 'sga_pointnet_bwd')
        s1, s2, s3 = ctx.wshapes
        return None, g[0].reshape(s1), g[1], g[2].reshape(s2), g[3], g[4].reshape(s3), g[5], None


def pointnet(x_tp3, w1, b1, w2, b2, w3, b3, bn_sums=None):
    return PointNetFn.apply(x_tp3, w1, b1, w2, b2, w3, b3, bn_sums)
