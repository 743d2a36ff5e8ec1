"""torch.autograd wrappers around the C-ABI HIP kernels.

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic on the hot path runs in
csrc/*.hip through `sgaligner_amd._lib`.  Every op requires contiguous fp32 HIP ("cuda") tensors and
raises otherwise -- there is no eager/CPU fallback.
"""
from __future__ import annotations

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'sgaligner_amd: `{name}` must be a HIP device tensor (got '
                           f'{t.device if isinstance(t, torch.Tensor) else type(t)}); there is no CPU path')
    if t.dtype != dtype:
        raise RuntimeError(f'sgaligner_amd: `{name}` must be {dtype} (got {t.dtype})')
    if not t.is_contiguous():
        raise RuntimeError(f'sgaligner_amd: `{name}` must be contiguous')
    if t.data_ptr() % 16:
        raise RuntimeError(f'sgaligner_amd: `{name}` must be 16-byte aligned')
    return t


def _p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------ PointNet
def pointnet_forward(x_tp3, w1, b1, w2, b2, w3, b3, want_argmax: bool):
    """x_tp3 [T,P,3] (point-major, as in data_dict['tot_obj_pts']).  Returns (y [T,C3], argmax|None)."""
    T, P, _ = x_tp3.shape
    C3 = w3.shape[0]
    y = torch.empty((T, C3), device=x_tp3.device, dtype=torch.float32)
    am = torch.empty((T, C3), device=x_tp3.device, dtype=torch.int32) if want_argmax else None
    rc = _lib.lib().sga_pointnet_fwd(_p(x_tp3), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(y), _p(am),
                                     T, P, C3, _stream())
    _lib.check(rc, 'sga_pointnet_fwd')
    return y, am
