"""torch.autograd wrappers around the C-ABI HIP kernels.

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic on the hot path runs in
csrc/*.hip through `sgaligner_amd._lib`.  Every op requires contiguous fp32 HIP ("cuda") tensors and
raises otherwise -- there is no eager/CPU fallback.
"""
from __future__ import annotations

import torch

from . import _lib


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  The raw getter is ~20x cheaper than building a
    torch.cuda.Stream object (13 launches per step: 0.17 ms of a 2.5 ms step at the reference's batch sizes)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
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


import os as _os_mode

MFMA_MODES = ('f32', 'bf16x6', 'bf16x3', 'f16', 'f16x2', 'f16x2p')
# Which arithmetic the MFMA kernels with more than one variant use is a POLICY OF THIS LAYER: the C-ABI library is stateless (every entry
# point's arithmetic is in its name or an explicit argument, include/sgaligner_hip.h).  One setting per process, like the reference's
# torch.backends flags; initial value from SGA_MFMA_MODE.
DEFAULT_MFMA_MODE = 'bf16x6'
_MODE = _os_mode.environ.get('SGA_MFMA_MODE', DEFAULT_MFMA_MODE)
if _MODE not in MFMA_MODES:
    raise ValueError(f"sgaligner_amd: SGA_MFMA_MODE must be one of {MFMA_MODES} (got {_MODE!r})")


def set_mfma_mode(mode: str) -> str:
    """Arithmetic of the MFMA kernels that have more than one variant.
    'f32': fp32 MFMA everywhere (v_mfma_f32_*_f32).
    'bf16x6' (THE DEFAULT): the fused 100-d loss sweeps (anchors x negatives: forward sums + gradient) with every fp32 operand split EXACTLY into three
    bf16 terms (8 + 8 + 8 significand bits, fp32's exponent range) and six bf16 MFMAs per product into one fp32 accumulator -- fp32
    arithmetic on the exact operands at 6/16 of the fp32 MFMA's matrix time (csrc/sweep3.hip; SURVEY 7 "fp32 MFMA or split-bf16 x3");
    everything else exact fp32.  M = 2, 3, 4 tables of emb_dim <= 100; other shapes take the 'f32' kernels.
    'bf16x3' (opt-in: each fp32 operand as bf16 hi + lo, 16 bits, three bf16 MFMAs per product; ~1e-5 relative error; PointNet forward +
    the fused loss sweeps); 'f16' (opt-in, BASELINE.json configs[4]: loss tables WIDER than 128 columns and the similarity ranking take
    fp16 inputs with fp32 accumulation -- csrc/wide16.hip, 1e-2 tolerance; the PointNet training forward as in 'f16x2'; 100-d tables and
    everything else stay exact fp32); 'f16x2' (opt-in: the fused loss sweeps with each operand as fp16 hi + lo of 4096 x -- 22 significand
    bits, three fp16 MFMAs per product, csrc/sweeph.hip -- and the PointNet training forward in the same split with every near-tied object
    re-run on the exact-fp32 kernel); 'f16x2p' (the same without the re-run).  Returns the previous mode."""
    global _MODE
    if mode not in MFMA_MODES:
        raise ValueError(f"sgaligner_amd: mfma mode must be one of {sorted(MFMA_MODES)} (got {mode!r})")
    old, _MODE = _MODE, mode
    return old


def get_mfma_mode() -> str:
    return _MODE


# PointNet forward arithmetic per mode (sga_pointnet_fwd_ws `mode`): 0 exact fp32, 1 bf16 hi + lo, 2 fp16 hi + lo + exact re-run of near-ties, 3 without,
# 4 three exact bf16 planes (six bf16 MFMAs per product: fp32 arithmetic on the bf16 matrix pipe, like the default loss sweeps)
# ('f16', the wide-table mode of configs[4], takes the three-plane forward too: with the BatchNorm side effect on, the fp16 split + re-run + one more
#  forward for the sums is slower than the fusable kernel)
_P3 = 4 if _os_mode.environ.get('SGA_POINTNET_P3', '1') != '0' else 0
_POINTNET_MODE = {'f32': 0, 'bf16x6': _P3, 'bf16x3': 1, 'f16': _P3 if _P3 else 2, 'f16x2': 2, 'f16x2p': 3}
POINTNET_TIE_EPS = -1.0                # 'f16x2' forward: < 0 = the library default 2^-17 (tools/dbg/f16x2_pointnet_flips.py sweeps it)
GROUP_LOSS_VALU = _os_mode.environ.get('SGA_GROUP_GRAD_VALU', '0') == '1'      # loss_group kernels: the VALU forms (cross-checks) instead of MFMA


# ------------------------------------------------------------------------------------------ PointNet
POINTNET_LAST_REDO = None              # 'f16x2': [count | object ids] re-run in exact fp32 by the last training forward (diagnostics: tests, bench)
POINTNET_SPLIT_MAX_OBJECTS = 1023      # the library uses the split form below 4 x CUs objects; above that a workspace is not allocated
def pointnet_bn_fusable() -> bool:
    """True when the PointNet forward of the current arithmetic mode is the exact-fp32 kernel, which can deliver the BatchNorm batch
    statistics of the reference's training forward from inside the kernel (sga_pointnet_fwd_bn)."""
    return _POINTNET_MODE[get_mfma_mode()] in (0, 4)


def pointnet_forward(x_tp3, w1, b1, w2, b2, w3, b3, want_argmax: bool, bn_sums=None):
    """x_tp3 [T,P,3] (point-major, as in data_dict['tot_obj_pts']).  Returns (y [T,C3], argmax|None).
    bn_sums: a float64 tensor of 265 + 2 C3 elements to receive the batch-statistic sums of the three pre-activations (layout:
    include/sgaligner_hip.h, sga_pointnet_fwd_bn) -- exact-fp32 forward only."""
    T, P, _ = x_tp3.shape
    C3 = w3.shape[0]
    y = torch.empty((T, C3), device=x_tp3.device, dtype=torch.float32)
    am = torch.empty((T, C3), device=x_tp3.device, dtype=torch.int32) if want_argmax else None
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    L = _lib.lib()
    ws, ws_bytes = None, 0
    global POINTNET_LAST_REDO
    POINTNET_LAST_REDO = None              # set below only when THIS call leaves a re-run list
    if 0 < T <= POINTNET_SPLIT_MAX_OBJECTS:      # few objects: split every object over a workgroup's 8 waves (needs a partials buffer)
        ws_bytes = int(L.sga_pointnet_fwd_ws_bytes(T, C3))
        ws = torch.empty((ws_bytes,), device=x_tp3.device, dtype=torch.uint8)
    elif T > 0 and _POINTNET_MODE[get_mfma_mode()] == 4 and C3 == 256:
        # many objects on three planes: 80 KiB of scratch for the l planes of W2 / W3 in operand order (one workgroup then serves whole objects)
        ws_bytes = 81920
        ws = torch.empty((ws_bytes,), device=x_tp3.device, dtype=torch.uint8)
    elif T > 0 and want_argmax and get_mfma_mode() in ('f16x2', 'f16'):
        # 'f16x2' training forward: [count | ids] of the objects with a near-tied arg-max -> those again on the exact-fp32 kernel
        ws_bytes = 4 * (T + 1)
        ws = POINTNET_LAST_REDO = torch.empty((T + 1,), device=x_tp3.device, dtype=torch.int32)
    if bn_sums is not None:
        if not pointnet_bn_fusable():
            raise RuntimeError(f"sgaligner_amd.pointnet_forward: the fused BatchNorm statistics need the exact-fp32 or the three-plane forward (mode {get_mfma_mode()!r} runs another)")
        if bn_sums.dtype != torch.float64 or bn_sums.numel() != 265 + 2 * C3 or not bn_sums.is_contiguous() or bn_sums.device != x_tp3.device:
            raise RuntimeError('sgaligner_amd.pointnet_forward: bn_sums must be a contiguous float64 tensor of 265 + 2 C3 elements on the input device')
        bws_bytes = int(L.sga_pointnet_fwd_bn_ws_bytes(T, C3))
        bws = torch.empty((max(bws_bytes, 8),), device=x_tp3.device, dtype=torch.uint8)
        rc = L.sga_pointnet_fwd_bn(_p(x_tp3), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(y), _p(am),
                                   T, P, C3, _p(ws), ws_bytes, _p(bws), bws_bytes, _p(bn_sums), _POINTNET_MODE[get_mfma_mode()], _stream())
        _lib.check(rc, 'sga_pointnet_fwd_bn')
    else:
        rc = L.sga_pointnet_fwd_ws(_p(x_tp3), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(y), _p(am),
                                   T, P, C3, _p(ws), ws_bytes, _POINTNET_MODE[get_mfma_mode()], float(POINTNET_TIE_EPS), _stream())
        _lib.check(rc, 'sga_pointnet_fwd')
    if ev is not None:
        ev[1].record()
        pm = _POINTNET_MODE[get_mfma_mode()]
        KERNEL_EVENTS.setdefault('pointnet_fwd_kernel', []).append(ev + ((T, P, w1.shape[0], w2.shape[0], C3, 'bf16x6' if pm == 4 else get_mfma_mode() if (ws is not None and ws_bytes == 4 * (T + 1)) else 'f32',
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
                                         _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]), _p(g[4]), _p(g[5]), T, P, C3, _stream())
        _lib.check(rc, 'sga_pointnet_bwd')
        s1, s2, s3 = ctx.wshapes
        return None, g[0].reshape(s1), g[1], g[2].reshape(s2), g[3], g[4].reshape(s3), g[5], None


def pointnet(x_tp3, w1, b1, w2, b2, w3, b3, bn_sums=None):
    return PointNetFn.apply(x_tp3, w1, b1, w2, b2, w3, b3, bn_sums)


# ------------------------------------------------------------------------------------------ GEMM / Linear
def gemm(a, b, trans_a: bool, trans_b: bool, m: int, n: int, k: int, bias=None, out=None, accumulate=False):
    """out[m,n] (+)= op(a)[m,k] @ op(b)[k,n] (+ bias).  a may be float64 (converted in the loader)."""
    dev = b.device
    if out is None:
        out = torch.empty((m, n), device=dev, dtype=torch.float32)
    lda = a.stride(0)
    ldb = b.stride(0)
    rc = _lib.lib().sga_gemm(int(trans_a), int(trans_b), m, n, k, _p(a), lda, int(a.dtype == torch.float64), _p(b), ldb,
                             _p(out), out.stride(0), _p(bias), int(accumulate), _stream())
    _lib.check(rc, 'sga_gemm')
    return out


def colsum(x, out=None):
    m, n = x.shape
    if out is None:
        out = torch.empty((n,), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().sga_colsum(_p(x), x.stride(0), m, n, _p(out), 0, _stream()), 'sga_colsum')
    return out


def cast_f32(x):
    if x.dtype == torch.float32:
        return x
    if x.dtype != torch.float64:
        raise RuntimeError(f'sgaligner_amd: unsupported feature dtype {x.dtype}')
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().sga_cast_f64_f32(_p(x), _p(out), x.numel(), _stream()), 'sga_cast_f64_f32')
    return out


class LinearFn(torch.autograd.Function):
    """y = x W^T + b  (nn.Linear; reference sg_aligner.py:112,116,119,122).  x may be float64."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise RuntimeError('sgaligner_amd.LinearFn: HIP device tensor required; there is no CPU path')
        if x.dtype not in (torch.float32, torch.float64):
            raise RuntimeError(f'sgaligner_amd.LinearFn: features must be float32 or float64 (the collated bag-of-words '
                               f'tables), got {x.dtype}')
        x = x.contiguous()
        _req(weight, 'weight'); _req(bias, 'bias')
        t, k = x.shape
        if x.dim() != 2 or k != weight.shape[1]:
            raise RuntimeError(f'sgaligner_amd.LinearFn: x is {tuple(x.shape)} but weight is {tuple(weight.shape)}')
        y = gemm(x, weight, False, True, t, weight.shape[0], k, bias=bias)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        t, k = x.shape
        n = weight.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(gy, weight.t().contiguous(), False, True, t, k, n)   # dX = dY W, as dY (W^T)^T: the prefetching NT kernel
        if ctx.needs_input_grad[1]:
            if k % 4 and t >= 4096:
                # an input width that is not a multiple of 4 (the 41 relation / 3 pose columns) misses the row-major TN kernel's alignment and
                # fell to the generic one: 1.9 ms per weight gradient at configs[4]'s 1024-d embeddings (6 of them: 14 % of that step).  Zero
                # columns up to the next multiple of 4 cost nothing and are cut off again.
                kp = (k + 3) // 4 * 4
                xp = torch.zeros((t, kp), device=x.device, dtype=torch.float32)
                xp[:, :k] = x
                gw = gemm(gy, xp, True, False, n, kp, t)[:, :k].contiguous()
            else:
                gw = gemm(gy, cast_f32(x), True, False, n, k, t)             # dW = dY^T X
        if ctx.needs_input_grad[2]:
            gb = colsum(gy)
        return gx, gw, gb


def linear(x, weight, bias):
    return LinearFn.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------ Fusion
import ctypes as _ct


def _ptr_array(tensors):
    arr = (_ct.c_void_p * len(tensors))(*[(t.data_ptr() if t is not None else None) for t in tensors])
    return arr


def _ev_start():
    """HIP-event pair around a launch on torch's current stream when bench.py asks for per-kernel timings (ops.KERNEL_EVENTS is a dict)."""
    if KERNEL_EVENTS is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    return ev


def _ev_stop(ev, key, shape):
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.setdefault(key, []).append(ev + (shape,))


class FusionFn(torch.autograd.Function):
    """MultiModalFusion.forward (reference sg_aligner.py:30-35)."""

    @staticmethod
    def forward(ctx, weight, *embs):
        m = len(embs)
        embs = [_req(e.contiguous(), f'embs[{i}]') for i, e in enumerate(embs)]
        w = _req(weight.contiguous(), 'fusion.weight')
        t, d = embs[0].shape
        for e in embs:
            if tuple(e.shape) != (t, d):
                raise RuntimeError('sgaligner_amd.FusionFn: all modality tables must share one shape')
        joint = torch.empty((t, m * d), device=w.device, dtype=torch.float32)
        arr = _ptr_array(embs)
        ev = _ev_start()
        _lib.check(_lib.lib().sga_fusion_fwd(arr, m, _p(w), _p(joint), t, d, _stream()), 'sga_fusion_fwd')
        _ev_stop(ev, 'fusion_fwd', (t, d, m))
        ctx.save_for_backward(w, *embs)
        return joint

    @staticmethod
    def backward(ctx, gj):
        w, *embs = ctx.saved_tensors
        m = len(embs)
        t, d = embs[0].shape
        gj = gj.contiguous()
        gembs = [torch.empty_like(e) for e in embs]
        gw = torch.empty_like(w)
        nb = _lib.lib().sga_fusion_bwd_workspace_bytes(m)
        ws = torch.empty((nb,), device=w.device, dtype=torch.uint8)
        ev = _ev_start()
        _lib.check(_lib.lib().sga_fusion_bwd(_ptr_array(embs), m, _p(w), _p(gj), _ptr_array(gembs), _p(gw), t, d,
                                             _p(ws), nb, _stream()), 'sga_fusion_bwd')
        _ev_stop(ev, 'fusion_bwd', (t, d, m))
        return (gw, *gembs)


def fusion(weight, embs):
    return FusionFn.apply(weight, *embs)


# ------------------------------------------------------------------------------------------ contrastive loss
import hashlib as _hashlib
import os as _os
from collections import OrderedDict as _OrderedDict

import numpy as _np


def _fingerprint(arrays, extra=()):
    """Content fingerprint of small host arrays: a cached device copy is reused only while the arrays a caller hands in
    still hold the same values (the reference's tester shifts e1i/e2i IN PLACE between uses, inference_align_reg.py:119-120)."""
    h = _hashlib.blake2b(digest_size=16)        # a cryptographic digest: a collision between two batches is not a practical event
    shapes = []
    for a in arrays:
        a = _np.ascontiguousarray(a)
        shapes.append((a.shape, a.dtype.str))
        h.update(a.reshape(-1).view(_np.uint8))
    return (h.digest(), tuple(shapes), tuple(extra))


class _SmallCache:
    """Tiny LRU keyed by content fingerprints.  Lives HERE, never in the caller's data_dict: a dict that is reused with
    different index arrays can not pick up a stale device copy."""

    def __init__(self, n=4):
        self.n, self.d = n, _OrderedDict()

    def get(self, key, make):
        v = self.d.get(key)
        if v is None:
            v = make()
            self.d[key] = v
            while len(self.d) > self.n:
                self.d.popitem(last=False)
        else:
            self.d.move_to_end(key)
        return v

    def clear(self):
        self.d.clear()


_H2D_RING = {'n': 0, 'bufs': [None] * 8, 'evs': [None] * 8}


def _h2d(host_array, device):
    """Small per-batch host array -> device without stalling the host: a pageable `tensor.to(device)` is a blocking copy that
    waits for everything queued on the stream (the previous step's backward); from a pinned staging tensor the copy is
    asynchronous and the host runs on.  CPU 'devices' (unit tests) take the plain path."""
    arr = _np.ascontiguousarray(host_array)
    t = torch.from_numpy(arr)
    device = torch.device(device)
    if device.type != 'cuda':
        return t.to(device)
    # a ring of reusable pinned byte buffers (Tensor.pin_memory() registers fresh memory on every call: ~1 ms, twice per batch)
    slot = _H2D_RING['n'] % len(_H2D_RING['bufs'])
    _H2D_RING['n'] += 1
    buf, ev = _H2D_RING['bufs'][slot], _H2D_RING['evs'][slot]
    if ev is not None:
        ev.synchronize()                                   # the copy that last read this buffer, several uploads ago
    nbytes = arr.nbytes
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(nbytes * 2, 4096),), dtype=torch.uint8).pin_memory()
        _H2D_RING['bufs'][slot] = buf
    if nbytes:
        _np.copyto(buf.numpy()[:nbytes], arr.reshape(-1).view(_np.uint8))
    out = buf[:nbytes].view(t.dtype).view(t.shape).to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _H2D_RING['evs'][slot] = ev
    return out


VALIDATE = _os.environ.get('SGA_VALIDATE', '1') != '0'     # host-side range checks of index sets / edge lists, once per batch


class _DeferredChecks:
    """Range checks whose operands live on the device (the collated edge list): reading the answer back at once would stall the
    host on everything already queued -- once per batch, i.e. once per training step.  The [min, max] pair is reduced on the
    device, copied to a pinned host slot without blocking, and examined when its event has fired: at the next batch's check (by
    then it has) or at `flush()` (blocking; trainers call it at epoch ends, tests directly).  A bad batch therefore raises at the
    latest one step after it was used; the kernels themselves never read out of bounds (they drop such endpoints)."""

    def __init__(self):
        self.pending = []          # (event, pinned host tensor, n values, verdict(values) -> error text or None)
        self.free = []

    def submit_fn(self, dev_tensor, verdict):
        """Queue `verdict(list of ints)` on the (<= 2 element, int32/int64) device tensor's values, read back without blocking."""
        n = int(dev_tensor.numel())
        host = self.free.pop() if self.free else torch.empty((2,), dtype=torch.int64).pin_memory()
        view = host[:n] if dev_tensor.dtype == torch.int64 else host.view(torch.int32)[:n]
        view.copy_(dev_tensor.reshape(-1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, host, view, verdict))

    def submit(self, minmax_dev, upper, what):
        upper = int(upper)
        self.submit_fn(minmax_dev, lambda v: (what % (v[0], v[1], upper)) if (v[0] < 0 or v[1] >= upper) else None)

    def poll(self, wait=False):
        keep = []
        err = None
        for ev, host, view, verdict in self.pending:
            if wait:
                ev.synchronize()
            if wait or ev.query():
                msg = verdict([int(x) for x in view])
                self.free.append(host)
                if msg is not None and err is None:
                    err = msg
            else:
                keep.append((ev, host, view, verdict))
        self.pending = keep
        if err is not None:
            raise RuntimeError(err)

    def flush(self):
        self.poll(wait=True)


DEFERRED_CHECKS = _DeferredChecks()


class IndexSets:
    """Device copy of the four host index arrays of a batch (reference scan3r.py:142-173 keeps them as
    numpy int32 on the host): packed [e1i | e2i | e1j | e2j], converted once per batch.
    `groups` (optional, loss_group=b): group id of every entry, same packing -- rows only interact within their group."""

    def __init__(self, data_dict, device, n_rows=None):
        arrs = [_np.ascontiguousarray(_np.asarray(data_dict[k]).astype(_np.int32)) for k in ('e1i', 'e2i', 'e1j', 'e2j')]
        if arrs[0].shape != arrs[1].shape:
            raise RuntimeError('sgaligner_amd: e1i and e2i must have the same length')
        self.A, self.J1, self.J2 = int(arrs[0].shape[0]), int(arrs[2].shape[0]), int(arrs[3].shape[0])
        self.R = 2 * self.A + self.J1 + self.J2
        host = _np.concatenate(arrs)
        if VALIDATE and host.size:
            lo, hi = int(host.min()), int(host.max())
            if lo < 0 or (n_rows is not None and hi >= n_rows):
                raise RuntimeError(f'sgaligner_amd: e1i/e2i/e1j/e2j hold object indices in [{lo}, {hi}] but the embedding '
                                   f'tables have {n_rows} rows')
        self.idx = _h2d(host, device)

    _cache = _SmallCache()

    @staticmethod
    def from_device(idx, A, J1, J2):
        """Wrap an already-packed device index array (the multi-GPU path gathers it on the device, dist.py)."""
        s = IndexSets.__new__(IndexSets)
        s.A, s.J1, s.J2 = int(A), int(J1), int(J2)
        s.R = 2 * s.A + s.J1 + s.J2
        if idx.dtype != torch.int32 or idx.numel() != s.R:
            raise RuntimeError('sgaligner_amd: packed index array must be int32 of length 2A+J1+J2')
        s.idx = idx.contiguous()
        return s

    @staticmethod
    def of(data_dict, device, n_rows=None):
        pre = data_dict.get('_sga_index_sets') if isinstance(data_dict, dict) else None
        if pre is not None:                          # set only by AlignerSteps._global_loss in its OWN dict
            return pre
        device = torch.device(device)
        key = _fingerprint([_np.asarray(data_dict[k]) for k in ('e1i', 'e2i', 'e1j', 'e2j')], (str(device), n_rows))
        return IndexSets._cache.get(key, lambda: IndexSets(data_dict, device, n_rows))


FUSED_ANCHOR_BWD = True
FUSED_AA_ONEPASS = True      # training: A x A terms + gradients from one pass in forward() when the loss head announces dL/d(terms)
# One-pass mode: walk the anchors x anchors pairs SYMMETRICALLY -- a block evaluates (i, j) and (j, i) from the same two similarities, every
# unordered pair once (sga_loss_anchor_multi_bwd_sym: -34 % per ordered pair, tools/bench_aa.py); across ranks by _sym_jobs.
AA_SYMMETRIC = True
AA_SYMMETRIC_MAX_M = int(_os.environ.get('SGA_AA_SYM_MAX_M', '4'))     # tools flip this to 3 to time M = 4 on the ordered walk
ONEPASS_MIN_ANCHORS = 256    # below this the A x A work is negligible and the saved gradients' bookkeeping is not worth its launches
WIDE_STASH = True         # tables wider than 128 columns: coefficient stash + GEMMs instead of the multi-pass gradient sweep (tests flip it)
FUSED_ANCHOR_FWD = True   # tests flip this to cross-check the two anchors x anchors forward kernels
KERNEL_EVENTS = None   # bench.py sets this to {} to time the dominant kernel with HIP events on the launch stream

def _default_stash_bytes():
    """Bound on the transposed dL/dS stashes of the A x A backward (and the wide-table coefficient stashes): 16 GiB on a device with
    >= 128 GiB of memory (MI355X: 288 GB -- 18 anchor-row blocks instead of 69 at configs[2], +1.3 % step rate, 37.8 GiB peak),
    4 GiB otherwise; env SGA_STASH_BYTES overrides.  Resolved at FIRST USE from the process's current device (importing this module
    neither initialises the HIP runtime nor looks at device 0 of a multi-GPU host)."""
    env = _os.environ.get('SGA_STASH_BYTES')
    if env:
        return int(env)
    try:
        if torch.cuda.is_available() and torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory >= (128 << 30):
            return 16 << 30
    except Exception:
        pass
    return 4 << 30


STASH_BYTES = None          # None: _default_stash_bytes() at first use; tests / callers may assign a number
_stash_default = []


def _stash_bytes():
    if STASH_BYTES is not None:
        return int(STASH_BYTES)
    if not _stash_default:
        _stash_default.append(_default_stash_bytes())
    return _stash_default[0]


def _anchor_chunks(a_lo, a_hi, A, n_tables):
    """Anchor-row blocks [c_lo, c_hi) of the anchors x anchors backward.  The coefficient stash of a block is
    n_tables x [A, c_hi - c_lo] fp32; blocks are sized so that it never exceeds STASH_BYTES, which keeps the loss
    backward O(A * D + STASH_BYTES) in memory whatever the batch (4096 pairs x 128 objects: A = 155 648, a full stash
    would be 3 x 97 GB).  Block sizes are multiples of 32 rows (kernel tile) except the last."""
    ns = a_hi - a_lo
    if ns <= 0 or A <= 0:
        return []
    rows = max(32, (_stash_bytes() // (4 * A * max(1, n_tables))) // 32 * 32)
    return [(c, min(c + rows, a_hi)) for c in range(a_lo, a_hi, rows)]


def _sym_chunks(A, n_tables):
    """Blocks of the SYMMETRIC anchors x anchors walk (csrc/contrastive.hip, anchor_multi_bwd16_kernel<.., SYM>): block [lo, hi) meets
    the columns >= lo and keeps two stashes, [A - lo, hi - lo] and [A - hi, hi - lo] floats per table, bounded together by STASH_BYTES --
    so blocks get taller as the walk moves right.  32-row boundaries except the end."""
    return [(lo, hi) for lo, hi, _, _, _ in _sym_jobs([0, A], 0, n_tables)]


def _sym_jobs(cuts, rank, n_tables):
    """The symmetric walk of ONE RANK of an anchor-sharded job (new design): every UNORDERED pair of anchors is visited once
    over all ranks, and every rank visits the same number of pairs.  cuts = [0, c_1, ..., A]: rank r owns the anchor rows
    [c_r, c_r+1) (multiples of 32).  With R ranks, rank r evaluates its own diagonal square and the rectangles (its rows) x (the rows of
    the next K ranks, cyclically), K = (R - 1) / 2 for odd R; for even R the ranks of the lower half take K = R / 2 and the upper half
    R / 2 - 1 (a pair of blocks R / 2 apart is visited by its lower rank only).  Returns launches (lo, hi, j_lo, j_hi, mir) for
    sga_loss_anchor_multi_bwd_symx / sga_loss_stash_grad_symx: rows [lo, hi) x columns [j_lo, j_hi), mirrored elements from column mir on;
    the two stashes of a launch, (j_hi - j_lo) + (j_hi - mir) rows of hi - lo floats per table, stay within STASH_BYTES (blocks get
    taller as the columns left to meet get fewer).  One rank, cuts = [0, A]: the single-GPU walk."""
    R = len(cuts) - 1
    A = cuts[-1]
    lo_r, hi_r = cuts[rank], cuts[rank + 1]
    if hi_r <= lo_r:
        return []
    K = (R - 1) // 2 if R % 2 else (R // 2 if rank < R // 2 else R // 2 - 1)
    right_end = cuts[min(rank + K, R - 1) + 1]                     # contiguous columns right of the own square
    wrap_end = cuts[(rank + K) % R + 1] if rank + K >= R else 0   # columns [0, wrap_end) of the ranks the cyclic order wraps to
    q = _stash_bytes() // (4 * max(1, n_tables))
    jobs, lo = [], lo_r
    while lo < hi_r:
        # columns this block meets: [lo, right_end) (own square ordered up to hi, mirrored from hi on) + [0, wrap_end) (all mirrored)
        per_row = 2 * (right_end - lo) + 2 * wrap_end
        rows = max(32, (q // max(1, per_row)) // 32 * 32)
        hi = min(lo + rows, hi_r)
        jobs.append((lo, hi, lo, right_end, hi))
        if wrap_end > 0:
            jobs.append((lo, hi, 0, wrap_end, 0))
        lo = hi
    return jobs


# What sga_loss_multi_grad launches (bench.py's roofline line): two owner sweeps x M tables x (S with K = 100 + gradient
# GEMM with 112 columns); the joint table is derived, never multiplied.
SWEEP_GRAD_INFO = {
    'tag': 'sweep16_kernel<%d,true>',           # M = 4 launches sweep16x2_kernel<true> (paired waves, two tables each)
    'what': 'loss: negatives backward',
    'executed_flops': lambda ns, j, m: 2.0 * (2.0 * ns * j) * 2.0 * m * (100 + 112),
}

SWEEP_SUMS_INFO = {                             # sga_loss_multi_sums: one owner sweep, S only (K = 100), the joint table derived
    'tag': 'sweep16_kernel<%d,false>',
    'what': 'loss: global sums over anchors x negatives (forward)',
    'executed_flops': lambda ns, j, m: (2.0 * ns * j) * 2.0 * m * 100,
}
BF16X3_COVERAGE = 'PointNet forward + loss sweeps on bf16 MFMA with hi/lo-split operands, fp32 accumulate'
F16X2_COVERAGE = ('anchors x negatives loss sweeps (forward sums + gradient) on fp16 MFMA with operands split into fp16 hi + lo of 4096 x '
                  '(22 significand bits), rows centred, fp32 accumulate; the similarities of the symmetric anchors x anchors kernel in the same split; the PointNet forward in the same split with every object whose point '
                  'max is near-tied (2^-17) re-run on the exact-fp32 kernel (same arg-max points as exact fp32); everything else exact fp32')
F16X2P_COVERAGE = ('the f16x2 loss sweeps + the PointNet forward in the same split WITHOUT the near-tie re-run (a tied point max may pick the other '
                   'point: not faithful); everything else exact fp32')
# 'f16x2' gradient sweep: the coefficients dL/dS as fp16 hi + lo (True) or rounded to fp16 (False: an independent, unbiased 2^-12 rounding
# per (anchor, negative) pair; 7 of 31 MFMAs and 1.5 VALU per pair less).  None (default) = hi + lo unless EVERY gradient row sums at least
# F16X2_COEF_LO_MIN_TERMS pairs: there the rounding noise of a row, 2^-12 / sqrt(terms), is below the accumulation error the exact-fp32
# sweep itself carries (8e-7 .. 2.5e-6 of a table gradient's maximum against fp64).  2^20: configs[2] (311 296 pairs per row)
# keeps hi + lo -- dropping lo there is +6 % pairs/s (1 447 vs 1 360) but puts meta_embedding_rel.* (the small remainder of cancelling sums
# over a table of nearly identical rows) at 2.9 .. 4.3 x its fp32 rerun noise from run to run instead of 2.3 .. 2.9 x.
# 'f16x2': the A x A stash products on split-fp16 MFMA (csrc/stashh.hip).  OFF by default: 7.4 vs 8.4 ms per 2048 x 155 648 block for the exact-fp32
# GEMMs (tools/bench_aa.py; plus one pass over the stash for its largest |value|) -- at best -0.04 s of a 2.9 s configs[2] step -- while the gate's margin on meta_embedding_rel.bias shrinks from 3.7 to 4.0 x
# the rerun noise (profiles/r04_v_bench_c3.json).  Kept as a measured experiment with its C-ABI test.
BF16X6_STASH = _os.environ.get('SGA_BF16X6_STASH', '1') != '0'   # 'bf16x6': the A x A stash products on the sweeps' three exact bf16 planes (off: fp32-MFMA GEMMs)
F16X2_STASH = _os.environ.get('SGA_F16X2_STASH', '0') == '1'
F16X2_AA = _os.environ.get('SGA_F16X2_AA', '1') != '0'       # 'f16x2': the symmetric A x A kernel's similarities on split-fp16 MFMA (tools / tests flip it)
F16X2_COEF_LO = {'1': True, '0': False}.get(_os.environ.get('SGA_F16X2_COEF_LO', ''), None)
F16X2_COEF_LO_MIN_TERMS = 1 << 20


# 'f16x2' forward sums: full three-product similarities (True) or hi.hi only on the 96 main columns (False: 4 instead of 10 MFMAs per tile).
# None (default) = full unless the smallest of the four global sums has at least F16X2_SUMS_LO_MIN_TERMS terms.
F16X2_SUMS_LO = {'1': True, '0': False}.get(_os.environ.get('SGA_F16X2_SUMS_LO', ''), None)
F16X2_SUMS_LO_MIN_TERMS = 1 << 24


def _f16x2_sums_lo(ns, J1, J2):
    if F16X2_SUMS_LO is not None:
        return bool(F16X2_SUMS_LO)
    return ns * min(J1, J2) < F16X2_SUMS_LO_MIN_TERMS


def _f16x2_coef_lo(ns, J1, J2):
    if F16X2_COEF_LO is not None:
        return bool(F16X2_COEF_LO)
    return min(J1 + J2, 2 * ns) < F16X2_COEF_LO_MIN_TERMS

TAU_ICL = 0.1      # losses.py:39 (ctor argument ignored by the reference)
TAU_IAL = 1.0      # losses.py:63
ALPHA = 0.5        # losses.py:36,60 defaults


class ContrastiveTermsFn(torch.autograd.Function):
    """Raw loss sums for NT tables (modalities..., joint):
        out[k]         = sum_ij -log(a qA + (1-a) qB)          k < NT     (ICL, tau 0.1)
        out[NT+m]      = sum_ij exp(qoA)(qoA - log qmA)        m < NT-1   (IAL a, tau 1, qm from the last table)
        out[NT+M+m]    = same with the B direction
    (reference losses.py:5-15,43-58,68-97).  NT == 1 -> ICL only."""

    @staticmethod
    def forward(ctx, index_sets, alpha, shard, reduce, *tables):
        """shard = (a_lo, a_hi[, ...]) / reduce: as in FusedContrastiveFn -- this rank evaluates its anchors' share of every global sum and loss
        term (all-reduced: the returned values are the batch-global ones on every rank) and, in backward, its share of dL/dE for ALL rows."""
        L = _lib.lib()
        nt = len(tables)
        m = nt - 1 if nt > 1 else 0
        a_lo, a_hi = (0, index_sets.A) if shard is None else (int(shard[0]), int(shard[1]))
        full = a_lo == 0 and a_hi == index_sets.A
        tables = [_req(t.contiguous(), f'table[{i}]') for i, t in enumerate(tables)]
        dev = tables[0].device
        s = index_sets
        T = tables[0].shape[0]
        zs, nrms, dps, zhs, zts = [], [], [], [], []
        f16 = get_mfma_mode() == 'f16'
        sums = torch.empty((nt, 8), device=dev, dtype=torch.float64)
        st = _stream()
        slots = 1 + L.sga_loss_slots()          # scalar accumulators are [result | per-wave slots] (contrastive.hip)
        for k, e in enumerate(tables):
            d = e.shape[1]
            dp = (d + 7) // 8 * 8
            z = torch.empty((s.R, dp), device=dev, dtype=torch.float32)
            nrm = torch.empty((s.R,), device=dev, dtype=torch.float32)
            _lib.check(L.sga_loss_gather(_p(e), T, d, _p(s.idx), s.R, _p(z), dp, _p(nrm), st), 'sga_loss_gather')
            sk = torch.empty((slots * 8,), device=dev, dtype=torch.float64)
            zh = zt = None
            if dp > 128 and not full:
                raise RuntimeError('sgaligner_amd: anchor sharding of the general loss path is implemented for tables of at most 128 columns')
            if dp > 128 and f16:
                # opt-in fp16-input MFMA for wide tables (configs[4]): fp16 copies of the normalised table, once per step
                ldt = int(L.sga_wide16_ldt(s.A, s.J1, s.J2))
                zh = torch.empty((max(s.R, 1), dp), device=dev, dtype=torch.float16)
                zt = torch.empty((dp, ldt), device=dev, dtype=torch.float16)
                _lib.check(L.sga_wide16_prepare(_p(z), dp, s.A, s.J1, s.J2, _p(zh), _p(zt), st), 'sga_wide16_prepare')
                ev = None
                if KERNEL_EVENTS is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                _lib.check(L.sga_loss_neg_sums_f16(_p(zh), dp, s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sk), st), 'sga_loss_neg_sums_f16')
                if ev is not None:
                    ev[1].record()
                    KERNEL_EVENTS.setdefault('wide16_sums', []).append(ev + ((s.A, s.J1, s.J2, dp),))
            else:
                _lib.check(L.sga_loss_neg_sums_shard(_p(z), dp, s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sk), a_lo, a_hi, st), 'sga_loss_neg_sums')
            sums[k].copy_(sk[:8])
            zs.append(z); nrms.append(nrm); dps.append(dp); zhs.append(zh); zts.append(zt)
        out = torch.empty((slots * (nt + 2 * m),), device=dev, dtype=torch.float64)
        zarr = _ptr_array(zs)
        dparr = (_ct.c_int * nt)(*dps)
        # (mode 'f16': the wide tables' anchors x anchors similarities take their fp16 copies as well -- fp16 inputs, fp32 accumulate)
        sums = _allreduce_sum(sums, reduce)
        _lib.check(L.sga_loss_anchor_fwd_f16(zarr, _ptr_array(zhs), dparr, nt, s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(out), a_lo, a_hi, st),
                   'sga_loss_anchor_fwd')
        out = _allreduce_sum(out[:nt + 2 * m].contiguous(), reduce)
        ctx.shard, ctx.reduce = (a_lo, a_hi), reduce
        ctx.s, ctx.alpha, ctx.dps, ctx.nt = s, float(alpha), dps, nt
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.f16 = [zh is not None for zh in zhs]
        ctx.save_for_backward(sums, *zs, *nrms, *[t for t in zhs if t is not None], *[t for t in zts if t is not None])
        return out[:nt + 2 * m].float()

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        s, nt, dps = ctx.s, ctx.nt, ctx.dps
        sums, *rest = ctx.saved_tensors
        zs, nrms = rest[:nt], rest[nt:2 * nt]
        n16 = sum(ctx.f16)
        h_it, t_it = iter(rest[2 * nt:2 * nt + n16]), iter(rest[2 * nt + n16:])
        zhs = [next(h_it) if f else None for f in ctx.f16]
        zts = [next(t_it) if f else None for f in ctx.f16]
        dev = sums.device
        st = _stream()
        coef = gout.contiguous().float()
        A = s.A
        slots = 1 + L.sga_loss_slots()
        dparr = (_ct.c_int * nt)(*dps)
        dzs = [torch.zeros((s.R, dp), device=dev, dtype=torch.float32) for dp in dps]
        gs = torch.zeros((nt, 8), device=dev, dtype=torch.float64)
        a_lo, a_hi = ctx.shard
        chunks = _anchor_chunks(a_lo, a_hi, A, nt)
        if chunks:
            cmax = max(hi - lo for lo, hi in chunks)
            m1 = [torch.empty((A * cmax,), device=dev, dtype=torch.float32) for _ in range(nt)]
            gsc = torch.empty((slots, nt, 8), device=dev, dtype=torch.float64)
            for lo, hi in chunks:          # bounded stash: one anchor-row block at a time
                _lib.check(L.sga_loss_anchor_bwd_f16(_ptr_array(zs), _ptr_array(zhs), dparr, nt, A, _p(sums), ctx.alpha, TAU_ICL, TAU_IAL, _p(coef),
                                                     _ptr_array(m1), _p(gsc), lo, hi, st), 'sga_loss_anchor_bwd')
                gs += gsc[0]
                for k in range(nt):
                    # dX1[i] = sum_j G[i,j] X2[j]  (M1 = G^T),  dX2[j] = sum_i G[i,j] X1[i]
                    _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zs[k]), A, dps[k], _p(dzs[k]), lo, hi, st), 'sga_loss_stash_grad')
            del m1
        gs = _allreduce_sum(gs, ctx.reduce)                      # dL/d(global sums) needs every shard's anchors x anchors tiles
        grads = []
        for k in range(nt):
            z, dp, dz = zs[k], dps[k], dzs[k]
            ev = None
            if KERNEL_EVENTS is not None and dp <= 128:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if zhs[k] is not None:
                # opt-in fp16-input MFMA (configs[4]): S and both gradient GEMMs on v_mfma_f32_32x32x16_f16 (csrc/wide16.hip)
                need = int(L.sga_loss_neg_grad_f16_bytes(A, s.J1, s.J2))
                have = max(min(need, _stash_bytes()), int(L.sga_loss_neg_grad_f16_bytes(min(A, 128), s.J1, s.J2)))
                stash = torch.empty((have,), device=dev, dtype=torch.uint8)
                ev16 = None
                if KERNEL_EVENTS is not None:
                    ev16 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev16[0].record()
                _lib.check(L.sga_loss_neg_grad_f16(_p(zhs[k]), _p(zts[k]), dp, A, s.J1, s.J2, TAU_ICL, TAU_IAL, gs[k].data_ptr(), _p(dz),
                                                   _p(stash), have, st), 'sga_loss_neg_grad_f16')
                if ev16 is not None:
                    ev16[1].record()
                    KERNEL_EVENTS.setdefault('wide16_grad', []).append(ev16 + ((A, s.J1, s.J2, dp),))
                del stash
            elif dp > 128 and WIDE_STASH:
                # wide rows: S is the expensive part -> coefficient stash + GEMMs, S computed once (csrc/contrastive.hip, sweep_coef_kernel)
                need = int(L.sga_loss_neg_grad_wide_floats(A, s.J1, s.J2))
                have = max(min(need, _stash_bytes() // 4), 2 * (s.J1 + s.J2) * min(A, 32))
                stash = torch.empty((have,), device=dev, dtype=torch.float32)
                _lib.check(L.sga_loss_neg_grad_wide(_p(z), dp, A, s.J1, s.J2, TAU_ICL, TAU_IAL, gs[k].data_ptr(), _p(dz), _p(stash), have, st),
                           'sga_loss_neg_grad_wide')
                del stash
            else:
                _lib.check(L.sga_loss_neg_grad_shard(_p(z), dp, A, s.J1, s.J2, TAU_ICL, TAU_IAL, gs[k].data_ptr(), _p(dz), a_lo, a_hi, st),
                           'sga_loss_neg_grad')
            if ev is not None:
                ev[1].record()
                KERNEL_EVENTS.setdefault('sweep_kernel<4,4,grad>', []).append(ev + ((A, s.J1, s.J2, dp),))
            t, d = ctx.shapes[k]
            de = torch.zeros((t, d), device=dev, dtype=torch.float32)
            _lib.check(L.sga_loss_scatter(_p(dz), _p(z), _p(nrms[k]), _p(s.idx), s.R, d, dp, _p(de), st), 'sga_loss_scatter')
            grads.append(de)
            dzs[k] = None
        return (None, None, None, None, *grads)


def contrastive_terms(tables, data_dict, alpha=ALPHA, shard=None, reduce=None):
    """shard / reduce: the anchor range this rank owns and an in-place SUM all-reduce (one process per GPU); None = everything here."""
    s = IndexSets.of(data_dict, tables[0].device, int(tables[0].shape[0]))
    return ContrastiveTermsFn.apply(s, alpha, shard, reduce, *tables), s


class LossHeadFn(torch.autograd.Function):
    """[loss, icl_unimodal, icl_multimodal, ial] from the raw loss terms and the two log_vars vectors: reference
    losses.py:114-152 + CustomMultiLossLayer.forward :28-34 as ONE launch forward and ONE backward (csrc/losshead.hip) instead of
    ~60 one-element torch kernels -- a fifth of all launches of a step at the reference's batch sizes."""

    @staticmethod
    def forward(ctx, sums, lv_ial, lv_icl, n_anchors, z_ial, alpha_ial, zoom):
        M = int(lv_ial.numel())
        if sums.dtype not in (torch.float32, torch.float64) or sums.numel() != 3 * M + 1 or lv_icl.numel() != M:
            raise RuntimeError('sgaligner_amd: LossHeadFn takes the 3M+1 loss terms (float32/float64) and two [M] log_vars vectors')
        sums = sums.contiguous()
        la, lc = _req(lv_ial.detach(), 'log_vars (ial)'), _req(lv_icl.detach(), 'log_vars (icl)')
        out = torch.empty((4,), device=sums.device, dtype=torch.float64)
        # a batch without anchors: the reference's .mean() over an empty A x A matrix is NaN, not an exception (losses.py:57)
        inv_aa = 1.0 / float(n_anchors * n_anchors) if n_anchors else float('nan')
        ctx.consts = (M, inv_aa, float(z_ial), float(alpha_ial), float(zoom))
        ctx.f64 = int(sums.dtype == torch.float64)
        _lib.check(_lib.lib().sga_loss_head_fwd(_p(sums), ctx.f64, _p(la), _p(lc), *ctx.consts, _p(out), _stream()), 'sga_loss_head_fwd')
        ctx.save_for_backward(sums, la, lc)
        return out

    @staticmethod
    def coef_hint(lv_ial, lv_icl, n_anchors, z_ial, alpha_ial, zoom):
        """dL/d(terms) of the standard composition `loss_dict['loss']` with upstream gradient 1 -- it depends on the two log_vars vectors
        and constants only, never on the term values, so it is known BEFORE the terms are (FusedContrastiveFn one-pass mode).  float32
        [3M+1] on the device; no autograd."""
        M = int(lv_ial.numel())
        la, lc = _req(lv_ial.detach(), 'log_vars (ial)'), _req(lv_icl.detach(), 'log_vars (icl)')
        dev = la.device
        key = (str(dev), M)
        cst = LossHeadFn._hint_const.get(key)
        if cst is None:          # gout = (1, 0, 0, 0) and a dummy terms vector (the kernel reads it for the log_vars gradients only)
            cst = LossHeadFn._hint_const[key] = (torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev, dtype=torch.float64),
                                                 torch.zeros((3 * M + 1,), device=dev, dtype=torch.float32))
        inv_aa = 1.0 / float(n_anchors * n_anchors) if n_anchors else float('nan')
        d = torch.empty((3 * M + 1,), device=dev, dtype=torch.float32)
        junk = torch.empty((2, M), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().sga_loss_head_bwd(_p(cst[0]), _p(cst[1]), 0, _p(la), _p(lc), M, inv_aa, float(z_ial), float(alpha_ial), float(zoom),
                                                _p(d), _p(junk[0]), _p(junk[1]), _stream()), 'sga_loss_head_bwd')
        return d

    _hint_const = {}

    @staticmethod
    def backward(ctx, gout):
        sums, la, lc = ctx.saved_tensors
        M = ctx.consts[0]
        gout = gout.to(torch.float64).contiguous()
        dsums = torch.empty_like(sums)
        dla, dlc = torch.empty_like(la), torch.empty_like(lc)
        _lib.check(_lib.lib().sga_loss_head_bwd(_p(gout), _p(sums), ctx.f64, _p(la), _p(lc), *ctx.consts, _p(dsums), _p(dla), _p(dlc),
                                                _stream()), 'sga_loss_head_bwd')
        return dsums, dla, dlc, None, None, None, None


# ------------------------------------------------------------------------------------------ loss_group = b
class LossGroups:
    """Partition of a batch's pairs into groups of `b` consecutive pairs (the reference's training batches,
    configs/scan3r/scan3r_ground_truth.yaml:27): per group the contiguous ranges it occupies in the packed
    anchor / N1 / N2 row blocks, and the offsets of its similarity blocks.  Built from the per-pair counts the
    collate provides (scan3r.py:142-173: e1i_count / e1j_count / e2j_count)."""

    def __init__(self, data_dict, b, device):
        ca = _np.asarray(data_dict['e1i_count']).reshape(-1).astype(_np.int64)
        c1 = _np.asarray(data_dict['e1j_count']).reshape(-1).astype(_np.int64)
        c2 = _np.asarray(data_dict['e2j_count']).reshape(-1).astype(_np.int64)
        if not (len(ca) == len(c1) == len(c2)):
            raise RuntimeError('sgaligner_amd: e1i_count / e1j_count / e2j_count disagree')
        if int(b) < 1:
            raise RuntimeError(f'sgaligner_amd: loss_group must be a positive number of pairs (got {b})')
        B = len(ca)
        cuts = list(range(0, B, int(b))) + [B]
        oa, o1, o2 = (_np.concatenate([[0], _np.cumsum(c)]) for c in (ca, c1, c2))
        g = _np.zeros((len(cuts) - 1, 8), dtype=_np.int32)
        for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            g[k, :6] = (oa[lo], oa[hi] - oa[lo], o1[lo], o1[hi] - o1[lo], o2[lo], o2[hi] - o2[lo])
        size = 2 * g[:, 1].astype(_np.int64) * (g[:, 1].astype(_np.int64) + g[:, 3] + g[:, 5])
        soff = _np.concatenate([[0], _np.cumsum(size)]).astype(_np.int64)
        self.G, self.b = int(g.shape[0]), int(b)
        self.host = g
        self.s_total = int(soff[-1])
        self.groups = torch.from_numpy(g).to(device)
        self.soff = torch.from_numpy(soff).to(device)
        self.na = torch.from_numpy(g[:, 1].astype(_np.float32)).to(device)
        self.totals = (int(oa[-1]), int(o1[-1]), int(o2[-1]))

    _cache = _SmallCache()

    @staticmethod
    def of(data_dict, b, device):
        device = torch.device(device)
        key = _fingerprint([_np.asarray(data_dict[k]) for k in ('e1i_count', 'e1j_count', 'e2j_count')], (str(device), int(b)))
        return LossGroups._cache.get(key, lambda: LossGroups(data_dict, b, device))


class GroupedContrastiveFn(torch.autograd.Function):
    """Raw loss terms of every loss group: out [G, NT + 2M] = [ICL_k sums | IALa_m | IALb_m] (NT = M+1 with the joint
    table derived from the M modality tables through beta; M == 1 -> ICL of the single table only, beta None).
    csrc/grouploss.hip: similarity blocks materialised per group (they are reference-sized), one set of launches."""

    @staticmethod
    def forward(ctx, index_sets, groups, alpha, beta, *tables):
        L = _lib.lib()
        M = len(tables)
        nt = M + 1 if M > 1 else 1
        no = nt + (2 * M if M > 1 else 0)
        tables = [_req(t.contiguous(), f'table[{i}]') for i, t in enumerate(tables)]
        dev = tables[0].device
        s, gr = index_sets, groups
        if (s.A, s.J1, s.J2) != gr.totals:
            raise RuntimeError('sgaligner_amd: per-pair counts (e1i_count/e1j_count/e2j_count) do not add up to the index sets')
        st = _stream()
        dp = 104
        T = tables[0].shape[0]
        zs, nrms = [], []
        poison = torch.zeros((1,), device=dev, dtype=torch.float32)
        for e in tables:
            d = e.shape[1]
            if d > dp:
                raise RuntimeError('sgaligner_amd: loss_group needs emb_dim <= 104')
            z = torch.empty((max(s.R, 1), dp), device=dev, dtype=torch.float32)
            nrm = torch.empty((max(s.R, 1),), device=dev, dtype=torch.float32)
            _lib.check(L.sga_loss_gather(_p(e), T, d, _p(s.idx), s.R, _p(z), dp, _p(nrm), st), 'sga_loss_gather')
            _lib.check(L.sga_loss_check_norms(_p(nrm), s.R, _p(poison), st), 'sga_loss_check_norms')
            zs.append(z); nrms.append(nrm)
        if beta is not None:
            beta = _req(beta.contiguous(), 'beta')
        S = torch.empty((M * max(gr.s_total, 1),), device=dev, dtype=torch.float32)
        sums = torch.empty((max(gr.G, 1), nt, 8), device=dev, dtype=torch.float64)
        out = torch.zeros((max(gr.G, 1), no), device=dev, dtype=torch.float64)
        _lib.check(L.sga_group_loss_fwd(_ptr_array(zs), M, _p(beta), s.A, s.J1, _p(gr.groups), gr.G, _p(gr.soff), gr.s_total,
                                        float(alpha), TAU_ICL, TAU_IAL, _p(S), _p(sums), _p(out), int(GROUP_LOSS_VALU), st), 'sga_group_loss_fwd')
        ctx.s, ctx.gr, ctx.alpha, ctx.M = s, gr, float(alpha), M
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.has_beta = beta is not None
        ctx.save_for_backward(S, sums, *( [beta] if beta is not None else []), *zs, *nrms)
        return out[:gr.G].float() + poison

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        s, gr, M = ctx.s, ctx.gr, ctx.M
        saved = list(ctx.saved_tensors)
        S, sums = saved[0], saved[1]
        beta = saved[2] if ctx.has_beta else None
        rest = saved[3:] if ctx.has_beta else saved[2:]
        zs, nrms = rest[:M], rest[M:]
        dev = S.device
        st = _stream()
        dp = 104
        coef = gout.contiguous().float()
        C = S.clone()                         # the backward overwrites the similarity blocks: keep the saved ones (retain_graph)
        dzs = [torch.zeros((max(s.R, 1), dp), device=dev, dtype=torch.float32) for _ in range(M)]
        gamma = torch.zeros((max(gr.G, 1), M), device=dev, dtype=torch.float64)
        _lib.check(L.sga_group_loss_bwd(_ptr_array(zs), M, _p(beta), s.A, s.J1, _p(gr.groups), gr.G, _p(gr.soff), gr.s_total,
                                        ctx.alpha, TAU_ICL, TAU_IAL, _p(C), _p(sums), _p(coef), _ptr_array(dzs), _p(gamma), int(GROUP_LOSS_VALU), st),
                   'sga_group_loss_bwd')
        grads = []
        for k in range(M):
            t, d = ctx.shapes[k]
            de = torch.zeros((t, d), device=dev, dtype=torch.float32)
            _lib.check(L.sga_loss_scatter(_p(dzs[k]), _p(zs[k]), _p(nrms[k]), _p(s.idx), s.R, d, dp, _p(de), st), 'sga_loss_scatter')
            grads.append(de)
        gbeta = gamma[:gr.G].sum(0).float() if ctx.has_beta else None
        return (None, None, None, gbeta, *grads)


def grouped_contrastive_terms(tables, fusion_weight, data_dict, b, alpha=ALPHA):
    """tables: the M modality tables (M >= 2: the joint is their fusion with `fusion_weight` [M,1]; M == 1: pass None).
    Returns (out [G, NT+2M], LossGroups)."""
    dev = tables[0].device
    s = IndexSets.of(data_dict, dev, int(tables[0].shape[0]))
    gr = LossGroups.of(data_dict, b, dev)
    beta = None
    if len(tables) > 1:
        w = torch.softmax(fusion_weight.reshape(-1), dim=0)            # sg_aligner.py:32
        beta = (w * w) / (w * w).sum()
    return GroupedContrastiveFn.apply(s, gr, alpha, beta, *tables), gr


def group_data_dicts(data_dict, b):
    """The index sets of every loss group as stand-alone dicts (global object indices kept): what the reference's loss
    would be handed for that group.  Used by the general (arbitrary joint table) loss_group path and by the tests."""
    ca, c1, c2 = (_np.asarray(data_dict[k]).reshape(-1) for k in ('e1i_count', 'e1j_count', 'e2j_count'))
    oa, o1, o2 = (_np.concatenate([[0], _np.cumsum(c)]) for c in (ca, c1, c2))
    B = len(ca)
    out = []
    for lo in range(0, B, int(b)):
        hi = min(B, lo + int(b))
        out.append({'e1i': _np.asarray(data_dict['e1i'])[oa[lo]:oa[hi]], 'e2i': _np.asarray(data_dict['e2i'])[oa[lo]:oa[hi]],
                    'e1j': _np.asarray(data_dict['e1j'])[o1[lo]:o1[hi]], 'e2j': _np.asarray(data_dict['e2j'])[o2[lo]:o2[hi]]})
    return out


# ------------------------------------------------------------------------------------------ GAT
class GraphBatch:
    """Device-side CSR-style description of the 2B scene graphs of a batch: node/edge offsets in the
    src,ref,src,ref... order of reference sg_aligner.py:86-110, plus the int64 [sum E, 2] edge list
    (graph-local node ids, column 0 = source j, column 1 = target i) exactly as collated."""

    def __init__(self, node_counts, edge_counts, edges, keep_edges=True):
        nc = _np.asarray(node_counts, dtype=_np.int64).reshape(-1)
        ec = _np.asarray(edge_counts, dtype=_np.int64).reshape(-1)
        if nc.shape != ec.shape:
            raise RuntimeError('sgaligner_amd: graph_per_obj_count and graph_per_edge_count disagree')
        self.G = int(nc.shape[0])
        self.nmax = int(nc.max()) if self.G else 0
        self.T = int(nc.sum())
        self.E = int(ec.sum())
        dev = edges.device
        if edges.dtype != torch.int64:
            edges = edges.to(torch.int64)
        edges = edges.contiguous()
        if edges.shape[0] < self.E:
            raise RuntimeError('sgaligner_amd: edge list shorter than graph_per_edge_count says')
        if VALIDATE and self.E and edges.is_cuda:
            # Node ids are graph-LOCAL (scan3r.py:99): anything outside [0, largest graph) can not be a node of any graph.
            # The kernels drop out-of-range endpoints (PyG would raise an index error); catch the gross case here, once per batch.
            DEFERRED_CHECKS.poll()                                   # earlier batches' answers (no waiting)
            DEFERRED_CHECKS.submit(torch.stack(torch.aminmax(edges[:self.E])), self.nmax,
                                   'sgaligner_amd: edge endpoints span [%d, %d] but the largest graph has %d nodes '
                                   '(edges must hold graph-local node ids)')
        self.edges = edges if keep_edges else None
        offs = _h2d(_np.concatenate([[0], _np.cumsum(nc), [0], _np.cumsum(ec)]).astype(_np.int32), dev)     # one upload
        self.node_off, self.edge_off = offs[:self.G + 1], offs[self.G + 1:]
        self.complete = self._complete_flags() if keep_edges else None

    def _complete_flags(self):
        """uint8 [G]: 1 = the graph is COMPLETE (every ordered pair once, nothing else); the attention kernels then never read its edge list.
        Recomputed from the edge tensor's CONTENT on every call (one streaming pass; a caller may refill the same device buffer)."""
        if not GAT_COMPLETE_FAST_PATH or self.G == 0 or self.edges is None or not self.edges.is_cuda:
            return None
        flags = torch.empty((self.G,), device=self.edges.device, dtype=torch.uint8)
        _lib.check(_lib.lib().sga_gat_complete_flags(_p(self.edges), _p(self.node_off), _p(self.edge_off), self.G, _p(flags), _stream()),
                   'sga_gat_complete_flags')
        return flags

    _cache = _SmallCache(2)

    @staticmethod
    def of(data_dict):
        """Offsets are cached by the CONTENT of the two host count arrays + the identity of the device edge list (nothing is
        stored in the caller's dict).  The cached object keeps only the small offset arrays, never the edge tensor."""
        edges = data_dict['edges']
        if VALIDATE:
            DEFERRED_CHECKS.poll()                                   # earlier batches' answers (no waiting): a bad batch raises here
        key = _fingerprint([_np.asarray(data_dict['graph_per_obj_count']), _np.asarray(data_dict['graph_per_edge_count'])],
                           (str(edges.device), edges.data_ptr(), tuple(edges.shape), str(edges.dtype)))
        proto = GraphBatch._cache.get(key, lambda: GraphBatch(data_dict['graph_per_obj_count'], data_dict['graph_per_edge_count'],
                                                              edges, keep_edges=False))
        gb = GraphBatch.__new__(GraphBatch)
        gb.__dict__.update(proto.__dict__)
        gb.edges = edges if edges.dtype == torch.int64 and edges.is_contiguous() else edges.to(torch.int64).contiguous()
        gb.complete = gb._complete_flags()
        return gb


GAT_COMPLETE_FAST_PATH = _os.environ.get('SGA_GAT_COMPLETE', '1') != '0'     # complete graphs (what the reference's preprocessing writes) skip the edge list in the attention kernels
def _gat_status_verdict(v):
    if v[0] == 0:
        return None
    return ('sgaligner_amd: a (source, target) edge occurs more than 255 times in one graph of the PREVIOUS batch; the GAT kernels count '
            'duplicate edges in 8 bits (PyG would count them all), so that step\'s structure embeddings were not PyG-equivalent -- '
            'deduplicate the edge list')


def _attn_fwd(h, att_s, att_d, bias, gb, check_status=False):
    out = torch.empty_like(h)
    st = None
    if check_status and VALIDATE:
        # a FRESH status word per batch: a sticky shared one re-raised for clean batches whose read-back was enqueued before its reset
        st = torch.zeros((1,), device=h.device, dtype=torch.int32)
    ev = _ev_start()
    _lib.check(_lib.lib().sga_gat_attn_fwd(_p(h), _p(att_s), _p(att_d), _p(bias), _p(gb.edges), _p(gb.node_off),
                                           _p(gb.edge_off), gb.G, gb.nmax, _p(out), _p(st), _p(getattr(gb, 'complete', None)), _stream()), 'sga_gat_attn_fwd')
    _ev_stop(ev, 'gat_attn_fwd', (int(h.shape[0]), int(gb.edges.shape[0]), getattr(gb, 'complete', None) is not None))
    if st is not None:            # read back without blocking; raises at the next batch's poll (or DEFERRED_CHECKS.flush())
        DEFERRED_CHECKS.submit_fn(st, _gat_status_verdict)
    return out


def _attn_bwd(h, d_o, att_s, att_d, gb):
    dh = torch.empty_like(h)
    dboth = torch.empty((2,) + tuple(att_s.shape), device=att_s.device, dtype=att_s.dtype)      # adjacent: zeroed in one launch
    das, dad = dboth[0], dboth[1]
    ev = _ev_start()
    _lib.check(_lib.lib().sga_gat_attn_bwd(_p(h), _p(d_o), _p(att_s), _p(att_d), _p(gb.edges), _p(gb.node_off),
                                           _p(gb.edge_off), gb.G, gb.nmax, _p(dh), _p(das), _p(dad), _p(getattr(gb, 'complete', None)), _stream()),
               'sga_gat_attn_bwd')
    _ev_stop(ev, 'gat_attn_bwd', (int(h.shape[0]), int(gb.edges.shape[0]), getattr(gb, 'complete', None) is not None))
    return dh, das, dad


def _elu(x):
    y = torch.empty_like(x)
    _lib.check(_lib.lib().sga_elu_fwd(_p(x), _p(y), x.numel(), _stream()), 'sga_elu_fwd')
    return y


class MultiGATFn(torch.autograd.Function):
    """MultiGAT.forward over ALL graphs of a batch (reference gat.py:40-48 x sg_aligner.py:86-110):
    GATConv(3->128, h=2), ELU, GATConv(256->128, h=2)."""

    @staticmethod
    def forward(ctx, gb, x, w0, as0, ad0, b0, w1, as1, ad1, b1):
        if not x.is_cuda:
            raise RuntimeError('sgaligner_amd.MultiGATFn: HIP device tensor required; there is no CPU path')
        if x.dtype not in (torch.float32, torch.float64):
            raise RuntimeError(f'sgaligner_amd.MultiGATFn: tot_rel_pose must be float32 or float64, got {x.dtype}')
        x32 = cast_f32(x.contiguous())
        ps = [_req(t.contiguous(), n) for t, n in ((w0, 'gat0.lin'), (as0.reshape(-1), 'gat0.att_src'), (ad0.reshape(-1), 'gat0.att_dst'),
                                                   (b0, 'gat0.bias'), (w1, 'gat1.lin'), (as1.reshape(-1), 'gat1.att_src'),
                                                   (ad1.reshape(-1), 'gat1.att_dst'), (b1, 'gat1.bias'))]
        w0, as0f, ad0f, b0, w1, as1f, ad1f, b1 = ps
        t = x32.shape[0]
        if t != gb.T:
            raise RuntimeError(f'sgaligner_amd: tot_rel_pose has {t} rows but the graphs hold {gb.T} nodes')
        if w0.shape[0] != 256 or w1.shape != (256, 256):
            raise RuntimeError('sgaligner_amd: the HIP GAT path implements hidden_units=[F,128,128], heads=[2,2]')
        h0 = gemm(x32, w0, False, True, t, 256, x32.shape[1])
        o0 = _attn_fwd(h0, as0f, ad0f, b0, gb, check_status=True)      # both layers see the same edge list: one check per batch
        x1 = _elu(o0)
        h1 = gemm(x1, w1, False, True, t, 256, 256)
        o1 = _attn_fwd(h1, as1f, ad1f, b1, gb)
        ctx.gb = gb
        ctx.att_shapes = (tuple(as0.shape), tuple(as1.shape))
        ctx.save_for_backward(x32, h0, o0, x1, h1, w0, as0f, ad0f, w1, as1f, ad1f)
        return o1

    @staticmethod
    def backward(ctx, d_o1):
        x32, h0, o0, x1, h1, w0, as0, ad0, w1, as1, ad1 = ctx.saved_tensors
        gb = ctx.gb
        t = x32.shape[0]
        d_o1 = d_o1.contiguous()
        dh1, das1, dad1 = _attn_bwd(h1, d_o1, as1, ad1, gb)
        db1 = colsum(d_o1)
        dw1 = gemm(dh1, x1, True, False, 256, 256, t)
        dx1 = gemm(dh1, w1, False, False, t, 256, 256)
        d_o0 = torch.empty_like(o0)
        _lib.check(_lib.lib().sga_elu_bwd(_p(o0), _p(dx1), _p(d_o0), o0.numel(), _stream()), 'sga_elu_bwd')
        dh0, das0, dad0 = _attn_bwd(h0, d_o0, as0, ad0, gb)
        db0 = colsum(d_o0)
        dw0 = gemm(dh0, x32, True, False, 256, x32.shape[1], t)
        s0, s1 = ctx.att_shapes
        return (None, None, dw0, das0.reshape(s0), dad0.reshape(s0), db0, dw1, das1.reshape(s1), dad1.reshape(s1), db1)


def multi_gat(gb, x, layer0, layer1):
    """layer = (lin_weight, att_src, att_dst, bias)."""
    return MultiGATFn.apply(gb, x, *layer0, *layer1)


# ------------------------------------------------------------------------------------------ similarity + ranking
class PairLayout:
    """Device offsets of the pairs of a batch for the similarity kernels.  Cached by content (small host arrays)."""

    def __init__(self, pair_counts, device):
        pc = _np.asarray(pair_counts, dtype=_np.int64).reshape(-1)
        self.B = int(len(pc))
        self.nmax = int(pc.max()) if self.B else 0
        self.T = int(pc.sum())
        self.off_host = _np.concatenate([[0], _np.cumsum(pc)])
        self.pair_off = torch.from_numpy(self.off_host.astype(_np.int32)).to(device)

    _cache = _SmallCache()

    @staticmethod
    def of(pair_counts, device):
        device = torch.device(device)
        return PairLayout._cache.get(_fingerprint([_np.asarray(pair_counts)], (str(device),)), lambda: PairLayout(pair_counts, device))


class QueryBlocks:
    """The (pair, 64-row block) list of the blocks that hold a query object, + the device copies of the query arrays: the
    similarity kernel launches one workgroup per listed block (nothing for blocks without a query, work spread over all XCDs)."""

    def __init__(self, layout, q_idx, q_tgt, device):
        qi = _np.ascontiguousarray(_np.asarray(q_idx, dtype=_np.int32).reshape(-1))
        if VALIDATE and qi.size and (int(qi.min()) < 0 or int(qi.max()) >= layout.T):
            raise RuntimeError(f'sgaligner_amd: query object indices must lie in [0, {layout.T})')
        pair = _np.searchsorted(layout.off_host, qi, side='right') - 1
        rb = (qi - layout.off_host[pair]) // 64
        nrb = (layout.nmax + 63) // 64 if layout.nmax else 1
        key = _np.unique(pair.astype(_np.int64) * nrb + rb)
        self.n_blocks = int(key.size)
        self.blk_pair = torch.from_numpy((key // nrb).astype(_np.int32)).to(device)
        self.blk_row = torch.from_numpy((key % nrb).astype(_np.int32)).to(device)
        self.q_idx = torch.from_numpy(qi).to(device)
        self.q_tgt = None if q_tgt is None else torch.from_numpy(_np.ascontiguousarray(_np.asarray(q_tgt, dtype=_np.int32).reshape(-1))).to(device)
        self.Q = int(qi.size)

    _cache = _SmallCache()

    @staticmethod
    def of(layout, pair_counts, q_idx, q_tgt, device):
        arrs = [_np.asarray(pair_counts), _np.asarray(q_idx)] + ([_np.asarray(q_tgt)] if q_tgt is not None else [])
        return QueryBlocks._cache.get(_fingerprint(arrs, (str(device), q_tgt is None)), lambda: QueryBlocks(layout, q_idx, q_tgt, device))


SIMRANK_F16 = False      # opt-in: fp16-input MFMA similarity (BASELINE.json configs[4]); the default is exact fp32 MFMA


def simrank(emb, pair_counts, q_idx, q_tgt, k: int, f16=None):
    """For each query object: rank of its target and the k nearest other objects of its pair.
    emb [T,D] fp32 (un-normalised: the kernel applies emb/||emb|| as inference_align_reg.py:126 does);
    pair_counts [B] objects per pair; q_idx / q_tgt host int arrays of global object indices (q_tgt may be None; each object
    may be queried once).
    Returns (rank [Q] int32, topk_idx [Q,k] int32 pair-local, topk_sim [Q,k] fp32, layout) on the device."""
    emb = _req(emb.contiguous(), 'embedding')
    dev = emb.device
    T, D = emb.shape
    if isinstance(q_idx, torch.Tensor):
        q_idx = q_idx.cpu().numpy()
    if isinstance(q_tgt, torch.Tensor):
        q_tgt = q_tgt.cpu().numpy()
    lay = PairLayout.of(pair_counts, dev)
    if lay.T != T:
        raise RuntimeError(f'sgaligner_amd: the pairs hold {lay.T} objects but the embedding table has {T} rows')
    # The kernel keeps ONE query slot per object.  An object queried several times (never produced by the reference's collate,
    # but legal for a caller of this function) is served in rounds of distinct objects and the rows are stitched back.
    qi_h = _np.asarray(q_idx).reshape(-1)
    if qi_h.size > 1:
        order = _np.argsort(qi_h, kind='stable')
        srt = qi_h[order]
        if (srt[1:] == srt[:-1]).any():
            occ = _np.zeros(qi_h.size, dtype=_np.int64)          # occurrence number of every query among those of its object
            run_start = _np.concatenate([[0], _np.flatnonzero(srt[1:] != srt[:-1]) + 1])
            occ[order] = _np.arange(qi_h.size) - _np.repeat(run_start, _np.diff(_np.concatenate([run_start, [qi_h.size]])))
            qt_h = None if q_tgt is None else _np.asarray(q_tgt).reshape(-1)
            rank = torch.empty((qi_h.size,), device=dev, dtype=torch.int32)
            tk = torch.empty((qi_h.size, k), device=dev, dtype=torch.int32)
            ts = torch.empty((qi_h.size, k), device=dev, dtype=torch.float32)
            for r in range(int(occ.max()) + 1):
                sel = _np.flatnonzero(occ == r)
                rr, kk, ss, _ = simrank(emb, pair_counts, qi_h[sel], None if qt_h is None else qt_h[sel], k, f16)
                sel_d = torch.from_numpy(sel).to(dev)
                rank[sel_d], tk[sel_d], ts[sel_d] = rr, kk, ss
            return rank, tk, ts, lay
    qb = QueryBlocks.of(lay, pair_counts, q_idx, q_tgt, dev)
    Q = qb.Q
    rank = torch.empty((max(Q, 1),), device=dev, dtype=torch.int32)
    tk = torch.empty((max(Q, 1), max(k, 1)), device=dev, dtype=torch.int32)
    ts = torch.empty((max(Q, 1), max(k, 1)), device=dev, dtype=torch.float32)
    use16 = (SIMRANK_F16 or get_mfma_mode() == 'f16') if f16 is None else bool(f16)
    nb = _lib.lib().sga_simrank_workspace_bytes_f16(T, D) if use16 else _lib.lib().sga_simrank_workspace_bytes(T)
    ws = torch.empty((nb,), device=dev, dtype=torch.uint8)
    _lib.check(_lib.lib().sga_simrank(_p(emb), T, D, _p(lay.pair_off), _p(qb.blk_pair), _p(qb.blk_row), qb.n_blocks, lay.B, lay.nmax,
                                      _p(qb.q_idx), _p(qb.q_tgt), Q, k, _p(rank), _p(tk), _p(ts), int(use16), _p(ws), nb, _stream()),
               'sga_simrank')
    return rank[:Q], tk[:Q, :k], ts[:Q, :k], lay


def pair_metrics(rank, topk_idx, topk_sim, q_tgt, layout, pair_q_counts):
    """Per-pair Hits@1..5 counts, #queries, sum of reciprocal ranks and SGAR('2','50','100') on the device: [B,12] fp32."""
    dev = rank.device
    qc = _np.asarray(pair_q_counts, dtype=_np.int64).reshape(-1)
    qoff = torch.from_numpy(_np.concatenate([[0], _np.cumsum(qc)]).astype(_np.int32)).to(dev)
    out = torch.zeros((max(layout.B, 1), 12), device=dev, dtype=torch.float32)
    qt = q_tgt if isinstance(q_tgt, torch.Tensor) else torch.from_numpy(_np.ascontiguousarray(_np.asarray(q_tgt, dtype=_np.int32))).to(dev)
    tki = topk_idx.contiguous()
    tks = topk_sim.contiguous()
    _lib.check(_lib.lib().sga_pair_metrics(_p(rank.contiguous()), _p(tki), _p(tks), int(tki.shape[1]), _p(qt), _p(layout.pair_off),
                                           _p(qoff), layout.B, _p(out), _stream()), 'sga_pair_metrics')
    return out[:layout.B]


def _allreduce_sum(t, group_reduce):
    """Sum a device tensor over the ranks that shard the anchors (identity on one GPU)."""
    if group_reduce is not None:
        group_reduce(t)
    return t


class FusedContrastiveFn(torch.autograd.Function):
    """Same outputs as ContrastiveTermsFn for tables (E_1..E_M, joint) when joint == MultiModalFusion(E_1..E_M):
    the joint similarities are derived from the modality tiles (S_J = sum_m beta_m S_m), so the 300-d table is
    never swept.  Inputs: beta [M] (= softmax(w)^2 / sum, differentiable), the M modality tables.

    Sharding (one process per GPU): `shard = (a_lo, a_hi)` is the anchor range this rank owns and `reduce` an in-place
    SUM all-reduce.  Each rank evaluates its shard's share of every global sum / loss term (all-reduced, so the
    returned values are the batch-global ones on every rank) and, in backward, its shard's share of dL/dE for ALL
    rows -- the caller sums those over ranks (dist.AllGatherRows with reduce_grad=True).  dL/dbeta is returned as
    this rank's share as well (the parameter-gradient all-reduce completes it)."""

    @staticmethod
    def forward(ctx, index_sets, alpha, shard, reduce, coef_hint, beta, *tables):
        L = _lib.lib()
        M = len(tables)
        nt = M + 1
        tables = [_req(t.contiguous(), f'table[{i}]') for i, t in enumerate(tables)]
        beta = _req(beta.contiguous(), 'beta')
        dev = tables[0].device
        s = index_sets
        a_lo, a_hi = (0, s.A) if shard is None else (int(shard[0]), int(shard[1]))     # shard = (a_lo, a_hi[, cuts, rank]): see _sym_jobs
        T = tables[0].shape[0]
        st = _stream()
        dp = 104
        zs, nrms = [], []
        poison = torch.zeros((1,), device=dev, dtype=torch.float32)
        for k, e in enumerate(tables):
            d = e.shape[1]
            if d > dp:
                raise RuntimeError('sgaligner_amd: the fused loss path needs emb_dim <= 104')
            z = torch.empty((s.R + 32, dp), device=dev, dtype=torch.float32)
            z[s.R:].zero_()
            nrm = torch.empty((s.R,), device=dev, dtype=torch.float32)
            _lib.check(L.sga_loss_gather(_p(e), T, d, _p(s.idx), s.R, _p(z), dp, _p(nrm), st), 'sga_loss_gather')
            _lib.check(L.sga_loss_check_norms(_p(nrm), s.R, _p(poison), st), 'sga_loss_check_norms')
            zs.append(z); nrms.append(nrm)
        zarr = _ptr_array(zs)
        slots = 1 + L.sga_loss_slots()
        sums = torch.empty((slots, nt, 8), device=dev, dtype=torch.float64)
        dmax = max(e.shape[1] for e in tables)          # real width: the K step that only covers zero padding is skipped
        # opt-in split-bf16 x3 sweeps: the tables additionally as blocked bf16 hi/lo planes (sweepb.hip)
        zbs, zcs = [], []
        split3 = False
        split16 = M in (2, 3, 4) and dmax <= 100 and get_mfma_mode() in ('f16x2', 'f16x2p')      # columns 100, 101 of the planes carry the centring's bookkeeping
        if split16:
            nb = L.sga_loss_split16_bytes(s.A, s.J1, s.J2)
            for z in zs:
                zb = torch.empty((nb,), device=dev, dtype=torch.uint8)
                _lib.check(L.sga_loss_split16_tables(_p(z), s.A, s.J1, s.J2, _p(zb), st), 'sga_loss_split16_tables')
                zbs.append(zb)
            ev = None
            if KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(L.sga_loss_multi_sums_f16x2(_ptr_array(zbs), M, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums),
                                                   a_lo, a_hi, 1 if _f16x2_sums_lo(a_hi - a_lo, s.J1, s.J2) else 0, st), 'sga_loss_multi_sums_f16x2')
            if ev is not None:
                ev[1].record()
                KERNEL_EVENTS.setdefault('loss_multi_sums_f16x2', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M),))
        elif M in (2, 3, 4) and dmax <= 100 and FUSED_ANCHOR_BWD and get_mfma_mode() == 'bf16x6':
            # three exact bf16 planes per table (csrc/sweep3.hip): blocked h / m / l planes of the centred rows, once per step
            split3 = True
            nb = L.sga_loss_split3_bytes(s.A, s.J1, s.J2)
            for z in zs:
                zb = torch.empty((nb,), device=dev, dtype=torch.uint8)
                # + the anchor rows as fp32 z - zbar with a ones column: the stash products' B operand (gradient in two parts, see
                # sga_loss_scatter_tangent)
                zc = torch.empty((2 * s.A + 32, dp), device=dev, dtype=torch.float32)
                zc[2 * s.A:].zero_()
                _lib.check(L.sga_loss_split3_tables(_p(z), s.A, s.J1, s.J2, _p(zb), _p(zc), st), 'sga_loss_split3_tables')
                zbs.append(zb); zcs.append(zc)
            ev = None
            if KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(L.sga_loss_multi_sums_bf16x6(_ptr_array(zbs), M, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums),
                                                    a_lo, a_hi, st), 'sga_loss_multi_sums_bf16x6')
            if ev is not None:
                ev[1].record()
                KERNEL_EVENTS.setdefault('loss_multi_sums_bf16x6', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M),))
        elif M <= 3 and get_mfma_mode() == 'bf16x3':
            nb = L.sga_loss_split_bytes(s.A, s.J1, s.J2)
            for z in zs:
                zb = torch.empty((nb,), device=dev, dtype=torch.uint8)
                _lib.check(L.sga_loss_split_tables(_p(z), s.A, s.J1, s.J2, _p(zb), st), 'sga_loss_split_tables')
                zbs.append(zb)
            ev = None
            if KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(L.sga_loss_multi_sums_bf16x3(_ptr_array(zbs), M, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums),
                                                    a_lo, a_hi, st), 'sga_loss_multi_sums_bf16x3')
            if ev is not None:
                ev[1].record()
                KERNEL_EVENTS.setdefault('loss_multi_sums', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M),))
        else:
            ev = None
            if KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(L.sga_loss_multi_sums(zarr, M, dmax, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums), a_lo, a_hi, st),
                       'sga_loss_multi_sums')
            if ev is not None:
                ev[1].record()
                KERNEL_EVENTS.setdefault('loss_multi_sums', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M),))
        sums = _allreduce_sum(sums[0].contiguous(), reduce)
        zj = torch.empty((2 * s.A, M * dp), device=dev, dtype=torch.float32)
        _lib.check(L.sga_loss_build_joint(zarr, M, _p(beta), 2 * s.A, _p(zj), st), 'sga_loss_build_joint')
        out = torch.empty((slots * (nt + 2 * M),), device=dev, dtype=torch.float64)
        dps = [dp] * M + [M * dp]
        onepass = coef_hint is not None and M <= 4 and FUSED_ANCHOR_BWD and FUSED_AA_ONEPASS and s.A >= ONEPASS_MIN_ANCHORS
        extra = []
        if onepass:
            # ONE pass over the anchors x anchors similarities: dL/d(terms) is known (coef_hint), so the backward kernel runs now, block
            # by block on the bounded stash, and returns the term values of its rows as well; backward() starts from the saved A x A
            # gradients and only has the negatives' sweep left.
            coef = _req(coef_hint.contiguous(), 'coef_hint')
            n_terms = nt + 2 * M
            dz_all = torch.zeros((M, s.R, dp), device=dev, dtype=torch.float32)
            zz = torch.zeros((n_terms + nt * 8 + M,), device=dev, dtype=torch.float64)      # terms | gs | gamma: one fill
            out_acc, gs_aa, gam_aa = zz[:n_terms], zz[n_terms:n_terms + nt * 8].view(nt, 8), zz[n_terms + nt * 8:]
            # symmetric walk: one GPU, or every rank of an anchor-sharded job when the caller passed all ranks' cuts (on 32-row boundaries)
            cuts, crank = (shard[2], shard[3]) if (shard is not None and len(shard) >= 4) else (([0, s.A], 0) if (a_lo == 0 and a_hi == s.A) else (None, 0))
            sym = AA_SYMMETRIC and M <= AA_SYMMETRIC_MAX_M and cuts is not None and all(c % 32 == 0 for c in cuts[:-1]) and cuts[-1] == s.A \
                and cuts[crank] == a_lo and cuts[crank + 1] == a_hi
            jobs = _sym_jobs(list(cuts), crank, M) if sym else []
            chunks = jobs if sym else _anchor_chunks(a_lo, a_hi, s.A, M)
            if sym and len(cuts) == 2 and len(jobs) < 2:             # one block = one diagonal square: nothing to mirror, the ordered kernel (unmasked interior) does it
                sym, chunks = False, _anchor_chunks(a_lo, a_hi, s.A, M)
            if chunks:
                gsc = torch.empty((slots + 1, nt, 8), device=dev, dtype=torch.float64)
                gam2 = torch.empty((slots, M), device=dev, dtype=torch.float64)
            if chunks and sym:
                # every unordered anchor pair once over all ranks: a launch also evaluates the mirrored elements from column `mir` on (second stash)
                fl = max(((jh - jl) + max(0, jh - mir)) * (hi - lo) for lo, hi, jl, jh, mir in chunks)
                buf = [torch.empty((fl,), device=dev, dtype=torch.float32) for _ in range(M)]
                # 'f16x2': the stash products on split-fp16 MFMA (csrc/stashh.hip) -- transposed fp16 hi / lo planes of every table's unit rows,
                # once per step; each launch's stash scaled by its largest |coefficient|
                split_st = F16X2_STASH and dp == 104 and dmax <= 100 and get_mfma_mode() in ('f16x2', 'f16x2p')     # (columns 100, 101 of the planes carry the centring's bookkeeping)
                if split_st:
                    planes = [torch.empty((int(L.sga_loss_stash_planes_bytes(s.A)),), device=dev, dtype=torch.uint8) for _ in range(M)]
                    for k in range(M):
                        _lib.check(L.sga_loss_stash_planes(_p(zs[k]), s.A, dp, _p(planes[k]), st), 'sga_loss_stash_planes')
                # 'f16x2': the A x A similarities on fp16 MFMA from fp16 hi + lo rows (same bytes per row; the epilogue is the exact-fp32 code)
                h16 = F16X2_AA and dp == 104 and dmax <= 100 and get_mfma_mode() in ('f16x2', 'f16x2p')      # wider tables (emb_dim 101..104): the fp32 kernel
                if h16:
                    zh = [torch.empty((2 * s.A + 1, dp), device=dev, dtype=torch.float32) for _ in range(M)]
                    for k in range(M):
                        _lib.check(L.sga_loss_aa_planes(_p(zs[k]), 2 * s.A, _p(zh[k]), st), 'sga_loss_aa_planes')
                    zharr = _ptr_array(zh)
                aa_fn = L.sga_loss_anchor_multi_bwd_symx_h16 if h16 else L.sga_loss_anchor_multi_bwd_symx
                for lo, hi, jl, jh, mir in chunks:
                    n1 = (jh - jl) * (hi - lo)
                    m1 = [b[:n1] for b in buf]
                    m2 = [b[n1:] for b in buf]
                    has2 = mir < jh
                    _lib.check(aa_fn(zharr if h16 else zarr, M, _p(beta), s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(coef),
                                     _ptr_array(m1), _ptr_array(m2) if has2 else (_ct.c_void_p * M)(), _p(gsc), _p(gam2),
                                     lo, hi, jl, jh, mir, _p(out), st), 'sga_loss_anchor_multi_bwd_symx')
                    out_acc += out[:n_terms]
                    gs_aa += gsc[0]
                    gam_aa += gam2[0]
                    on8 = (hi - lo) % 8 == 0 and lo % 8 == 0 and jl % 8 == 0 and (not has2 or mir % 8 == 0)      # (a ragged last block: exact fp32)
                    for k in range(M):
                        if split_st and on8:
                            cmx = torch.maximum(m1[k].abs().max(), m2[k][:(jh - mir) * (hi - lo)].abs().max()) if has2 else m1[k].abs().max()
                            _lib.check(L.sga_loss_stash_grad_symx_f16x2(_p(m1[k]), _p(m2[k]) if has2 else None, _p(planes[k]), cmx.data_ptr(), s.A,
                                                                        _p(dz_all[k]), lo, hi, jl, jh, mir, st), 'sga_loss_stash_grad_symx_f16x2')
                        elif split3 and BF16X6_STASH:
                            # the four stash products on the sweeps' three exact bf16 planes (csrc/sweep3.hip: stash3_kernel)
                            _lib.check(L.sga_loss_stash_grad_symx_bf16x6(_p(m1[k]), _p(m2[k]) if has2 else None, _p(zbs[k]), s.A, s.J1, s.J2,
                                                                         _p(dz_all[k]), lo, hi, jl, jh, mir, st), 'sga_loss_stash_grad_symx_bf16x6')
                        else:
                            _lib.check(L.sga_loss_stash_grad_symx(_p(m1[k]), _p(m2[k]) if has2 else None, _p(zcs[k] if split3 else zs[k]), s.A, dp,
                                                                  _p(dz_all[k]), lo, hi, jl, jh, mir, st), 'sga_loss_stash_grad_symx')
                del buf, m1, m2
                if split_st:
                    del planes
                if h16:
                    del zh, zharr
            elif chunks:
                cmax = max(hi - lo for lo, hi in chunks)
                m1 = [torch.empty((s.A * cmax,), device=dev, dtype=torch.float32) for _ in range(M)]
                for lo, hi in chunks:
                    _lib.check(L.sga_loss_anchor_multi_bwd(zarr, M, _p(beta), s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(coef),
                                                           _ptr_array(m1), _p(gsc), _p(gam2), lo, hi, _p(out), st), 'sga_loss_anchor_multi_bwd')
                    out_acc += out[:n_terms]
                    gs_aa += gsc[0]
                    gam_aa += gam2[0]
                    for k in range(M):
                        if split3 and BF16X6_STASH:
                            _lib.check(L.sga_loss_stash_grad_symx_bf16x6(_p(m1[k]), None, _p(zbs[k]), s.A, s.J1, s.J2, _p(dz_all[k]), lo, hi, 0, s.A, s.A, st),
                                       'sga_loss_stash_grad_symx_bf16x6')
                        else:
                            _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zcs[k] if split3 else zs[k]), s.A, dp, _p(dz_all[k]), lo, hi, st), 'sga_loss_stash_grad')
                del m1
            out = _allreduce_sum(out_acc.clone(), reduce)
            extra = [dz_all, gs_aa.clone(), gam_aa.clone(), coef]
        else:
            if M <= 4 and FUSED_ANCHOR_FWD:      # joint similarities derived in registers, I block resident in LDS
                _lib.check(L.sga_loss_anchor_multi_fwd(zarr, M, _p(beta), s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(out),
                                                       a_lo, a_hi, st), 'sga_loss_anchor_multi_fwd')
            else:
                _lib.check(L.sga_loss_anchor_fwd(_ptr_array(zs + [zj]), (_ct.c_int * nt)(*dps), nt, s.A, _p(sums), float(alpha),
                                                 TAU_ICL, TAU_IAL, _p(out), a_lo, a_hi, st), 'sga_loss_anchor_fwd')
            out = _allreduce_sum(out[:nt + 2 * M].contiguous(), reduce)
        ctx.s, ctx.alpha, ctx.M, ctx.shard, ctx.reduce = s, float(alpha), M, (a_lo, a_hi), reduce
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.n_zb = len(zbs)
        ctx.split16 = split16
        ctx.split3 = split3
        ctx.onepass = onepass
        ctx.n_zc = len(zcs)
        ctx.save_for_backward(sums, beta, zj, *zs, *nrms, *zbs, *zcs, *extra)
        return out.float() + poison

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        s, M = ctx.s, ctx.M
        a_lo, a_hi = ctx.shard
        ns = a_hi - a_lo
        nt = M + 1
        sums, beta, zj, *rest = ctx.saved_tensors
        onepass_saved = None
        if ctx.onepass:
            rest, onepass_saved = rest[:-4], rest[-4:]
        zs, nrms, zbs = rest[:M], rest[M:2 * M], rest[2 * M:2 * M + ctx.n_zb]
        zcs = rest[2 * M + ctx.n_zb:]
        dev = sums.device
        st = _stream()
        dp = 104
        A = s.A
        coef = gout.contiguous().float()
        slots = 1 + L.sga_loss_slots()
        gam_neg = torch.empty((slots, M), device=dev, dtype=torch.float64)   # dL/dbeta via the negatives (zeroed by the callee)
        if onepass_saved is not None:
            # The A x A part was done in forward() for coef_hint; everything it produced is linear in dL/d(terms), so an upstream factor
            # (loss / k, loss * w) is applied here: u = <gout, hint> / <hint, hint>.  A gout that is NOT a multiple of the hint (a caller
            # who backpropagates one of the returned components alone, or re-weights them) cannot be served from the saved gradients:
            # that is detected on the device and raised at the next batch (deferred, no host sync) -- set ops.FUSED_AA_ONEPASS = False.
            dz_aa, gs_aa, gam_aa, hint = onepass_saved
            hh = torch.dot(hint, hint)
            u = torch.dot(coef, hint) / hh
            # Always (VALIDATE or not), on the device and without a host sync: a mismatching gradient POISONS what this node returns --
            # every table gradient, dL/d(sums) and dL/dbeta become NaN -- so the wrong gradients can never be consumed silently by an
            # optimiser step that runs before the deferred error below is polled.
            mismatch = (coef - u * hint).abs().max() > 1e-4 * hh.sqrt()
            u = torch.where(mismatch, torch.full_like(u, float('nan')), u)
            if VALIDATE:
                bad = mismatch.to(torch.int32).reshape(1)
                DEFERRED_CHECKS.submit_fn(bad, lambda v: None if v[0] == 0 else (
                    'sgaligner_amd: the gradient that reached the loss terms is not a multiple of the one OverallLoss announced at forward time '
                    "(backward through something other than loss_dict['loss'] up to a factor); set sgaligner_amd.ops.FUSED_AA_ONEPASS = False"))
            dz_all = dz_aa * u.float()                     # a fresh tensor: backward() may run twice on one graph (retain_graph)
            dzs = [dz_all[k] for k in range(M)]
            gs = _allreduce_sum(gs_aa * u, ctx.reduce)
            gam_anc = gam_aa * u
            chunks = []
        else:
            dz_all = torch.zeros((M, s.R, dp), device=dev, dtype=torch.float32)      # one fill for the M accumulation targets
            dzs = [dz_all[k] for k in range(M)]
            gam_anc = torch.zeros((M,), device=dev, dtype=torch.float64)         # ... via the anchors x anchors terms
        # The anchors x anchors backward runs one anchor-row block [lo, hi) at a time: the kernel writes the block's
        # transposed coefficient stash M1[m] [A, hi-lo], two GEMMs turn it into dX1 / dX2, the next block reuses the
        # buffers -- memory O(A*D + STASH_BYTES), never A x A (nothing of that size at configs[2]).
        fused = M <= 4 and FUSED_ANCHOR_BWD
        ntab = M if fused else nt
        if onepass_saved is None:
            chunks = _anchor_chunks(a_lo, a_hi, A, ntab)
            zz = torch.zeros((nt * 8 + M,), device=dev, dtype=torch.float64)              # gs and gam_acc: one fill
            gs = zz[:nt * 8].view(nt, 8)
        if onepass_saved is not None:
            pass                                                                       # nothing of the A x A part is left to do
        elif fused:
            gsc = torch.empty((slots + 1, nt, 8), device=dev, dtype=torch.float64)     # + one block: float copy of 1/(sums+eps)
            gam2 = torch.empty((slots, M), device=dev, dtype=torch.float64)
            gam_acc = zz[nt * 8:]
        else:
            gsc = torch.empty((slots, nt, 8), device=dev, dtype=torch.float64)
            dps = [dp] * M + [M * dp]
            dzj = torch.zeros((2 * A, M * dp), device=dev, dtype=torch.float32) if chunks else None
        if chunks:
            cmax = max(hi - lo for lo, hi in chunks)
            m1 = [torch.empty((A * cmax,), device=dev, dtype=torch.float32) for _ in range(ntab)]
        for lo, hi in chunks:
            if fused:
                # M1[m] already holds dL/dS_m + beta_m dL/dS_J, dL/dbeta comes out directly; no joint operand / stash
                _lib.check(L.sga_loss_anchor_multi_bwd(_ptr_array(zs), M, _p(beta), A, _p(sums), ctx.alpha, TAU_ICL, TAU_IAL, _p(coef),
                                                       _ptr_array(m1), _p(gsc), _p(gam2), lo, hi, None, st), 'sga_loss_anchor_multi_bwd')
                gam_acc += gam2[0]
            else:
                _lib.check(L.sga_loss_anchor_bwd(_ptr_array(list(zs) + [zj]), (_ct.c_int * nt)(*dps), nt, A, _p(sums), ctx.alpha,
                                                 TAU_ICL, TAU_IAL, _p(coef), _ptr_array(m1), _p(gsc), lo, hi, st), 'sga_loss_anchor_bwd')
                _lib.check(L.sga_loss_stash_grad(_p(m1[M]), _p(zj), A, M * dp, _p(dzj), lo, hi, st), 'sga_loss_stash_grad')
            gs += gsc[0]
            for k in range(M):
                if ctx.split3 and BF16X6_STASH:
                    _lib.check(L.sga_loss_stash_grad_symx_bf16x6(_p(m1[k]), None, _p(zbs[k]), A, s.J1, s.J2, _p(dzs[k]), lo, hi, 0, A, A, st),
                               'sga_loss_stash_grad_symx_bf16x6')
                else:
                    _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zcs[k] if ctx.split3 else zs[k]), A, dp, _p(dzs[k]), lo, hi, st), 'sga_loss_stash_grad')
        if chunks:
            del m1
            if fused:
                gam_anc = gam_acc
            else:
                gam_sq = torch.zeros((M,), device=dev, dtype=torch.float64)
                _lib.check(L.sga_loss_fold_joint(_ptr_array(zs), M, _p(beta), _p(dzj), 2 * A, _ptr_array(dzs), _p(gam_sq), st),
                           'sga_loss_fold_joint')
                gam_anc = gam_sq / (2.0 * torch.sqrt(beta.double()))     # through sqrt(beta_m) in the anchor rows of ZJ
                del dzj
        if onepass_saved is None:
            gs = _allreduce_sum(gs, ctx.reduce)                          # dL/d(global sums) needs every shard's tiles
        ev = None
        if KERNEL_EVENTS is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if ctx.n_zb and ctx.split3:           # the forward ran in bf16x6 mode: its blocked bf16 h / m / l planes are there
            _lib.check(L.sga_loss_multi_grad_bf16x6(_ptr_array(zbs), M, _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                                    _p(gam_neg), a_lo, a_hi, st), 'sga_loss_multi_grad_bf16x6')
        elif ctx.n_zb and ctx.split16:        # the forward ran in f16x2 mode: its blocked fp16 hi/lo planes are there
            _lib.check(L.sga_loss_multi_grad_f16x2(_ptr_array(zbs), M, _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                                   _p(gam_neg), a_lo, a_hi, 1 if _f16x2_coef_lo(ns, s.J1, s.J2) else 0, st), 'sga_loss_multi_grad_f16x2')
        elif ctx.n_zb:          # the forward ran in bf16x3 mode: its blocked bf16 planes are there
            _lib.check(L.sga_loss_multi_grad_bf16x3(_ptr_array(zbs), M, _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                                    _p(gam_neg), a_lo, a_hi, st), 'sga_loss_multi_grad_bf16x3')
        else:
            _lib.check(L.sga_loss_multi_grad(_ptr_array(zs), M, max(d for _, d in ctx.shapes), _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                             _p(gam_neg), a_lo, a_hi, st), 'sga_loss_multi_grad')
        if ev is not None:
            ev[1].record()
            KERNEL_EVENTS.setdefault('loss_multi_grad_bf16x6' if (ctx.n_zb and ctx.split3) else 'loss_multi_grad_f16x2' if (ctx.n_zb and ctx.split16) else 'loss_multi_grad', []).append(ev + ((ns, A, s.J1, s.J2, M),))
        grads = []
        same = all(sh == ctx.shapes[0] for sh in ctx.shapes)
        de_all = torch.zeros((M,) + tuple(ctx.shapes[0]), device=dev, dtype=torch.float32) if same else None   # one fill
        for k in range(M):
            t, d = ctx.shapes[k]
            de = de_all[k] if same else torch.zeros((t, d), device=dev, dtype=torch.float32)
            if ctx.n_zb and ctx.split3:       # the gradient is in two parts (sum c (z - zbar) | sum c): projected without forming their sum
                _lib.check(L.sga_loss_scatter_tangent(_p(dzs[k]), _p(zs[k]), _p(nrms[k]), _p(s.idx), A, s.J1, s.J2, d, _p(zbs[k]), _p(de), st),
                           'sga_loss_scatter_tangent')
            else:
                _lib.check(L.sga_loss_scatter(_p(dzs[k]), _p(zs[k]), _p(nrms[k]), _p(s.idx), s.R, d, dp, _p(de), st), 'sga_loss_scatter')
            grads.append(de)
        # d/dbeta_m: through the negatives (gamma) + through sqrt(beta_m) in the anchor rows of ZJ
        gbeta = (gam_neg[0] + gam_anc).float()
        return (None, None, None, None, None, gbeta, *grads)


def fused_contrastive_terms(tables, fusion_weight, data_dict, alpha=ALPHA, shard=None, reduce=None, coef_hint=None):
    """tables: the M modality tables the joint table was fused from; fusion_weight: the [M,1] parameter.
    shard / reduce: see FusedContrastiveFn (anchor range owned by this rank, in-place SUM all-reduce).
    coef_hint (optional, float32 [3M+1], no grad): dL/d(terms) as the caller's loss head will deliver it (LossHeadFn.coef_hint).  With it,
    and gradients enabled, the anchors x anchors similarities are computed ONCE -- terms and their gradients from the same launches in
    forward(), backward() only scales them by the upstream factor -- instead of once per direction."""
    s = IndexSets.of(data_dict, tables[0].device, int(tables[0].shape[0]))
    w = torch.softmax(fusion_weight.reshape(-1), dim=0)                # sg_aligner.py:32
    beta = (w * w) / (w * w).sum()
    if coef_hint is not None and not (torch.is_grad_enabled() and (beta.requires_grad or any(t.requires_grad for t in tables))):
        coef_hint = None
    return FusedContrastiveFn.apply(s, alpha, shard, reduce, coef_hint, beta, *tables), s
