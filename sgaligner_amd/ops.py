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

MFMA_MODES = ('f32', 'bf16x6', 'f16')
# Which arithmetic the MFMA kernels with more than one variant use is a POLICY OF THIS LAYER: the C-ABI library is stateless (every entry
# point's arithmetic is in its name or an explicit argument, include/sgaligner_hip.h).  One setting per process, like the reference's
# torch.backends flags; initial value from SGA_MFMA_MODE.
DEFAULT_MFMA_MODE = 'bf16x6'
_MODE = _os_mode.environ.get('SGA_MFMA_MODE', DEFAULT_MFMA_MODE)
if _MODE not in MFMA_MODES:
    raise ValueError(f"sgaligner_amd: SGA_MFMA_MODE must be one of {MFMA_MODES} (got {_MODE!r})")


def set_mfma_mode(mode: str) -> str:
    """Arithmetic of the MFMA kernels that have more than one variant.
    'f32': fp32 MFMA everywhere (v_mfma_f32_*_f32); the fused loss sweeps run over centred fp32 tables and deliver the gradient in the
    same two parts as the default (CENTRED_F32).
    'bf16x6' (THE DEFAULT): the fused 100-d loss sweeps (anchors x negatives: forward sums + gradient) with every fp32 operand split EXACTLY into three
    bf16 terms (8 + 8 + 8 significand bits, fp32's exponent range) and six bf16 MFMAs per product into one fp32 accumulator -- fp32
    arithmetic on the exact operands at 6/16 of the fp32 MFMA's matrix time (csrc/sweep3.hip; SURVEY 7 "fp32 MFMA or split-bf16 x3");
    everything else exact fp32.  M = 2, 3, 4 tables of emb_dim <= 100; other shapes take the 'f32' kernels.
    'f16' (opt-in, BASELINE.json configs[4]: loss tables WIDER than 128 columns and the similarity ranking take fp16 inputs with fp32
    accumulation -- csrc/wide16.hip, 1e-2 tolerance; 100-d tables and everything else as in the default mode).
    Returns the previous mode."""
    global _MODE
    if mode not in MFMA_MODES:
        raise ValueError(f"sgaligner_amd: mfma mode must be one of {sorted(MFMA_MODES)} (got {mode!r})")
    old, _MODE = _MODE, mode
    return old


def get_mfma_mode() -> str:
    return _MODE


# PointNet forward arithmetic per mode (sga_pointnet_fwd_ws `mode`): 0 exact fp32 (v_mfma_f32_32x32x2_f32), 4 three exact bf16 planes (six bf16 MFMAs
# per product: fp32 arithmetic on the bf16 matrix pipe, like the default loss sweeps).  SGA_POINTNET_P3=0: the fp32 kernel in every mode.
_P3 = 4 if _os_mode.environ.get('SGA_POINTNET_P3', '1') != '0' else 0
_POINTNET_MODE = {'f32': 0, 'bf16x6': _P3, 'f16': _P3}
GROUP_LOSS_VALU = _os_mode.environ.get('SGA_GROUP_GRAD_VALU', '0') == '1'      # loss_group kernels: the VALU forms (cross-checks) instead of MFMA


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
CENTRED_F32 = True        # 'f32' mode, fused joint path, emb_dim <= 100: fp32-MFMA sweeps over centred tables, gradient in two parts (tests flip it to compare)
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


# ---- run-time switches of the kernels whose autograd nodes live in pointnet_ops / loss_ops / gat_ops / rank_ops (read there as ops.<FLAG>)
POINTNET_SPLIT_MAX_OBJECTS = 1023      # the library uses the split form below 4 x CUs objects; above that a workspace is not allocated
# 'bf16x6' forward sums: all six partial products (False), or the h and m planes only (True: h h + h m + m h, 11 of 20 MFMAs, no l-plane traffic; every similarity
# with an unbiased 2^-16 rounding).  None (default) = lite only when the smallest of the four global sums has >= BF16X6_SUMS_LITE_MIN_TERMS terms:
# the roundings average out (relative 1e-4 per term / sqrt(terms), bias 1e-8) far below the fp32 rounding of the sums' own accumulation.
BF16X6_SUMS_LITE = {'1': True, '0': False}.get(_os.environ.get('SGA_BF16X6_SUMS_LITE', ''), None)
BF16X6_SUMS_LITE_MIN_TERMS = 1 << 24
BF16X6_STASH = _os.environ.get('SGA_BF16X6_STASH', '1') != '0'   # 'bf16x6': the A x A stash products on the sweeps' three exact bf16 planes (off: fp32-MFMA GEMMs)
GAT_COMPLETE_FAST_PATH = _os.environ.get('SGA_GAT_COMPLETE', '1') != '0'     # complete graphs (what the reference's preprocessing writes) skip the edge list in the attention kernels
SIMRANK_F16 = False      # opt-in: fp16-input MFMA similarity (BASELINE.json configs[4]); the default is exact fp32 MFMA


# ---- the autograd nodes, by kernel family; `sgaligner_amd.ops.<name>` resolves to them lazily (any import order works)
_REEXPORT = {
    'pointnet_ops': ('pointnet_bn_fusable', 'pointnet_forward', 'PointNetFn', 'pointnet'),
    'loss_ops': ('_anchor_chunks', '_sym_chunks', '_sym_jobs', 'SWEEP_GRAD_INFO', 'SWEEP_SUMS_INFO', 'TAU_ICL', 'TAU_IAL', 'ALPHA', 'ContrastiveTermsFn',
                 'contrastive_terms', 'LossHeadFn', 'LossGroups', 'GroupedContrastiveFn', 'grouped_contrastive_terms', 'group_data_dicts',
                 '_allreduce_sum', 'FusedContrastiveFn', 'fused_contrastive_terms'),
    'gat_ops': ('GraphBatch', '_attn_fwd', '_attn_bwd', '_elu', 'MultiGATFn', 'multi_gat'),
    'rank_ops': ('PairLayout', 'QueryBlocks', 'simrank', 'pair_metrics'),
}
_WHERE = {name: mod for mod, names in _REEXPORT.items() for name in names}


def __getattr__(name):
    mod = _WHERE.get(name)
    if mod is None:
        raise AttributeError(f"module 'sgaligner_amd.ops' has no attribute {name!r}")
    import importlib
    value = getattr(importlib.import_module('.' + mod, __package__), name)
    globals()[name] = value
    return value
