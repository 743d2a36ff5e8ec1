"""The contrastive / alignment loss (reference src/aligner/losses.py) on csrc/contrastive.hip, sweep3.hip, wide16.hip, grouploss.hip, losshead.hip.

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

def _anchor_chunks(a_lo, a_hi, A, n_tables):
    """Anchor-row blocks [c_lo, c_hi) of the anchors x anchors backward.  The coefficient stash of a block is
    n_tables x [A, c_hi - c_lo] fp32; blocks are sized so that it never exceeds _o.STASH_BYTES, which keeps the loss
    backward O(A * D + _o.STASH_BYTES) in memory whatever the batch (4096 pairs x 128 objects: A = 155 648, a full stash
    would be 3 x 97 GB).  Block sizes are multiples of 32 rows (kernel tile) except the last."""
    ns = a_hi - a_lo
    if ns <= 0 or A <= 0:
        return []
    rows = max(32, (_stash_bytes() // (4 * A * max(1, n_tables))) // 32 * 32)
    return [(c, min(c + rows, a_hi)) for c in range(a_lo, a_hi, rows)]


def _sym_chunks(A, n_tables):
    """Blocks of the SYMMETRIC anchors x anchors walk (csrc/contrastive.hip, anchor_multi_bwd16_kernel<.., SYM>): block [lo, hi) meets
    the columns >= lo and keeps two stashes, [A - lo, hi - lo] and [A - hi, hi - lo] floats per table, bounded together by _o.STASH_BYTES --
    so blocks get taller as the walk moves right.  32-row boundaries except the end."""
    return [(lo, hi) for lo, hi, _, _, _ in _sym_jobs([0, A], 0, n_tables)]


def _sym_jobs(cuts, rank, n_tables):
    """The symmetric walk of ONE RANK of an anchor-sharded job (new design): every UNORDERED pair of anchors is visited once
    over all ranks, and every rank visits the same number of pairs.  cuts = [0, c_1, ..., A]: rank r owns the anchor rows
    [c_r, c_r+1) (multiples of 32).  With R ranks, rank r evaluates its own diagonal square and the rectangles (its rows) x (the rows of
    the next K ranks, cyclically), K = (R - 1) / 2 for odd R; for even R the ranks of the lower half take K = R / 2 and the upper half
    R / 2 - 1 (a pair of blocks R / 2 apart is visited by its lower rank only).  Returns launches (lo, hi, j_lo, j_hi, mir) for
    sga_loss_anchor_multi_bwd_symx / sga_loss_stash_grad_symx: rows [lo, hi) x columns [j_lo, j_hi), mirrored elements from column mir on;
    the two stashes of a launch, (j_hi - j_lo) + (j_hi - mir) rows of hi - lo floats per table, stay within _o.STASH_BYTES (blocks get
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
            if dp > 128 and not full and a_lo % 8 and a_hi > a_lo:
                raise RuntimeError('sgaligner_amd: an anchor shard of wide tables (more than 128 columns) must start at a multiple of 8 anchors')
            if dp > 128 and f16:
                # opt-in fp16-input MFMA for wide tables (configs[4]): fp16 copies of the normalised table, once per step
                ldt = int(L.sga_wide16_ldt(s.A, s.J1, s.J2))
                zh = torch.empty((max(s.R, 1), dp), device=dev, dtype=torch.float16)
                zt = torch.empty((dp, ldt), device=dev, dtype=torch.float16)
                _lib.check(L.sga_wide16_prepare(_p(z), dp, s.A, s.J1, s.J2, _p(zh), _p(zt), st), 'sga_wide16_prepare')
                ev = None
                if _o.KERNEL_EVENTS is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                _lib.check(L.sga_loss_neg_sums_f16(_p(zh), dp, s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sk), a_lo, a_hi, st), 'sga_loss_neg_sums_f16')
                if ev is not None:
                    ev[1].record()
                    _o.KERNEL_EVENTS.setdefault('wide16_sums', []).append(ev + ((s.A, s.J1, s.J2, dp),))
            else:
                _lib.check(L.sga_loss_neg_sums_shard(_p(z), dp, s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sk), a_lo, a_hi, st), 'sga_loss_neg_sums')
            sums[k].copy_(sk[:8])
            zs.append(z); nrms.append(nrm); dps.append(dp); zhs.append(zh); zts.append(zt)
        out = torch.empty((slots * (nt + 2 * m),), device=dev, dtype=torch.float64)
        zarr = _ptr_array(zs)
        dparr = (_ct.c_int * nt)(*dps)
        # (mode 'f16': the wide tables' anchors x anchors similarities take their fp16 copies as well -- fp16 inputs, fp32 accumulate)
        sums = _allreduce_sum(sums, reduce)
        # (all tables wide: their 2 nt similarity blocks on the fp16 tile core first, then the epilogue-only kernel -- one anchor-row block at a time)
        all16 = all(dp > 128 for dp in dps)          # every table wide: similarity blocks first (fp16 tile core / fp32 GEMM), epilogue-only kernel
        if all16 and a_hi > a_lo:
            eva = None
            if _o.KERNEL_EVENTS is not None:
                eva = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                eva[0].record()
            out.zero_()
            part = torch.empty_like(out)
            chunks = _anchor_chunks(a_lo, a_hi, s.A, 2 * nt)
            ws = torch.empty((int(L.sga_loss_anchor_f16_ws_bytes(nt, s.A, max(hi - lo for lo, hi in chunks))),), device=dev, dtype=torch.uint8)
            for lo, hi in chunks:
                _lib.check(L.sga_loss_anchor_fwd_f16(zarr, _ptr_array(zhs), dparr, nt, s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(part), lo, hi,
                                                     _p(ws), ws.numel(), st), 'sga_loss_anchor_fwd')
                out += part
            del ws, part
            if eva is not None:
                eva[1].record()
                if all(zh is not None for zh in zhs):
                    _o.KERNEL_EVENTS.setdefault('wide16_aa_fwd', []).append(eva + ((s.A, sum(dps)),))
        else:
            _lib.check(L.sga_loss_anchor_fwd_f16(zarr, _ptr_array(zhs), dparr, nt, s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(out), a_lo, a_hi,
                                                 None, 0, st), 'sga_loss_anchor_fwd')
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
        all16 = nt > 0 and all(dp > 128 for dp in dps)
        chunks = _anchor_chunks(a_lo, a_hi, A, 3 * nt if all16 else nt)      # (all16: stash + the two similarity blocks per table)
        if chunks:
            cmax = max(hi - lo for lo, hi in chunks)
            m1 = [torch.empty((A * cmax,), device=dev, dtype=torch.float32) for _ in range(nt)]
            gsc = torch.empty((slots, nt, 8), device=dev, dtype=torch.float64)
            ws = ws16 = evb = None
            if all16 and _o.KERNEL_EVENTS is not None:
                evb = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                evb[0].record()
            if all16:
                ws = torch.empty((int(L.sga_loss_anchor_f16_ws_bytes(nt, A, cmax)),), device=dev, dtype=torch.uint8)
                if any(zh is not None for zh in zhs):
                    ws16 = torch.empty((int(L.sga_loss_stash_grad_f16_bytes(A, cmax)),), device=dev, dtype=torch.uint8)
            for lo, hi in chunks:          # bounded stash: one anchor-row block at a time
                _lib.check(L.sga_loss_anchor_bwd_f16(_ptr_array(zs), _ptr_array(zhs), dparr, nt, A, _p(sums), ctx.alpha, TAU_ICL, TAU_IAL, _p(coef),
                                                     _ptr_array(m1), _p(gsc), lo, hi, _p(ws) if all16 else None, ws.numel() if all16 else 0, st),
                           'sga_loss_anchor_bwd')
                gs += gsc[0]
                for k in range(nt):
                    # dX1[i] = sum_j G[i,j] X2[j]  (M1 = G^T),  dX2[j] = sum_i G[i,j] X1[i]
                    if zhs[k] is not None and lo % 8 == 0:
                        _lib.check(L.sga_loss_stash_grad_f16(_p(m1[k]), _p(zts[k]), dps[k], A, s.J1, s.J2, _p(dzs[k]), lo, hi, _p(ws16) if ws16 is not None else None,
                                                             ws16.numel() if ws16 is not None else 0, st), 'sga_loss_stash_grad_f16')
                    else:
                        _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zs[k]), A, dps[k], _p(dzs[k]), lo, hi, st), 'sga_loss_stash_grad')
            del m1, ws, ws16
            if evb is not None:
                evb[1].record()
                if all(zh is not None for zh in zhs):
                    _o.KERNEL_EVENTS.setdefault('wide16_aa_bwd', []).append(evb + ((A, sum(dps)),))
        gs = _allreduce_sum(gs, ctx.reduce)                      # dL/d(global sums) needs every shard's anchors x anchors tiles
        grads = []
        for k in range(nt):
            z, dp, dz = zs[k], dps[k], dzs[k]
            ev = None
            if _o.KERNEL_EVENTS is not None and dp <= 128:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if zhs[k] is not None:
                # opt-in fp16-input MFMA (configs[4]): S and both gradient GEMMs on v_mfma_f32_32x32x16_f16 (csrc/wide16.hip)
                need = int(L.sga_loss_neg_grad_f16_bytes(A, s.J1, s.J2))
                have = max(min(need, _stash_bytes()), int(L.sga_loss_neg_grad_f16_bytes(min(A, 128), s.J1, s.J2)))
                stash = torch.empty((have,), device=dev, dtype=torch.uint8)
                ev16 = None
                if _o.KERNEL_EVENTS is not None:
                    ev16 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev16[0].record()
                _lib.check(L.sga_loss_neg_grad_f16(_p(zhs[k]), _p(zts[k]), dp, A, s.J1, s.J2, TAU_ICL, TAU_IAL, gs[k].data_ptr(), _p(dz),
                                                   _p(stash), have, a_lo, a_hi, st), 'sga_loss_neg_grad_f16')
                if ev16 is not None:
                    ev16[1].record()
                    _o.KERNEL_EVENTS.setdefault('wide16_grad', []).append(ev16 + ((A, s.J1, s.J2, dp),))
                del stash
            elif dp > 128 and _o.WIDE_STASH and a_lo == 0 and a_hi == A:
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
                _o.KERNEL_EVENTS.setdefault('sweep_kernel<4,4,grad>', []).append(ev + ((A, s.J1, s.J2, dp),))
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
                                        float(alpha), TAU_ICL, TAU_IAL, _p(S), _p(sums), _p(out), int(_o.GROUP_LOSS_VALU), st), 'sga_group_loss_fwd')
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
                                        ctx.alpha, TAU_ICL, TAU_IAL, _p(C), _p(sums), _p(coef), _ptr_array(dzs), _p(gamma), int(_o.GROUP_LOSS_VALU), st),
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


def _sums_lite(ns, J1, J2):
    """Forward sums of the three-plane sweeps from the h and m planes only (csrc/sweep3.hip, LITE)?  ops.BF16X6_SUMS_LITE = True / False forces it;
    None (default): only when the smallest of the four global sums of this rank's shard has at least BF16X6_SUMS_LITE_MIN_TERMS terms."""
    if _o.BF16X6_SUMS_LITE is not None:
        return bool(_o.BF16X6_SUMS_LITE)
    return ns * min(J1, J2) >= _o.BF16X6_SUMS_LITE_MIN_TERMS


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
        zbs, zcs = [], []
        split3 = centred32 = False
        if M in (2, 3, 4) and dmax <= 100 and _o.FUSED_ANCHOR_BWD and get_mfma_mode() in ('bf16x6', 'f16'):
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
            if _o.KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            lite = _sums_lite(a_hi - a_lo, s.J1, s.J2)
            _lib.check(L.sga_loss_multi_sums_bf16x6(_ptr_array(zbs), M, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums),
                                                    a_lo, a_hi, 1 if lite else 0, st), 'sga_loss_multi_sums_bf16x6')
            if ev is not None:
                ev[1].record()
                _o.KERNEL_EVENTS.setdefault('loss_multi_sums_bf16x6', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M, bool(lite)),))
        elif M in (2, 3, 4) and dmax <= 100 and _o.FUSED_ANCHOR_BWD and _o.CENTRED_F32:
            # fp32-MFMA sweeps over the centred fp32 tables (z - zbar | b | 1): the gradient in the same two parts as the three-plane sweeps
            centred32 = True
            nst = int(L.sga_loss_centre_bytes())
            for z in zs:
                zc = torch.empty((s.R + 32, dp), device=dev, dtype=torch.float32)
                zc[s.R:].zero_()
                stw = torch.empty((nst,), device=dev, dtype=torch.uint8)
                _lib.check(L.sga_loss_centre_tables(_p(z), s.A, s.J1, s.J2, _p(zc), _p(stw), st), 'sga_loss_centre_tables')
                zbs.append(stw); zcs.append(zc)
            ev = None
            if _o.KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(L.sga_loss_multi_sums_centred(_ptr_array(zcs), M, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums), a_lo, a_hi, st),
                       'sga_loss_multi_sums_centred')
            if ev is not None:
                ev[1].record()
                _o.KERNEL_EVENTS.setdefault('loss_multi_sums', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M),))
        else:
            ev = None
            if _o.KERNEL_EVENTS is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(L.sga_loss_multi_sums(zarr, M, dmax, _p(beta), s.A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(sums), a_lo, a_hi, st),
                       'sga_loss_multi_sums')
            if ev is not None:
                ev[1].record()
                _o.KERNEL_EVENTS.setdefault('loss_multi_sums', []).append(ev + ((a_hi - a_lo, s.A, s.J1, s.J2, M),))
        sums = _allreduce_sum(sums[0].contiguous(), reduce)
        zj = torch.empty((2 * s.A, M * dp), device=dev, dtype=torch.float32)
        _lib.check(L.sga_loss_build_joint(zarr, M, _p(beta), 2 * s.A, _p(zj), st), 'sga_loss_build_joint')
        out = torch.empty((slots * (nt + 2 * M),), device=dev, dtype=torch.float64)
        dps = [dp] * M + [M * dp]
        onepass = coef_hint is not None and M <= 4 and _o.FUSED_ANCHOR_BWD and _o.FUSED_AA_ONEPASS and s.A >= _o.ONEPASS_MIN_ANCHORS
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
            sym = _o.AA_SYMMETRIC and M <= _o.AA_SYMMETRIC_MAX_M and cuts is not None and all(c % 32 == 0 for c in cuts[:-1]) and cuts[-1] == s.A \
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
                for lo, hi, jl, jh, mir in chunks:
                    n1 = (jh - jl) * (hi - lo)
                    m1 = [b[:n1] for b in buf]
                    m2 = [b[n1:] for b in buf]
                    has2 = mir < jh
                    _lib.check(L.sga_loss_anchor_multi_bwd_symx(zarr, M, _p(beta), s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(coef),
                                     _ptr_array(m1), _ptr_array(m2) if has2 else (_ct.c_void_p * M)(), _p(gsc), _p(gam2),
                                     lo, hi, jl, jh, mir, _p(out), st), 'sga_loss_anchor_multi_bwd_symx')
                    out_acc += out[:n_terms]
                    gs_aa += gsc[0]
                    gam_aa += gam2[0]
                    for k in range(M):
                        if split3 and _o.BF16X6_STASH:
                            # the four stash products on the sweeps' three exact bf16 planes (csrc/sweep3.hip: stash3_kernel)
                            _lib.check(L.sga_loss_stash_grad_symx_bf16x6(_p(m1[k]), _p(m2[k]) if has2 else None, _p(zbs[k]), s.A, s.J1, s.J2,
                                                                         _p(dz_all[k]), lo, hi, jl, jh, mir, st), 'sga_loss_stash_grad_symx_bf16x6')
                        else:
                            _lib.check(L.sga_loss_stash_grad_symx(_p(m1[k]), _p(m2[k]) if has2 else None, _p(zcs[k] if (split3 or centred32) else zs[k]), s.A, dp,
                                                                  _p(dz_all[k]), lo, hi, jl, jh, mir, st), 'sga_loss_stash_grad_symx')
                del buf, m1, m2
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
                        if split3 and _o.BF16X6_STASH:
                            _lib.check(L.sga_loss_stash_grad_symx_bf16x6(_p(m1[k]), None, _p(zbs[k]), s.A, s.J1, s.J2, _p(dz_all[k]), lo, hi, 0, s.A, s.A, st),
                                       'sga_loss_stash_grad_symx_bf16x6')
                        else:
                            _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zcs[k] if (split3 or centred32) else zs[k]), s.A, dp, _p(dz_all[k]), lo, hi, st), 'sga_loss_stash_grad')
                del m1
            out = _allreduce_sum(out_acc.clone(), reduce)
            extra = [dz_all, gs_aa.clone(), gam_aa.clone(), coef]
        else:
            if M <= 4 and _o.FUSED_ANCHOR_FWD:      # joint similarities derived in registers, I block resident in LDS
                _lib.check(L.sga_loss_anchor_multi_fwd(zarr, M, _p(beta), s.A, _p(sums), float(alpha), TAU_ICL, TAU_IAL, _p(out),
                                                       a_lo, a_hi, st), 'sga_loss_anchor_multi_fwd')
            else:
                _lib.check(L.sga_loss_anchor_fwd(_ptr_array(zs + [zj]), (_ct.c_int * nt)(*dps), nt, s.A, _p(sums), float(alpha),
                                                 TAU_ICL, TAU_IAL, _p(out), a_lo, a_hi, st), 'sga_loss_anchor_fwd')
            out = _allreduce_sum(out[:nt + 2 * M].contiguous(), reduce)
        ctx.s, ctx.alpha, ctx.M, ctx.shard, ctx.reduce = s, float(alpha), M, (a_lo, a_hi), reduce
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.n_zb = len(zbs)
        ctx.split3 = split3
        ctx.centred32 = centred32
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
            # Always (_o.VALIDATE or not), on the device and without a host sync: a mismatching gradient POISONS what this node returns --
            # every table gradient, dL/d(sums) and dL/dbeta become NaN -- so the wrong gradients can never be consumed silently by an
            # optimiser step that runs before the deferred error below is polled.
            mismatch = (coef - u * hint).abs().max() > 1e-4 * hh.sqrt()
            u = torch.where(mismatch, torch.full_like(u, float('nan')), u)
            if _o.VALIDATE:
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
        # buffers -- memory O(A*D + _o.STASH_BYTES), never A x A (nothing of that size at configs[2]).
        fused = M <= 4 and _o.FUSED_ANCHOR_BWD
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
                if ctx.split3 and _o.BF16X6_STASH:
                    _lib.check(L.sga_loss_stash_grad_symx_bf16x6(_p(m1[k]), None, _p(zbs[k]), A, s.J1, s.J2, _p(dzs[k]), lo, hi, 0, A, A, st),
                               'sga_loss_stash_grad_symx_bf16x6')
                else:
                    _lib.check(L.sga_loss_stash_grad(_p(m1[k]), _p(zcs[k] if (ctx.split3 or ctx.centred32) else zs[k]), A, dp, _p(dzs[k]), lo, hi, st), 'sga_loss_stash_grad')
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
        if _o.KERNEL_EVENTS is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if ctx.n_zb and ctx.split3:           # the forward ran in bf16x6 mode: its blocked bf16 h / m / l planes are there
            _lib.check(L.sga_loss_multi_grad_bf16x6(_ptr_array(zbs), M, _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                                    _p(gam_neg), a_lo, a_hi, st), 'sga_loss_multi_grad_bf16x6')
        elif ctx.centred32:                   # fp32 MFMA over the centred fp32 tables
            _lib.check(L.sga_loss_multi_grad_centred(_ptr_array(zcs), M, _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                                     _p(gam_neg), a_lo, a_hi, st), 'sga_loss_multi_grad_centred')
        else:
            _lib.check(L.sga_loss_multi_grad(_ptr_array(zs), M, max(d for _, d in ctx.shapes), _p(beta), A, s.J1, s.J2, TAU_ICL, TAU_IAL, _p(gs), _ptr_array(dzs),
                                             _p(gam_neg), a_lo, a_hi, st), 'sga_loss_multi_grad')
        if ev is not None:
            ev[1].record()
            _o.KERNEL_EVENTS.setdefault('loss_multi_grad_bf16x6' if ctx.split3 else 'loss_multi_grad', []).append(ev + ((ns, A, s.J1, s.J2, M),))
        grads = []
        same = all(sh == ctx.shapes[0] for sh in ctx.shapes)
        de_all = torch.zeros((M,) + tuple(ctx.shapes[0]), device=dev, dtype=torch.float32) if same else None   # one fill
        for k in range(M):
            t, d = ctx.shapes[k]
            de = de_all[k] if same else torch.zeros((t, d), device=dev, dtype=torch.float32)
            if ctx.n_zb and ctx.split3:       # the gradient is in two parts (sum c (z - zbar) | sum c): projected without forming their sum
                _lib.check(L.sga_loss_scatter_tangent(_p(dzs[k]), _p(zs[k]), _p(nrms[k]), _p(s.idx), A, s.J1, s.J2, d, _p(zbs[k]), _p(de), st),
                           'sga_loss_scatter_tangent')
            elif ctx.centred32:
                _lib.check(L.sga_loss_scatter_tangent_stat(_p(dzs[k]), _p(zs[k]), _p(nrms[k]), _p(s.idx), s.R, d, _p(zbs[k]), _p(de), st),
                           'sga_loss_scatter_tangent_stat')
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
