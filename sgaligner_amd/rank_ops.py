"""Per-pair similarity + ranking for Hits@K / MRR (reference src/inference/sgaligner/inference_align_reg.py:125-143, utils/alignment.py) on csrc/simrank.hip.

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
        if _o.VALIDATE and qi.size and (int(qi.min()) < 0 or int(qi.max()) >= layout.T):
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
    use16 = (_o.SIMRANK_F16 or get_mfma_mode() == 'f16') if f16 is None else bool(f16)
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
