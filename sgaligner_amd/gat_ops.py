"""MultiGAT / GATConv message passing (reference src/aligner/networks/gat.py:27-48) on csrc/gat.hip.

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
        if _o.VALIDATE and self.E and edges.is_cuda:
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
        if not _o.GAT_COMPLETE_FAST_PATH or self.G == 0 or self.edges is None or not self.edges.is_cuda:
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
        if _o.VALIDATE:
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


def _gat_status_verdict(v):
    if v[0] == 0:
        return None
    return ('sgaligner_amd: a (source, target) edge occurs more than 255 times in one graph of the PREVIOUS batch; the GAT kernels count '
            'duplicate edges in 8 bits (PyG would count them all), so that step\'s structure embeddings were not PyG-equivalent -- '
            'deduplicate the edge list')


def _attn_fwd(h, att_s, att_d, bias, gb, check_status=False):
    out = torch.empty_like(h)
    st = None
    if check_status and _o.VALIDATE:
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
