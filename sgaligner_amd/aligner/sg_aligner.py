"""Multi-modal node encoder -- drop-in for reference src/aligner/sg_aligner.py (same class names,
constructor signatures, attributes, state_dict keys and data_dict contract), computing on the HIP
kernels through sgaligner_amd.ops.  There is no CPU path: tensors must live on the MI355X."""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (re-exported by `from aligner.sg_aligner import *`, as the reference does)

from .. import ops
from .networks.gat import MultiGAT
from .networks.pct import NaivePCT
from .networks.pointnet import PointNetfeat


class _Linear(nn.Linear):
    """nn.Linear parameters (same state_dict keys / init) with forward/backward on the MFMA GEMM."""

    def forward(self, x):
        if self.bias is None:
            raise NotImplementedError('sgaligner_amd: bias-free Linear is not on the hot path')
        return ops.linear(x, self.weight, self.bias)


class ProjectionHead(nn.Module):
    """sg_aligner.py:9-21 -- defined by the reference, never used on the path; kept for import parity."""

    def __init__(self, in_dim, hidden_dim, out_dim, dropout):
        super().__init__()
        self.l1 = nn.Linear(in_dim, hidden_dim, bias=False)
        self.l2 = nn.Linear(hidden_dim, out_dim, bias=False)
        self.dropout = dropout

    def forward(self, x):
        raise NotImplementedError('ProjectionHead is dead code in the reference (sg_aligner.py:9-21); not on the HIP path')


class MultiModalFusion(nn.Module):
    """sg_aligner.py:23-35."""

    def __init__(self, modal_num, with_weight=1):
        super().__init__()
        self.modal_num = modal_num
        self.requires_grad = True if with_weight > 0 else False
        self.weight = nn.Parameter(torch.ones((self.modal_num, 1)), requires_grad=self.requires_grad)

    def forward(self, embs):
        assert len(embs) == self.modal_num
        if any(e is None for e in embs):
            raise NotImplementedError('sgaligner_amd MultiModalFusion: None entries are not supported')
        joint = ops.fusion(self.weight, list(embs))
        # provenance tag: lets OverallLoss derive the joint similarities from the modality tiles
        # (S_joint = sum_m beta_m S_m) instead of sweeping the 100*M-d table (ops.FusedContrastiveFn)
        joint._sga_fusion = (self.weight, tuple(embs))
        return joint


class MultiModalEncoder(nn.Module):
    """sg_aligner.py:37-137.  `modules` (list of 'point' | 'pct' | 'gat' | 'rel' | 'attr') is kept as an attribute
    with the reference's name -- it shadows nn.Module.modules(), exactly as in the reference (:41)."""

    def __init__(self, modules, rel_dim, attr_dim, hidden_units=[3, 128, 128], heads=[2, 2], emb_dim=100,
                 pt_out_dim=256, dropout=0.0, attn_dropout=0.0, instance_norm=False):
        super().__init__()
        self.modules = modules
        self.pt_out_dim = pt_out_dim
        self.rel_dim = rel_dim
        self.emb_dim = emb_dim
        self.attr_dim = attr_dim
        self.hidden_units = hidden_units
        self.heads = heads
        self.dropout = dropout
        self.attn_dropout = attn_dropout
        self.instance_norm = instance_norm
        self.inner_view_num = len(self.modules)

        self.meta_embedding_rel = _Linear(self.rel_dim, self.emb_dim)
        self.meta_embedding_attr = _Linear(self.attr_dim, self.emb_dim)
        if 'point' in self.modules:
            self.object_encoder = PointNetfeat(global_feat=True, batch_norm=True, point_size=3, input_transform=False,
                                               feature_transform=False, out_size=self.pt_out_dim)
        elif 'pct' in self.modules:
            self.object_encoder = NaivePCT()                             # sg_aligner.py:59-60 (eval and training: pct_ops.py)
        else:
            raise NotImplementedError                                   # sg_aligner.py:61-62
        self.object_embedding = _Linear(self.pt_out_dim, self.emb_dim)
        self.structure_encoder = MultiGAT(n_units=self.hidden_units, n_heads=self.heads, dropout=self.dropout)
        self.structure_embedding = _Linear(256, self.emb_dim)
        self.fusion = MultiModalFusion(modal_num=self.inner_view_num, with_weight=1)
        self._on_table = None          # optional callable(module, table), see forward

    def forward(self, data_dict):
        pts = data_dict['tot_obj_pts']
        if not pts.is_cuda:
            raise RuntimeError('sgaligner_amd.MultiModalEncoder: data_dict tensors must be on the HIP device '
                               '(utils/torch_util.to_cuda in the reference); there is no CPU path')
        # Order of evaluation: the reference's module order -- unless a table hook is installed (multi-GPU, dist.EarlyGather: every
        # finished table starts its all-gather at once), then the cheap modalities go first so that their tables travel while the
        # object encoder runs.  The tables do not depend on each other, so the values are the same either way.
        hook = self._on_table
        order = self.modules if hook is None else sorted(self.modules, key=lambda m: m in ('point', 'pct'))
        done = {}
        for module in order:
            if module == 'gat':
                # all 2B graphs in one launch per layer (reference: 2B sequential GATConv calls, :86-110)
                gb = ops.GraphBatch.of(data_dict)
                emb = self.structure_encoder.forward_batched(data_dict['tot_rel_pose'], gb)
                emb = self.structure_embedding(emb)
            elif module in ('point', 'pct'):
                emb = self.object_encoder(pts.permute(0, 2, 1))         # :72,:115
                emb = self.object_embedding(emb)
            elif module == 'rel':
                emb = self.meta_embedding_rel(data_dict['tot_bow_vec_object_edge_feats'])   # f64 cast fused in the loader
            elif module == 'attr':
                emb = self.meta_embedding_attr(data_dict['tot_bow_vec_object_attr_feats'])
            else:
                raise NotImplementedError                               # :124-125
            done[module] = emb
            if hook is not None:
                hook(module, emb)
        embs = {m: done[m] for m in self.modules}
        if len(self.modules) > 1:
            embs['joint'] = self.fusion([embs[m] for m in self.modules])
        return embs
