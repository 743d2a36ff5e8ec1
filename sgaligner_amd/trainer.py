"""Engine hooks for the path: `train_step / val_step / test_step / eval_step` with the reference's
signatures (src/engine/epoch_based_trainer.py:56-60, src/engine/single_tester.py:52-63,
src/trainers/trainval_sgaligner.py:71-79, src/inference/sgaligner/inference_align_reg.py:74-76,98-143),
so the reference's EpochBasedTrainer / SingleTester loops can call them unchanged.
Multi-GPU: one process per GPU, pairs sharded, tables all-gathered for the batch-global loss."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import dist as sdist
from . import ops
from .aligner import losses
from .aligner.losses import CustomMultiLossLayer, OverallLoss
from .aligner.sg_aligner import MultiModalEncoder
from .utils import alignment


WIDE_SHARDED = True      # N > 1, tables outside the fused path: anchor-sharded general kernels (False: a replica of the whole loss per rank)

class _ScaleGrad(torch.autograd.Function):
    """Identity whose backward multiplies the gradient by a constant (the replica fallback of AlignerSteps._global_loss)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = float(k)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.k, None


class AlignerSteps:
    def __init__(self, modules, rel_dim=41, attr_dim=164, zoom=0.1, device='cuda', seed=42, loss_group='global', emb_dim=100):
        if not torch.cuda.is_available() and str(device).startswith('cuda'):
            raise RuntimeError('sgaligner_amd.AlignerSteps: no HIP device; the product path has no CPU fallback')
        self.modules = list(modules)
        self.device = torch.device(device)
        torch.manual_seed(seed)                                   # identical replicas on every rank
        self.model = MultiModalEncoder(modules=self.modules, rel_dim=rel_dim, attr_dim=attr_dim, emb_dim=emb_dim).to(self.device)
        m = len(self.modules)
        self.multi_loss_layer_icl = CustomMultiLossLayer(loss_num=m, device=self.device).to(self.device)
        self.multi_loss_layer_ial = CustomMultiLossLayer(loss_num=m, device=self.device).to(self.device)
        # loss_group: 'global' (one loss over the whole batch, all-gathered across ranks) or b = reference-sized groups of b pairs
        self.loss_group = loss_group
        meta = {'zoom': zoom, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': self.modules, 'loss_group': loss_group}
        self.loss_func = OverallLoss(self.multi_loss_layer_ial, self.multi_loss_layer_icl, self.device, meta)
        self.params = list(self.model.parameters())
        if m > 1:
            self.params += list(self.multi_loss_layer_ial.parameters()) + list(self.multi_loss_layer_icl.parameters())

    # -- reference hook names ---------------------------------------------------------------------
    # Multi-GPU, batch-global loss: launch each modality table's all-gather the moment the encoder has produced it (dist.EarlyGather)
    # instead of gathering all tables after the encoder.  Same numbers (tests/test_dist_gpu.py); False = the blocking path.
    overlap_gather = True

    def train_step(self, epoch, iteration, data_dict):
        if dist.is_initialized() and dist.get_world_size() > 1 and self.loss_group in (None, 'global'):
            layout = sdist.layout_of(data_dict, self.device)                  # [world, 4]: rows, |e1i|, |e1j|, |e2j|
            early = None
            if len(self.modules) > 1 and self.overlap_gather:
                early = sdist.EarlyGather([int(v) for v in layout[:, 0]])
                self.model._on_table = early
            try:
                output_dict = self.model(data_dict)
                loss_dict = self._global_loss(output_dict, data_dict, layout, early)
            finally:
                self.model._on_table = None
                if early is not None:
                    early.drain()                        # no-op after tables(); waits for the gathers in flight if the step raised
        else:
            output_dict = self.model(data_dict)
            loss_dict = self.loss_func(output_dict, data_dict)
        return output_dict, loss_dict

    val_step = train_step

    def test_step(self, iteration, data_dict):
        with torch.no_grad():                       # the reference's tester runs under set_eval_mode / no_grad (single_tester.py:52-63)
            return self.model(data_dict)

    def eval_step(self, iteration, data_dict, output_dict, all_k=(1, 2, 3, 4, 5), reg_k=0):
        emb = output_dict['joint'] if len(self.modules) > 1 else output_dict[self.modules[0]]
        return alignment.evaluate_batch(emb.detach(), data_dict, all_k=all_k, reg_k=reg_k)

    # -- fwd + loss + bwd (what bench.py times) ---------------------------------------------------
    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def forward_backward(self, data_dict):
        self.zero_grad()
        output_dict, loss_dict = self.train_step(0, 0, data_dict)
        loss_dict['loss'].backward()
        if dist.is_initialized() and dist.get_world_size() > 1:
            self.reduce_grads()
        return output_dict, loss_dict

    def reduce_grads(self):
        """One flat SUM all-reduce of the parameter gradients.  With the batch-global loss the log_vars see the full
        (replicated) loss value on every rank -- pre-divide them; everything else (and everything under loss_group=b, where
        each rank's loss is the sum over ITS groups) holds this rank's share only."""
        if self.loss_group in (None, 'global'):
            w = dist.get_world_size()
            for p in list(self.multi_loss_layer_ial.parameters()) + list(self.multi_loss_layer_icl.parameters()):
                if p.grad is not None:
                    p.grad /= w
        sdist.allreduce_grads(self.params)

    # -- batch-global loss across ranks -------------------------------------------------------------
    def _global_loss(self, output_dict, data_dict, layout=None, early=None):
        """Batch-global loss over all ranks' pairs.  Tables and index sets are all-gathered (RCCL).
        M >= 2 (fused joint path): the anchors are SHARDED -- each rank evaluates the loss terms / global sums of its own
        anchors against all negatives (partial scalars all-reduced inside ops.FusedContrastiveFn, so every rank holds the
        global loss value) and its share of dL/dE for all rows, summed over ranks in AllGatherRows.backward.
        M == 1 (ICL of the one table of <= 128 columns, the general per-table kernels): sharded by anchors the same way.
        Anything else (wide tables -- BASELINE configs[4] --, FUSED_JOINT off): the general per-table kernels, sharded by anchors as well
        (balanced cuts on 32-anchor boundaries); WIDE_SHARDED = False: a replica of the whole loss on every rank (the round-5 form, a cross-check)."""
        world, rank = dist.get_world_size(), dist.get_rank()
        if layout is None:
            layout = sdist.layout_of(data_dict, self.device)                # [world, 4]: rows, |e1i|, |e1j|, |e2j|
        rows = [int(v) for v in layout[:, 0]]
        anchors = [int(v) for v in layout[:, 1]]
        mods = list(self.modules)
        widths = [int(output_dict[m].shape[1]) for m in mods]
        fused = len(mods) in (2, 3, 4) and max(widths) <= 104 and losses.FUSED_JOINT
        idx, A, J1, J2 = sdist.gather_index_sets_device(data_dict, layout, self.device)
        gdd = {'_sga_index_sets': ops.IndexSets.from_device(idx, A, J1, J2)}   # our own dict, never the caller's
        if early is not None:                                                # launched from inside the encoder, table by table
            tabs = early.tables(mods)
        else:
            tabs = sdist.gather_tables({m: output_dict[m] for m in mods}, rows, reduce_grad=True)
        general_wide = not fused and not (len(mods) == 1 and widths[0] <= 128)
        if general_wide and not WIDE_SHARDED:
            # REPLICA of the whole loss on every rank (kept as a cross-check of the sharded form below: trainer.WIDE_SHARDED = False).  The table
            # gradients are still summed over ranks by AllGatherRows.backward and the fusion weight's by the parameter all-reduce, and every rank
            # holds the FULL gradient, not a share: both enter through a 1 / world factor.
            k = 1.0 / world
            gathered = {m: _ScaleGrad.apply(tabs[m], k) for m in mods}
            if len(mods) > 1:
                fusion = self.model.fusion
                joint = ops.fusion(_ScaleGrad.apply(fusion.weight, k), [gathered[m] for m in mods])     # sg_aligner.py:30-35 on the gathered rows
                joint._sga_fusion = (fusion.weight, tuple(gathered[m] for m in mods))
                gathered['joint'] = joint
            return self.loss_func(gathered, gdd)

        def _reduce(t):                                                   # fp64 partial sums / loss terms / dL/d(sums)
            sdist._log('all_reduce', t, t)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if general_wide:
            # Tables the fused kernels do not take (wider than 104 columns under a fused joint -- BASELINE configs[4], emb_dim 1024 --, a single table
            # wider than 128, FUSED_JOINT off): the general per-table kernels, SHARDED BY ANCHORS like the fused path (round 6: the wide-table
            # kernels take an anchor range) -- every rank forms the joint table from the gathered modality tables (sg_aligner.py:30-35), evaluates
            # the global sums / loss terms of its anchor range against all rows (all-reduced) and its share of dL/dE for all rows (summed over ranks
            # in AllGatherRows.backward; the fusion weight's share by the parameter all-reduce).  Any partition of [0, A) is valid since every
            # rank holds all rows: balanced cuts on 32-anchor boundaries (the wide kernels want a_lo % 8 == 0).
            gathered = dict(tabs)
            if len(mods) > 1:
                fusion = self.model.fusion
                joint = ops.fusion(fusion.weight, [gathered[m] for m in mods])
                joint._sga_fusion = (fusion.weight, tuple(gathered[m] for m in mods))
                gathered['joint'] = joint
            wcuts = [min(A, (A * r // world + 31) // 32 * 32) for r in range(world)] + [A]
            gdd['_sga_shard'] = (wcuts[rank], wcuts[rank + 1])
            gdd['_sga_reduce'] = _reduce
            return self.loss_func(gathered, gdd)
        gathered = dict(tabs)
        if fused:
            # only the M modality tables travel: the fused loss derives every joint similarity from them (S_J = sum beta_m
            # S_m with the replicated fusion weight), so the 100*M-wide joint table is neither gathered nor reduced --
            # half of the bytes of both collectives.  The placeholder only carries the provenance tag OverallLoss checks.
            joint = torch.empty((0,), device=self.device)
            joint._sga_fusion = (self.model.fusion.weight, tuple(gathered[m] for m in mods))
            gathered['joint'] = joint
        # (a_lo, a_hi) of this rank + every rank's cut and the rank: with all cuts on 32-row boundaries the anchors x anchors pairs are
        # walked symmetrically ACROSS ranks (ops._sym_jobs: every unordered pair once, the same number on every rank)
        cuts = [sum(anchors[:r]) for r in range(world + 1)]
        gdd['_sga_shard'] = (cuts[rank], cuts[rank + 1], cuts, rank)
        gdd['_sga_reduce'] = _reduce
        return self.loss_func(gathered, gdd)
