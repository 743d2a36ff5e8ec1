"""One-process-per-GPU data parallelism for the path: subscan pairs shard across ranks; the only data-path
collective is the all-gather of the embedding tables for the batch-global contrastive terms (RCCL over
xGMI; `nccl` backend on ROCm), plus the usual parameter-gradient all-reduce.  The reference has no
working multi-GPU path (engine/base_trainer.py:70 hard-codes distributed=False); this is new design
(SURVEY.md 8e).  Everything here also runs on CPU tensors with the `gloo` backend (tests)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for 1 process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('SGA_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()       # several ranks may share a GPU in gloo test runs
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition [lo, hi) of n_items over `world` ranks (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class AllGatherRows(torch.autograd.Function):
    """out = cat_r(x_r) along dim 0 (ranks may hold different row counts).  Backward: every rank holds
    dL/d(out) of ITS OWN loss replica/shard; the gradient of the summed objective wrt x_r is the sum
    over ranks of the corresponding row block -> all-reduce(sum) then slice (== reduce-scatter)."""

    @staticmethod
    def forward(ctx, x, rows_per_rank, reduce_grad):
        world = dist.get_world_size()
        rank = dist.get_rank()
        ctx.rows, ctx.rank, ctx.reduce_grad = list(rows_per_rank), rank, reduce_grad
        mx = max(ctx.rows)
        pad = x
        if x.shape[0] < mx:
            pad = torch.cat([x, x.new_zeros((mx - x.shape[0],) + tuple(x.shape[1:]))])
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad.contiguous())
        return torch.cat([b[:n] for b, n in zip(bufs, ctx.rows)])

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.reduce_grad:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        lo = sum(ctx.rows[:ctx.rank])
        return g[lo:lo + ctx.rows[ctx.rank]], None, None


def gather_tables(output_dict, rows_per_rank, reduce_grad):
    return {k: AllGatherRows.apply(v, rows_per_rank, reduce_grad) for k, v in output_dict.items()}


def gather_index_sets(data_dict, rows_per_rank):
    """Global e1i/e2i/e1j/e2j (host numpy) from every rank's local ones, offset by the ranks' object counts."""
    world, rank = dist.get_world_size(), dist.get_rank()
    local = {k: np.asarray(data_dict[k]).astype(np.int64) for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    offs = np.concatenate([[0], np.cumsum(rows_per_rank)])
    out = {}
    for k in ('e1i', 'e2i', 'e1j', 'e2j'):
        out[k] = np.concatenate([g[k] + offs[r] for r, g in enumerate(gathered)]).astype(np.int32)
    return out


def allreduce_grads(params, average: bool = False):
    """Sum (or average) parameter gradients across ranks in ONE flat fp32 message (182 440 params = 0.73 MB:
    latency-bound, so a single bucket)."""
    ps = [p for p in params if p.grad is not None]
    if not ps or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


def shard_data_dict(data_dict, lo: int, hi: int):
    """The pairs [lo, hi) of a collated batch as a self-contained data_dict (what this rank's loader would have
    collated): tensors sliced by object / edge ranges, index sets re-based to the shard's first object."""
    cnt = np.asarray(data_dict['tot_obj_count']).reshape(-1)
    offs = np.concatenate([[0], np.cumsum(cnt)])
    o0, o1 = int(offs[lo]), int(offs[hi])
    out = {}
    for k in ('tot_obj_pts', 'tot_bow_vec_object_attr_feats', 'tot_bow_vec_object_edge_feats', 'tot_rel_pose'):
        if k in data_dict:
            out[k] = data_dict[k][o0:o1]
    if 'edges' in data_dict:
        ec = np.asarray(data_dict['graph_per_edge_count']).reshape(-1, 2)
        eo = np.concatenate([[0], np.cumsum(ec.sum(1))])
        out['edges'] = data_dict['edges'][int(eo[lo]):int(eo[hi])]
        out['graph_per_edge_count'] = ec[lo:hi]
    for name, ck in (('e1i', 'e1i_count'), ('e2i', 'e2i_count'), ('e1j', 'e1j_count'), ('e2j', 'e2j_count')):
        c = np.asarray(data_dict[ck]).reshape(-1)
        co = np.concatenate([[0], np.cumsum(c)])
        out[name] = (np.asarray(data_dict[name])[int(co[lo]):int(co[hi])] - o0).astype(np.int32)
        out[ck] = c[lo:hi]
    out['tot_obj_count'] = cnt[lo:hi]
    out['graph_per_obj_count'] = np.asarray(data_dict['graph_per_obj_count'])[lo:hi]
    out['batch_size'] = hi - lo
    return out
