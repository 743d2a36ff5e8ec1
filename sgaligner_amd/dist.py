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


def _all_gather_flat(out, inp):
    """out [world * n, ...] = cat of every rank's inp [n, ...].  RCCL: one all_gather_into_tensor; gloo (CPU and the
    shared-GPU tests): the list form, which it supports for host and device tensors alike."""
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(out, inp)
    else:
        n = inp.shape[0]
        dist.all_gather([out[r * n:(r + 1) * n] for r in range(dist.get_world_size())], inp)


def _has_reduce_scatter():
    return dist.get_backend() == 'nccl'          # RCCL; gloo (CPU / shared-GPU tests) has no reduce-scatter


class AllGatherRows(torch.autograd.Function):
    """out = cat_r(x_r) along dim 0 (ranks may hold different row counts).  Backward: every rank holds
    dL/d(out) of ITS OWN loss replica/shard; the gradient of the summed objective wrt x_r is the sum over ranks of
    the corresponding row block -> ONE reduce-scatter (RCCL), each rank receiving only its own rows: half the bytes on
    the wire of all-reduce + slice (2.5 GB -> 1.26 GB per rank at configs[2]).  gloo falls back to all-reduce + slice."""

    @staticmethod
    def forward(ctx, x, rows_per_rank, reduce_grad):
        world = dist.get_world_size()
        rank = dist.get_rank()
        ctx.rows, ctx.rank, ctx.reduce_grad = list(rows_per_rank), rank, reduce_grad
        mx = max(ctx.rows)
        pad = x
        if x.shape[0] < mx:
            pad = torch.cat([x, x.new_zeros((mx - x.shape[0],) + tuple(x.shape[1:]))])
        if all(n == mx for n in ctx.rows):           # the usual case (equal shards): gather straight into the result
            out = torch.empty((world * mx,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
            _all_gather_flat(out, pad.contiguous())
            return out
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad.contiguous())
        return torch.cat([b[:n] for b, n in zip(bufs, ctx.rows)])

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        rows, rank = ctx.rows, ctx.rank
        lo = sum(rows[:rank])
        if not ctx.reduce_grad:
            return g[lo:lo + rows[rank]], None, None
        if _has_reduce_scatter():
            world, mx = len(rows), max(rows)
            if all(n == mx for n in rows):
                inp = g
            else:                                    # ragged shards: blocks padded to the largest one
                inp = g.new_zeros((world * mx,) + tuple(g.shape[1:]))
                o = 0
                for r, n in enumerate(rows):
                    inp[r * mx:r * mx + n] = g[o:o + n]
                    o += n
            out = torch.empty((mx,) + tuple(g.shape[1:]), device=g.device, dtype=g.dtype)
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)
            return out[:rows[rank]], None, None
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[lo:lo + rows[rank]], None, None


def gather_tables(output_dict, rows_per_rank, reduce_grad):
    return {k: AllGatherRows.apply(v, rows_per_rank, reduce_grad) for k, v in output_dict.items()}


def gather_batch_layout(data_dict, device):
    """ONE small integer all-gather per step: every rank's (object rows, |e1i|, |e1j|, |e2j|) -> [world, 4] on the host.
    (Replaces two pickled all_gather_object calls; the read-back of 4*world integers is the step's only host sync.)"""
    world = dist.get_world_size()
    mine = torch.tensor([int(data_dict['tot_obj_pts'].shape[0]), len(data_dict['e1i']), len(data_dict['e1j']),
                         len(data_dict['e2j'])], dtype=torch.int64, device=device)
    if len(data_dict['e1i']) != len(data_dict['e2i']):
        raise RuntimeError('sgaligner_amd: e1i and e2i must have the same length')
    out = torch.empty((world * 4,), dtype=torch.int64, device=device)
    _all_gather_flat(out, mine)
    return out.view(world, 4).cpu().numpy()


def gather_index_sets_device(data_dict, layout, device):
    """Global packed index array [e1i | e2i | e1j | e2j] of ALL ranks, built on the device from one int32 all-gather of
    the ranks' local packed arrays (offset by the ranks' object counts): what ops.IndexSets holds for the global batch.
    Returns (idx int32 [R], A, J1, J2)."""
    world = dist.get_world_size()
    local = np.concatenate([np.asarray(data_dict[k]).astype(np.int32).reshape(-1) for k in ('e1i', 'e2i', 'e1j', 'e2j')])
    lens = layout[:, 1] * 2 + layout[:, 2] + layout[:, 3]
    mx = int(lens.max()) if world else 0
    buf = torch.zeros((max(mx, 1),), dtype=torch.int32, device=device)
    if local.size:
        buf[:local.size] = torch.from_numpy(local).to(device)
    allb = torch.empty((world * max(mx, 1),), dtype=torch.int32, device=device)
    _all_gather_flat(allb, buf)
    allb = allb.view(world, max(mx, 1))
    offs = np.concatenate([[0], np.cumsum(layout[:, 0])])
    parts = [[], [], [], []]
    for r in range(world):
        a, j1, j2 = int(layout[r, 1]), int(layout[r, 2]), int(layout[r, 3])
        cuts = [0, a, 2 * a, 2 * a + j1, 2 * a + j1 + j2]
        for q in range(4):
            if cuts[q + 1] > cuts[q]:
                parts[q].append(allb[r, cuts[q]:cuts[q + 1]] + int(offs[r]))
    flat = [t for q in parts for t in q]
    idx = torch.cat(flat) if flat else torch.zeros((0,), dtype=torch.int32, device=device)
    return idx.contiguous(), int(layout[:, 1].sum()), int(layout[:, 2].sum()), int(layout[:, 3].sum())


def gather_index_sets(data_dict, rows_per_rank):
    """Global e1i/e2i/e1j/e2j (host numpy) from every rank's local ones, offset by the ranks' object counts
    (host-side variant for callers that need numpy arrays, e.g. the M == 1 replica path and the CPU tests)."""
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
