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


# bench.py sets this to [] for the timed steps: every data-path collective appends (kind, shape, dtype, bytes in, bytes out), from
# which `collective_summary` builds the bench line's `collectives` object (what travelled per step, and how long each takes).
COLLECTIVE_EVENTS = None


def _log(kind, inp, out):
    if COLLECTIVE_EVENTS is not None:
        COLLECTIVE_EVENTS.append((kind, tuple(inp.shape), str(inp.dtype), inp.numel() * inp.element_size(),
                                  out.numel() * out.element_size()))


def _all_gather_flat(out, inp, async_op=False):
    """out [world * n, ...] = cat of every rank's inp [n, ...].  RCCL: one all_gather_into_tensor; gloo (CPU and the
    shared-GPU tests): the list form, which it supports for host and device tensors alike.  async_op: returns the work handle
    (the collective runs on the backend's own stream behind everything queued so far; `.wait()` orders the caller's stream after it)."""
    _log('all_gather', inp, out)
    if dist.get_backend() == 'nccl':
        return dist.all_gather_into_tensor(out, inp, async_op=async_op)
    n = inp.shape[0]
    return dist.all_gather([out[r * n:(r + 1) * n] for r in range(dist.get_world_size())], inp, async_op=async_op)


def _has_reduce_scatter():
    return dist.get_backend() == 'nccl'          # RCCL; gloo (CPU / shared-GPU tests) has no reduce-scatter


class AllGatherRows(torch.autograd.Function):
    """out = cat_r(x_r) along dim 0 (ranks may hold different row counts).  Backward: every rank holds
    dL/d(out) of ITS OWN loss replica/shard; the gradient of the summed objective wrt x_r is the sum over ranks of
    the corresponding row block -> ONE reduce-scatter (RCCL), each rank receiving only its own rows: half the bytes on
    the wire of all-reduce + slice (2.5 GB -> 1.26 GB per rank at configs[2]).  gloo falls back to all-reduce + slice."""

    @staticmethod
    def forward(ctx, x, rows_per_rank, reduce_grad, works=None):
        """works: a list -> with equal shards the gather is launched ASYNCHRONOUSLY (it runs on the backend's stream under whatever
        the caller queues next) and its handle appended; the returned tensor must not be read before `handle.wait()`
        (EarlyGather.wait).  Ragged shards and works=None take the blocking path."""
        world = dist.get_world_size()
        rank = dist.get_rank()
        ctx.rows, ctx.rank, ctx.reduce_grad = list(rows_per_rank), rank, reduce_grad
        mx = max(ctx.rows)
        pad = x
        if x.shape[0] < mx:
            pad = torch.cat([x, x.new_zeros((mx - x.shape[0],) + tuple(x.shape[1:]))])
        if all(n == mx for n in ctx.rows):           # the usual case (equal shards): gather straight into the result
            out = torch.empty((world * mx,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
            w = _all_gather_flat(out, pad.contiguous(), async_op=works is not None)
            if works is not None:
                works.append(w)
            return out
        bufs = [torch.empty_like(pad) for _ in range(world)]
        _log('all_gather', pad, pad.new_empty((world,) + tuple(pad.shape)))
        dist.all_gather(bufs, pad.contiguous())
        return torch.cat([b[:n] for b, n in zip(bufs, ctx.rows)])

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        rows, rank = ctx.rows, ctx.rank
        lo = sum(rows[:rank])
        if not ctx.reduce_grad:
            return g[lo:lo + rows[rank]], None, None, None
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
            _log('reduce_scatter', inp, out)
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)
            return out[:rows[rank]], None, None, None
        _log('all_reduce', g, g)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[lo:lo + rows[rank]], None, None, None


def gather_tables(output_dict, rows_per_rank, reduce_grad):
    return {k: AllGatherRows.apply(v, rows_per_rank, reduce_grad) for k, v in output_dict.items()}


class EarlyGather:
    """Overlap of the loss's table all-gathers with the encoder: installed as `MultiModalEncoder._on_table`, it is
    called with every modality table the moment the encoder has produced it and launches that table's all-gather asynchronously --
    the encoder evaluates its cheap modalities (rel / attr / gat) first, so their tables travel over xGMI while the object encoder
    (PointNet, ~all of the encoder's time) runs, and only the last table's gather is exposed.  `tables()` waits for all of them.
    Backward is AllGatherRows.backward per table (one reduce-scatter each), as on the blocking path: same numbers."""

    def __init__(self, rows_per_rank):
        self.rows = list(rows_per_rank)
        self.out, self.works = {}, []

    def __call__(self, module, emb):
        self.out[module] = AllGatherRows.apply(emb, self.rows, True, self.works)

    def tables(self, modules):
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []
        return {m: self.out[m] for m in modules}

    def drain(self):
        """Wait for whatever is still in flight (the encoder raised after some hooks had fired): an outstanding asynchronous all-gather must
        not be dropped un-waited -- its output buffer would be freed under the collective."""
        for w in self.works:
            if w is not None:
                try:
                    w.wait()
                except Exception:
                    pass
        self.works = []


def known_layout(rows, n_e1i, n_e1j, n_e2j, world):
    """[world, 4] layout array for batches whose per-rank shape is known without communication (bench.py's uniform synthetic
    shards; a trainer that shards one global collated batch knows every rank's slice).  Put it into the data_dict as
    '_sga_layout': the step then has NO host synchronisation (gather_batch_layout's read-back is skipped)."""
    return np.tile(np.asarray([[int(rows), int(n_e1i), int(n_e1j), int(n_e2j)]], dtype=np.int64), (int(world), 1))


def layout_of(data_dict, device):
    pre = data_dict.get('_sga_layout') if isinstance(data_dict, dict) else None
    if pre is not None:
        pre = np.asarray(pre, dtype=np.int64).reshape(-1, 4)
        if pre.shape[0] != dist.get_world_size():
            raise RuntimeError('sgaligner_amd: _sga_layout must have one row per rank')
        r = dist.get_rank()
        mine = (int(data_dict['tot_obj_pts'].shape[0]), len(data_dict['e1i']), len(data_dict['e1j']), len(data_dict['e2j']))
        if tuple(int(v) for v in pre[r]) != mine:
            raise RuntimeError(f'sgaligner_amd: _sga_layout row {r} = {tuple(pre[r])} does not describe this rank\'s batch {mine}')
        return pre
    return gather_batch_layout(data_dict, device)


def collective_summary(events, n_steps, device, repeats=3):
    """bench.py's `collectives` object (N > 1): what the timed steps moved -- per kind: calls and bytes per step on this rank -- and
    how long ONE call of each distinct (kind, shape) takes when run alone (blocking, HIP events on the current stream, mean of
    `repeats` after one warm-up), + backend and the ranks / devices that took part.  Collective: every rank must call it."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per_kind, shapes = {}, {}
    for kind, shape, dtype, b_in, b_out in (events or []):
        d = per_kind.setdefault(kind, {'calls_per_step': 0.0, 'bytes_in_per_step': 0.0, 'bytes_out_per_step': 0.0})
        d['calls_per_step'] += 1.0 / max(1, n_steps)
        d['bytes_in_per_step'] += b_in / max(1, n_steps)
        d['bytes_out_per_step'] += b_out / max(1, n_steps)
        shapes.setdefault((kind, shape, dtype), 0)
        shapes[(kind, shape, dtype)] += 1
    timed = []
    cuda = torch.device(device).type == 'cuda'
    for (kind, shape, dtype, ), cnt in sorted(shapes.items(), key=lambda kv: str(kv[0])):
        dt = getattr(torch, dtype.replace('torch.', ''))
        inp = torch.zeros(shape, device=device, dtype=dt)
        if kind == 'all_gather':
            out = torch.empty((world * shape[0],) + tuple(shape[1:]), device=device, dtype=dt)
            run = lambda: _all_gather_flat(out, inp)
        elif kind == 'reduce_scatter':
            out = torch.empty((shape[0] // world,) + tuple(shape[1:]), device=device, dtype=dt)
            run = lambda: dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)
        else:
            out = inp
            run = lambda: dist.all_reduce(inp, op=dist.ReduceOp.SUM)
        global COLLECTIVE_EVENTS
        saved, COLLECTIVE_EVENTS = COLLECTIVE_EVENTS, None
        try:
            run()
            ms = []
            for _ in range(repeats):
                if cuda:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    dist.barrier()
                    e0.record()
                    run()
                    e1.record()
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                else:
                    import time
                    dist.barrier()
                    t0 = time.perf_counter()
                    run()
                    ms.append((time.perf_counter() - t0) * 1e3)
        finally:
            COLLECTIVE_EVENTS = saved
        nbytes = inp.numel() * inp.element_size()
        timed.append({'kind': kind, 'shape': list(shape), 'dtype': dtype, 'calls_in_timed_steps': cnt, 'bytes_in': nbytes,
                      'bytes_out': out.numel() * out.element_size(), 'ms_each': round(float(np.mean(ms)), 4)})
    seen = [None] * world
    name = torch.cuda.get_device_name(torch.cuda.current_device()) if cuda else 'cpu'
    dist.all_gather_object(seen, {'rank': rank, 'device': str(device), 'name': name})
    out = {'backend': dist.get_backend(), 'world_size': world, 'ranks_seen': seen, 'per_step_this_rank': per_kind, 'timed_alone': timed,
           'timing': f'HIP events on the launch stream, each distinct call run alone {repeats}x after a warm-up (not inside the step)'}
    for kind in ('all_gather', 'reduce_scatter', 'all_reduce'):
        out[kind + '_bytes'] = int(per_kind.get(kind, {}).get('bytes_out_per_step' if kind == 'all_gather' else 'bytes_in_per_step', 0))
    return out


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
    return out.view(world, 4).cpu().numpy()           # host sync; skipped when the caller supplies '_sga_layout' (layout_of)


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
    _log('all_reduce', flat, flat)
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
