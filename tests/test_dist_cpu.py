"""CPU, world_size 2, gloo: the data-parallel plumbing of the path (pair sharding, table all-gather with
its backward, global index sets, flat gradient all-reduce).  Kernels are not involved (no GPU here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import sga_oracle as O
    from sgaligner_amd import dist as sdist
    from sgaligner_amd.synthetic import make_batch
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # every rank owns a contiguous block of pairs of ONE global batch
        n_pairs = 5
        full = make_batch(n_pairs, 6, 8, seed=9, ragged=True)
        lo, hi = sdist.shard_range(n_pairs, rank, world)
        offs = np.concatenate([[0], np.cumsum(full['tot_obj_count'])])
        aoff = np.concatenate([[0], np.cumsum(full['e1i_count'])])
        joff = np.concatenate([[0], np.cumsum(full['e1j_count'])])
        j2off = np.concatenate([[0], np.cumsum(full['e2j_count'])])
        o0, o1 = offs[lo], offs[hi]
        local = {'tot_obj_pts': full['tot_obj_pts'][o0:o1],
                 'e1i': full['e1i'][aoff[lo]:aoff[hi]] - o0, 'e2i': full['e2i'][aoff[lo]:aoff[hi]] - o0,
                 'e1j': full['e1j'][joff[lo]:joff[hi]] - o0, 'e2j': full['e2j'][j2off[lo]:j2off[hi]] - o0}
        rows = [None] * world
        dist.all_gather_object(rows, int(o1 - o0))
        # (1) global index sets rebuilt from the local ones
        gidx = sdist.gather_index_sets(local, rows)
        for k in ('e1i', 'e2i', 'e1j', 'e2j'):
            assert np.array_equal(gidx[k], full[k]), k
        # (1b) the per-step variant: one int all-gather for the layout, one for the packed index arrays
        layout = sdist.gather_batch_layout(local, 'cpu')
        assert layout[:, 0].tolist() == rows and int(layout[:, 1].sum()) == len(full['e1i'])
        idx, A, J1, J2 = sdist.gather_index_sets_device(local, layout, 'cpu')
        assert (A, J1, J2) == (len(full['e1i']), len(full['e1j']), len(full['e2j']))
        assert np.array_equal(idx.numpy(), np.concatenate([full[k] for k in ('e1i', 'e2i', 'e1j', 'e2j')]))
        # (2) table all-gather + loss replica + backward into own rows == single-process gradient
        torch.manual_seed(0)
        table = torch.randn(int(offs[-1]), 16, dtype=torch.float64)
        w = torch.randn(16, 16, dtype=torch.float64)             # a 'parameter' shared by all ranks
        ref_t = table.clone().requires_grad_(True)
        ref_w = w.clone().requires_grad_(True)
        O.icl_loss(ref_t @ ref_w, full).backward()
        mine = table[o0:o1].clone().requires_grad_(True)
        my_w = w.clone().requires_grad_(True)
        gathered = sdist.AllGatherRows.apply(mine @ my_w, rows, False)
        O.icl_loss(gathered, gidx).backward()
        assert torch.allclose(mine.grad, ref_t.grad[o0:o1], atol=1e-12)
        sdist.allreduce_grads([my_w])                             # one flat message, summed over ranks
        assert torch.allclose(my_w.grad, ref_w.grad, atol=1e-10)
        # (3) sharded-loss variant: gradients of a SUM of per-rank partial losses need the reduce in backward
        mine2 = table[o0:o1].clone().requires_grad_(True)
        g2 = sdist.AllGatherRows.apply(mine2, rows, True)
        part = (g2 ** 2).sum() * (rank + 1)                       # different partial objective per rank
        part.backward()
        expect = 2 * table[o0:o1] * sum(r + 1 for r in range(world))
        assert torch.allclose(mine2.grad, expect, atol=1e-12)
        # (4) the overlapped path: equal shards, every table's gather launched asynchronously by dist.EarlyGather (as
        # MultiModalEncoder does through its table hook), values and gradients identical to the blocking gather; the step's
        # collectives logged and summarised (bench.py's `collectives` object); a known layout replaces the layout all-gather.
        torch.manual_seed(1)
        tabs = {m: torch.randn(world * 6, 8, dtype=torch.float64) for m in ('rel', 'gat', 'point')}
        sdist.COLLECTIVE_EVENTS = []
        eg = sdist.EarlyGather([6] * world)
        loc = {m: t[rank * 6:(rank + 1) * 6].clone().requires_grad_(True) for m, t in tabs.items()}
        for m in ('rel', 'gat', 'point'):
            eg(m, loc[m] * 2.0)
        got = eg.tables(['point', 'gat', 'rel'])
        assert list(got) == ['point', 'gat', 'rel'] and not eg.works
        sum((got[m] ** 2).sum() * (rank + 1) for m in got).backward()
        for m in tabs:
            assert torch.equal(got[m].detach(), tabs[m] * 2.0), m
            assert torch.allclose(loc[m].grad, 8 * tabs[m][rank * 6:(rank + 1) * 6] * sum(r + 1 for r in range(world)), atol=1e-12), m
        ev, sdist.COLLECTIVE_EVENTS = sdist.COLLECTIVE_EVENTS, None
        summ = sdist.collective_summary(ev, 1, 'cpu', repeats=2)
        assert summ['backend'] == 'gloo' and summ['world_size'] == world and [r['rank'] for r in summ['ranks_seen']] == list(range(world))
        assert summ['per_step_this_rank']['all_gather']['calls_per_step'] == 3 and summ['all_gather_bytes'] == 3 * world * 6 * 8 * 8
        assert summ['per_step_this_rank']['all_reduce']['calls_per_step'] == 3          # gloo: all-reduce + slice instead of reduce-scatter
        assert all(t['ms_each'] >= 0 for t in summ['timed_alone']) and len(summ['timed_alone']) == 2
        lay = sdist.known_layout(6, 2, 4, 4, world)
        dd = {'tot_obj_pts': torch.zeros(6, 4, 3), 'e1i': np.zeros(2), 'e2i': np.zeros(2), 'e1j': np.zeros(4), 'e2j': np.zeros(4), '_sga_layout': lay}
        assert np.array_equal(sdist.layout_of(dd, 'cpu'), lay)
        del dd['_sga_layout']
        assert np.array_equal(sdist.layout_of(dd, 'cpu'), lay)                            # the gathered one agrees
        dd['_sga_layout'] = sdist.known_layout(7, 2, 4, 4, world)
        try:
            sdist.layout_of(dd, 'cpu')
            raise AssertionError('a layout that does not describe the batch must be refused')
        except RuntimeError:
            pass
        out[rank] = 'ok'
    finally:
        dist.destroy_process_group()


def test_data_parallel_plumbing_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 'ok', 1: 'ok'}


def test_shard_range_partitions():
    from sgaligner_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
