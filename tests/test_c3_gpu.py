"""GPU parity at BASELINE.json configs[2]: 4096 pairs x 128 objects x 512 points, sharded 8 ways = 512 pairs per GPU
(131 072 objects, 1024 graphs of 128 nodes, A = 38 anchors/pair -> 19 456 anchors per shard), plus the whole 4096-pair
batch on ONE GPU (the north-star target config: 1 048 576 objects, A = 155 648, J = 368 640 per side).

As in test_fullsize_gpu.py the oracle cannot run the batch-global loss at these sizes, so: per-object / per-graph parts
are checked against the oracle on samples, the loss through two independent HIP implementations, shard additivity with
the 8-way anchor cuts the ranks use, and invariance under the stash block size of the anchors x anchors backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PAIRS, NOBJ, NPTS, RANKS = 512, 128, 512, 8


def test_c3_shard_pointnet_sampled_oracle_and_chunk_additivity():
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    T = PAIRS * 2 * NOBJ
    torch.manual_seed(0)
    p = O.init_params(['point'])
    ws = [p['object_encoder.conv1.weight'].reshape(64, 3).contiguous(), torch.randn(64) * 0.1,
          p['object_encoder.conv2.weight'].reshape(128, 64).contiguous(), torch.randn(128) * 0.1,
          p['object_encoder.conv3.weight'].reshape(256, 128).contiguous(), torch.randn(256) * 0.1]
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(T, NPTS, 3, device='cuda', generator=g)
    cot = torch.randn(T, 256, device='cuda', generator=g)

    def run(lo, hi):
        wd = [w.cuda().requires_grad_(True) for w in ws]
        y = ops.pointnet(x[lo:hi], *wd)
        (y * cot[lo:hi]).sum().backward()
        return y.detach(), [w.grad for w in wd]

    y, gfull = run(0, T)
    idx = torch.randperm(T, generator=torch.Generator().manual_seed(2))[:64]
    wo = [w.clone().requires_grad_(True) for w in ws]
    yo = O.pointnet_feat(x[idx.cuda()].cpu().permute(0, 2, 1), *wo)
    assert (y[idx.cuda()].cpu() - yo).abs().max() < 2e-5
    # weight gradients are sums over objects: 8 rank-sized chunks add up to the full batch (what the grad all-reduce does)
    parts = [run(T * r // RANKS, T * (r + 1) // RANKS)[1] for r in range(RANKS)]
    for k in range(6):
        tot = sum(pp[k] for pp in parts)
        sc = gfull[k].abs().max().item()
        assert (tot - gfull[k]).abs().max().item() < 2e-4 * max(1.0, sc), k
    # the sampled objects' gradient contribution against the oracle (cotangent restricted to the sample)
    (yo * cot[idx.cuda()].cpu()).sum().backward()
    wd = [w.cuda().requires_grad_(True) for w in ws]
    (ops.pointnet(x[idx.cuda()].contiguous(), *wd) * cot[idx.cuda()]).sum().backward()
    for k in range(6):
        sc = wo[k].grad.abs().max().item()
        assert (wd[k].grad.cpu() - wo[k].grad).abs().max().item() < 1e-3 * max(1.0, sc), k


def test_c3_shard_gat_128_nodes_all_graphs():
    """MultiGAT over the shard's 1024 complete graphs of 128 nodes in one launch per layer: every output row of sampled
    graphs == oracle; parameter gradients of the whole shard == sum over 8 sub-batches, and == oracle on a sub-batch."""
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    G = 2 * PAIRS
    ii, jj = np.meshgrid(np.arange(NOBJ), np.arange(NOBJ), indexing='ij')
    m = ii != jj
    tmpl = torch.from_numpy(np.stack([ii[m], jj[m]], 1).astype(np.int64))
    E1 = tmpl.shape[0]
    edges = tmpl.cuda().repeat(G, 1)
    g = torch.Generator(device='cuda').manual_seed(3)
    x = (torch.rand(G * NOBJ, 3, device='cuda', generator=g) * 6 - 3).double()
    cot = torch.randn(G * NOBJ, 256, device='cuda', generator=g)
    p = O.init_params(['point', 'gat'], dtype=torch.float64, seed=3)
    layers = O._gat_layers(p)
    for l in layers:
        l['bias'] = torch.randn_like(l['bias']) * 0.1

    def run(g0, g1):
        gb = ops.GraphBatch(np.full(g1 - g0, NOBJ), np.full(g1 - g0, E1), edges[g0 * E1:g1 * E1])
        dl = [[l[k].detach().float().cuda().requires_grad_(True) for k in ('lin_w', 'att_src', 'att_dst', 'bias')] for l in layers]
        out = ops.multi_gat(gb, x[g0 * NOBJ:g1 * NOBJ], dl[0], dl[1])
        (out * cot[g0 * NOBJ:g1 * NOBJ]).sum().backward()
        return out.detach(), [t.grad for li in dl for t in li]

    out, gfull = run(0, G)
    assert torch.isfinite(out).all()
    parts = [run(G * r // RANKS, G * (r + 1) // RANKS)[1] for r in range(RANKS)]
    for k in range(8):
        tot = sum(pp[k] for pp in parts)
        sc = gfull[k].abs().max().item()
        assert (tot - gfull[k]).abs().max().item() < 5e-4 * max(1.0, sc), (k, (tot - gfull[k]).abs().max().item(), sc)
    # oracle on sampled graphs (forward) and on one 3-graph sub-batch (forward + every parameter gradient)
    et = tmpl.t()
    for gi in (0, 1, 517, G - 1):
        ref = O.multi_gat(x[gi * NOBJ:(gi + 1) * NOBJ].cpu(), et, layers)
        assert (out[gi * NOBJ:(gi + 1) * NOBJ].cpu().double() - ref).abs().max() < 1e-4, gi
    g0, g1 = 300, 303
    lo = [{k: v.clone().requires_grad_(True) for k, v in l.items()} for l in layers]
    ref = torch.cat([O.multi_gat(x[gi * NOBJ:(gi + 1) * NOBJ].cpu(), et, lo) for gi in range(g0, g1)])
    (ref * cot[g0 * NOBJ:g1 * NOBJ].cpu().double()).sum().backward()
    _, gsub = run(g0, g1)
    names = [(li, k) for li in range(2) for k in ('lin_w', 'att_src', 'att_dst', 'bias')]
    for (li, k), gg in zip(names, gsub):
        gref = lo[li][k].grad
        assert (gg.cpu().double() - gref).abs().max().item() < 1e-3 * max(1.0, gref.abs().max().item()), (li, k)


def _loss_setup(pairs, nobj, mods, seed):
    from sgaligner_amd.synthetic import make_batch_fast
    dd = make_batch_fast(pairs, nobj, 4, seed=seed, device='cuda')        # the loss only needs the index sets
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(seed)
    # unit-ish rows with structure: common objects of a pair are near copies (like trained embeddings), so the A x A
    # terms are not all alike
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in mods]
    return dd, T, base


def test_c3_shard_loss_two_implementations():
    from test_fullsize_gpu import _run_overall
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(PAIRS, NOBJ, mods, seed=31)
    assert T == PAIRS * 2 * NOBJ and len(dd['e1i']) == PAIRS * 38
    w0 = torch.tensor([[0.7], [1.2], [0.9]], device='cuda')
    lv1 = torch.tensor([0.1, -0.2, 0.05], device='cuda')
    lv2 = torch.tensor([-0.1, 0.15, 0.0], device='cuda')
    lf, gf, gwf, g1f, g2f = _run_overall(base, dd, mods, w0, lv1, lv2, fused=True)
    lg, gg, gwg, g1g, g2g = _run_overall(base, dd, mods, w0, lv1, lv2, fused=False)
    assert np.isfinite(lf) and abs(lf - lg) < 1e-5 * abs(lg), (lf, lg)
    for k in mods:
        sc = gg[k].abs().max().item()
        assert (gf[k] - gg[k]).abs().max().item() < 1e-3 * sc, (k, (gf[k] - gg[k]).abs().max().item(), sc)
    assert (gwf - gwg).abs().max().item() < 1e-3 * max(1e-3, gwg.abs().max().item())
    assert torch.allclose(g1f, g1g, rtol=1e-4) and torch.allclose(g2f, g2g, rtol=1e-4)


def _replay_sharded(base, w0, cot, dd, cuts, general=False):
    """Deterministic replay of the all-reduces inside ops.FusedContrastiveFn (general=True: ops.ContrastiveTermsFn, the per-table path; w0
    unused) for len(cuts)-1 simulated ranks (see test_modules_gpu.py::test_anchor_sharded_loss_equals_unsharded)."""
    from sgaligner_amd import ops
    R = len(cuts) - 1
    M = len(base)

    def run(shard, reduce):
        tabs = [b.clone().requires_grad_(True) for b in base]
        w = w0.clone().requires_grad_(True)
        if general:
            sums, s = ops.contrastive_terms(tabs, dd, shard=shard, reduce=reduce)
            (sums * cot).sum().backward()
            return sums.detach(), [t.grad for t in tabs], torch.zeros_like(w0)
        sums, s = ops.fused_contrastive_terms(tabs, w, dd, shard=shard, reduce=reduce)
        (sums * cot).sum().backward()
        return sums.detach(), [t.grad for t in tabs], w.grad

    totals, n_reduces, results = [], 3, None
    for rnd in range(n_reduces + 1):
        partial = [None] * R
        results = []
        for rank in range(R):
            state = {'n': 0}

            def reduce(t, rank=rank, state=state):
                n = state['n']
                state['n'] += 1
                if n < len(totals):
                    t.copy_(totals[n])
                elif n == len(totals):
                    partial[rank] = t.clone()
            res = run((cuts[rank], cuts[rank + 1]), reduce)
            if rnd == n_reduces:                       # keep only the final round's gradients, summed on the fly
                if not results:
                    results = [res[0], [g.clone() for g in res[1]], res[2].clone(), [res[0]]]
                else:
                    for a, b in zip(results[1], res[1]):
                        a += b
                    results[2] += res[2]
                    results[3].append(res[0])
        if rnd < n_reduces:
            assert all(p is not None for p in partial), rnd
            totals.append(sum(partial))
    return results


def test_c3_shard_loss_8way_anchor_shards_equal_unsharded():
    """The 8 anchor ranges the ranks of configs[2] own (512-pair shard shape here; rank r owns the anchors of its pairs):
    every rank ends with the global loss terms, and the ranks' dL/dE and dL/dbeta shares sum to the unsharded gradients."""
    from sgaligner_amd import ops
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(PAIRS, NOBJ, mods, seed=32)
    M = len(mods)
    w0 = torch.tensor([[0.3], [1.1], [-0.4]], device='cuda')
    cot = torch.randn(M + 1 + 2 * M, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
    tabs = [b.clone().requires_grad_(True) for b in base]
    w = w0.clone().requires_grad_(True)
    ref_sums, s = ops.fused_contrastive_terms(tabs, w, dd)
    (ref_sums * cot).sum().backward()
    A = s.A
    per_pair = A // PAIRS
    cuts = [per_pair * (PAIRS * r // RANKS) for r in range(RANKS + 1)]      # rank r owns the anchors of ITS pairs
    sums0, grads, gw, all_sums = _replay_sharded(base, w0, cot, dd, cuts)
    for sr in all_sums:
        assert torch.allclose(sr, ref_sums.detach(), rtol=1e-4, atol=1e-5)
    for m in range(M):
        sc = tabs[m].grad.abs().max().item()
        assert (grads[m] - tabs[m].grad).abs().max().item() < 2e-4 * max(1.0, sc), m
    assert (gw - w.grad).abs().max().item() < 2e-4 * max(1.0, w.grad.abs().max().item())


def test_stash_block_size_does_not_change_the_gradient():
    """The anchors x anchors backward walks anchor-row blocks whose coefficient stash is bounded by ops.STASH_BYTES; the
    result must not depend on the block size (1 block vs ~20 blocks; fused and general path)."""
    from sgaligner_amd import ops
    from test_fullsize_gpu import _run_overall
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(96, NOBJ, mods, seed=33)                       # A = 3648
    w0 = torch.tensor([[0.7], [1.2], [0.9]], device='cuda')
    lv1 = torch.tensor([0.1, -0.2, 0.05], device='cuda')
    lv2 = torch.tensor([-0.1, 0.15, 0.0], device='cuda')
    keep = ops.STASH_BYTES
    try:
        for fused in (True, False):
            ops.STASH_BYTES = 1 << 40
            assert len(ops._anchor_chunks(0, 3648, 3648, 3)) == 1
            l1, g1, gw1, a1, b1 = _run_overall(base, dd, mods, w0, lv1, lv2, fused=fused)
            ops.STASH_BYTES = 4 * 3648 * 3 * 200                                 # ~192-row blocks (+ a ragged last one)
            assert len(ops._anchor_chunks(0, 3648, 3648, 3)) >= 19
            l2, g2, gw2, a2, b2 = _run_overall(base, dd, mods, w0, lv1, lv2, fused=fused)
            assert l1 == l2
            for k in mods:
                sc = g1[k].abs().max().item()
                assert (g1[k] - g2[k]).abs().max().item() < 1e-5 * sc, (fused, k)
            assert (gw1 - gw2).abs().max().item() < 1e-5 * max(1e-3, gw1.abs().max().item())
            assert torch.allclose(a1, a2, rtol=1e-5) and torch.allclose(b1, b2, rtol=1e-5)
    finally:
        ops.STASH_BYTES = keep


def test_c3_full_batch_runs_on_one_gpu():
    """North-star target config on ONE GPU: 4096 pairs x 128 objects x 512 points, P+S+R, batch-global loss.
    Must fit (the A x A backward never holds more than ops.STASH_BYTES of stash: a full one would be 3 x 97 GB), give finite
    loss / gradients, and agree with the sum of the 8 ranks' partial loss sums on the same embeddings."""
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch_fast
    from sgaligner_amd.trainer import AlignerSteps
    mods = ['point', 'gat', 'rel']
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    dd = make_batch_fast(4096, NOBJ, NPTS, seed=44, device='cuda')
    T = int(dd['tot_obj_pts'].shape[0])
    assert T == 4096 * 2 * NOBJ and len(dd['e1i']) == 4096 * 38 and dd['edges'].shape[0] == 8192 * NOBJ * (NOBJ - 1)
    steps = AlignerSteps(mods, device='cuda', seed=42)
    out, loss = steps.forward_backward(dd)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert peak < 120.0, f'peak {peak:.1f} GiB'
    assert np.isfinite(loss['loss'].item())
    for name, prm in steps.model.named_parameters():
        if name.startswith('object_encoder.bn') or 'meta_embedding_attr' in name:
            continue
        assert prm.grad is not None and torch.isfinite(prm.grad).all() and prm.grad.abs().max() > 0, name
    # per-object part on samples of the full batch against the oracle
    from oracle import sga_oracle as O
    sd = {k: v.detach().cpu() for k, v in steps.model.state_dict().items()}
    idx = torch.randperm(T, generator=torch.Generator().manual_seed(5))[:32]
    ws = [sd['object_encoder.conv1.weight'].reshape(64, 3), sd['object_encoder.conv1.bias'],
          sd['object_encoder.conv2.weight'].reshape(128, 64), sd['object_encoder.conv2.bias'],
          sd['object_encoder.conv3.weight'].reshape(256, 128), sd['object_encoder.conv3.bias']]
    feat = O.pointnet_feat(dd['tot_obj_pts'][idx.cuda()].cpu().permute(0, 2, 1), *ws)
    emb_o = feat @ sd['object_embedding.weight'].t() + sd['object_embedding.bias']
    assert (out['point'][idx.cuda()].detach().cpu() - emb_o).abs().max() < 1e-3
    # loss terms == sum of the 8 ranks' partial sums (rank r owns the anchors of its 512 pairs)
    tabs = [out[m].detach() for m in mods]
    w = steps.model.fusion.weight.detach()
    del out, loss
    with torch.no_grad():
        ref, s = ops.fused_contrastive_terms(tabs, w, dd)
        A = s.A
        cuts = [A * r // RANKS for r in range(RANKS + 1)]
        firsts, seconds = [], []
        for r in range(RANKS):
            seen = []
            ops.fused_contrastive_terms(tabs, w, dd, shard=(cuts[r], cuts[r + 1]), reduce=lambda t, seen=seen: seen.append(t.clone()))
            firsts.append(seen[0])
        tot = sum(firsts)
        seen = []
        ops.fused_contrastive_terms(tabs, w, dd, shard=(0, A), reduce=lambda t: seen.append(t.clone()))
        assert torch.allclose(tot, seen[0], rtol=1e-6), (tot, seen[0])
    assert torch.isfinite(ref).all()
