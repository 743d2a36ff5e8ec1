"""GPU parity at BASELINE.json's full size (configs[1]: 512 pairs x 64 objects x 512 points = 65,536 objects).

The oracle cannot run the batch-global loss at this size in seconds, so full-size correctness is pinned through
size-independent properties of the path, plus oracle checks on sampled objects where the computation is
per-object (PointNet), plus a mid-size (64 pairs, 8192 objects) direct comparison with the fp64 oracle:

  * PointNet: max-pool over points => bit-exact invariance under a permutation of each object's points,
    bit-exact equivariance under a permutation of objects, weight gradients additive over object chunks,
    sampled objects equal to the oracle.
  * Loss: two independent HIP implementations (fused multi-table sweeps vs per-table general sweeps) agree;
    cosine similarities are invariant under positive per-row scaling of the embeddings (loss unchanged,
    dE scales by 1/c); anchor-range shards of the loss partial sums add up to the unsharded sums.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PAIRS, NOBJ, NPTS = 512, 64, 512


def _pointnet_weights(seed=0):
    from oracle import sga_oracle as O
    torch.manual_seed(seed)
    p = O.init_params(['point'])
    return [p['object_encoder.conv1.weight'].reshape(64, 3).contiguous(), torch.randn(64) * 0.1,
            p['object_encoder.conv2.weight'].reshape(128, 64).contiguous(), torch.randn(128) * 0.1,
            p['object_encoder.conv3.weight'].reshape(256, 128).contiguous(), torch.randn(256) * 0.1]


def test_fullsize_pointnet_forward_properties_and_sampled_oracle():
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    T = PAIRS * 2 * NOBJ
    ws = _pointnet_weights()
    wd = [w.cuda() for w in ws]
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(T, NPTS, 3, device='cuda', generator=g)
    y, am = ops.pointnet_forward(x, *wd, want_argmax=True)
    # sampled objects against the oracle (the encoder is per-object)
    idx = torch.randperm(T, generator=torch.Generator().manual_seed(1))[:48]
    yo = O.pointnet_feat(x[idx.cuda()].cpu().permute(0, 2, 1), *ws)
    assert (y[idx.cuda()].cpu() - yo).abs().max() < 2e-5
    # permutation of the points of every object: same maxima, bit for bit; winners point at the same coordinates
    perm = torch.randperm(NPTS, device='cuda', generator=g)
    xp = x[:, perm].contiguous()
    y2, am2 = ops.pointnet_forward(xp, *wd, want_argmax=True)
    assert torch.equal(y, y2)
    pos = y > 0
    w1 = torch.gather(x, 1, am.long().clamp(0, NPTS - 1).unsqueeze(-1).expand(-1, -1, 3))
    w2 = torch.gather(xp, 1, am2.long().clamp(0, NPTS - 1).unsqueeze(-1).expand(-1, -1, 3))
    assert (w1[pos] == w2[pos]).float().mean() > 0.9999           # exact ties between two points are the only exception
    del xp, y2, am2, w1, w2
    # permutation of objects: rows move with their objects
    operm = torch.randperm(T, device='cuda', generator=g)
    y3, _ = ops.pointnet_forward(x[operm].contiguous(), *wd, want_argmax=False)
    assert torch.equal(y3, y[operm])


def test_fullsize_pointnet_backward_additive_over_objects_and_oracle_chunk():
    from oracle import sga_oracle as O
    from sgaligner_amd import ops
    T = PAIRS * 2 * NOBJ
    ws = _pointnet_weights(1)
    g = torch.Generator(device='cuda').manual_seed(6)
    x = torch.randn(T, NPTS, 3, device='cuda', generator=g)
    cot = torch.randn(T, 256, device='cuda', generator=g)

    def grads(lo, hi):
        w = [t.cuda().requires_grad_(True) for t in ws]
        y = ops.pointnet(x[lo:hi], *w)
        (y * cot[lo:hi]).sum().backward()
        torch.cuda.synchronize()
        return [t.grad.double() for t in w]

    full = grads(0, T)
    cuts = [0, 9, 20000, 20031, 50000, T]            # ragged chunk sizes, including ones smaller than the launch grid
    parts = [grads(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    for k in range(6):
        s = sum(p[k] for p in parts)
        scale = max(1.0, s.abs().max().item())
        assert (full[k] - s).abs().max().item() < 2e-4 * scale, (k, (full[k] - s).abs().max().item(), scale)
    # the 9-object chunk against the oracle's autograd
    wo = [t.clone().requires_grad_(True) for t in ws]
    yo = O.pointnet_feat(x[:9].cpu().permute(0, 2, 1), *wo)
    (yo * cot[:9].cpu()).sum().backward()
    for k in range(6):
        ref = wo[k].grad.double()
        assert (parts[0][k].cpu() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item()), k


def _loss_setup(B, N, mods, seed, requires_grad=True):
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(B, N, 1, seed=seed)
    T = int(dd['tot_obj_count'].sum())
    g = torch.Generator(device='cuda').manual_seed(seed)
    base = [torch.randn(T, 100, device='cuda', generator=g) for _ in mods]
    return dd, T, base


def _run_overall(base, dd, mods, w0, lv1, lv2, fused, scale=None):
    from sgaligner_amd.aligner import losses as L
    from sgaligner_amd.aligner.sg_aligner import MultiModalFusion
    m = len(mods)
    L.FUSED_JOINT = fused
    try:
        e = {}
        for i, k in enumerate(mods):
            t = base[i] if scale is None else base[i] * scale[i][:, None]
            e[k] = t.clone().requires_grad_(True)
        fus = MultiModalFusion(m).cuda()
        ial, icl = L.CustomMultiLossLayer(m).cuda(), L.CustomMultiLossLayer(m).cuda()
        with torch.no_grad():
            fus.weight.copy_(w0)
            ial.log_vars.copy_(lv1); icl.log_vars.copy_(lv2)
        out = dict(e)
        out['joint'] = fus([e[k] for k in mods])
        fn = L.OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
        res = fn(out, dd)
        res['loss'].backward()
        torch.cuda.synchronize()
        return res['loss'].item(), {k: e[k].grad for k in mods}, fus.weight.grad.clone(), ial.log_vars.grad.clone(), icl.log_vars.grad.clone()
    finally:
        L.FUSED_JOINT = True


def test_fullsize_loss_two_implementations_and_row_scaling_invariance():
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(PAIRS, NOBJ, mods, seed=21)
    assert T == PAIRS * 2 * NOBJ
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
    del gg
    # positive per-row scaling leaves every cosine similarity, hence the loss, unchanged; dE scales by 1/c
    g = torch.Generator(device='cuda').manual_seed(3)
    scale = [0.25 + 3.0 * torch.rand(T, device='cuda', generator=g) for _ in mods]
    ls, gs, gws, _, _ = _run_overall(base, dd, mods, w0, lv1, lv2, fused=True, scale=scale)
    assert abs(ls - lf) < 2e-5 * abs(lf), (ls, lf)
    for i, k in enumerate(mods):
        back = gs[k] * scale[i][:, None]
        sc = gf[k].abs().max().item()
        assert (back - gf[k]).abs().max().item() < 2e-3 * sc, (k, (back - gf[k]).abs().max().item(), sc)
    assert (gws - gwf).abs().max().item() < 2e-3 * max(1e-3, gwf.abs().max().item())


def test_fullsize_loss_anchor_shards_add_up():
    """Forward partial sums of R anchor-range shards (what each rank contributes before the all-reduce) add up
    to the unsharded sums at full size."""
    from sgaligner_amd import ops
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(PAIRS, NOBJ, mods, seed=22)
    w = torch.tensor([[0.5], [1.0], [1.5]], device='cuda')
    with torch.no_grad():
        ref, s = ops.fused_contrastive_terms(base, w, dict(dd))
        A = s.A
        R = 8
        cuts = [A * r // R for r in range(R + 1)]
        first = []                                   # the first all-reduced tensor of every shard = its partial sums

        for r in range(R):
            seen = []

            def reduce(t, seen=seen):
                if not seen:
                    seen.append(t.clone())
            try:
                ops.fused_contrastive_terms(base, w, dict(dd), shard=(cuts[r], cuts[r + 1]), reduce=reduce)
            except Exception:
                if not seen:
                    raise
            first.append(seen[0])
        tot = sum(first)
        with torch.no_grad():
            seen = []
            ops.fused_contrastive_terms(base, w, dict(dd), shard=(0, A), reduce=lambda t: seen.append(t.clone()) if not seen else None)
        assert torch.allclose(tot.double(), seen[0].double(), rtol=1e-6), (tot, seen[0])
    assert torch.isfinite(ref).all()


def test_midsize_loss_vs_fp64_oracle():
    """64 pairs x 64 objects (8192 objects, 19 anchors/pair): the largest batch-global loss the fp64 oracle finishes in seconds."""
    from oracle import sga_oracle as O
    mods = ['point', 'gat', 'rel']
    dd, T, base = _loss_setup(64, NOBJ, mods, seed=23)
    w0 = torch.tensor([[0.7], [1.2], [0.9]], device='cuda')
    lv1 = torch.tensor([0.1, -0.2, 0.05], device='cuda')
    lv2 = torch.tensor([-0.1, 0.15, 0.0], device='cuda')
    lf, gf, gwf, g1f, g2f = _run_overall(base, dd, mods, w0, lv1, lv2, fused=True)
    eo = {k: base[i].cpu().double().requires_grad_(True) for i, k in enumerate(mods)}
    wo = w0.cpu().double().requires_grad_(True)
    lo1, lo2 = lv1.cpu().double().requires_grad_(True), lv2.cpu().double().requires_grad_(True)
    out_o = dict(eo)
    out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
    ref = O.overall_loss(out_o, dd, mods, lo1, lo2)
    ref['loss'].backward()
    assert abs(lf - ref['loss'].item()) < 1e-4 * abs(ref['loss'].item())
    for k in mods:
        gref = eo[k].grad
        err = (gf[k].cpu().double() - gref).abs().max().item()
        assert err < 1e-3 * gref.abs().max().item(), (k, err, gref.abs().max().item())
    assert (gwf.cpu().double() - wo.grad).abs().max().item() < 1e-3 * max(1e-3, wo.grad.abs().max().item())
    assert torch.allclose(g1f.cpu().double(), lo1.grad, rtol=1e-3) and torch.allclose(g2f.cpu().double(), lo2.grad, rtol=1e-3)
