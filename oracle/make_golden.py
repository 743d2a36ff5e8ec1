#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE ITSELF (build container only).

Run:  python oracle/make_golden.py            (needs /root/reference; never runs on the GPU box)

The reference (pure Python) imports here with three sys.modules stubs for packages that are absent
and unused on this path: `torchsummary` (pointnet.py:11, __main__ only), `pointnet2_ops`
(pct.py:6, PCT/SG only) and `torch_geometric` (gat.py:4).  The torch_geometric stub's GATConv is the
oracle's restatement of PyG 2.2.0 -- so every vector that touches 'gat' is labelled GAT-UNPINNED.
Only inputs/outputs (data) are written; no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('SGA_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

from oracle import sga_oracle as O  # noqa: E402


# ---------------------------------------------------------------------------- stubs + import
class _StubGATConv(nn.Module):
    """PyG-2.2.0-shaped GATConv whose forward is oracle.gat_conv (GAT-UNPINNED)."""

    def __init__(self, in_channels, out_channels, heads=1):
        super().__init__()
        self.lin_src = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        nn.init.xavier_uniform_(self.lin_src.weight)
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)

    def forward(self, x, edge_index):
        return O.gat_conv(x, edge_index, self.lin_src.weight, self.att_src, self.att_dst, self.bias)


def import_reference():
    ts = types.ModuleType('torchsummary'); ts.summary = lambda *a, **k: None
    p2 = types.ModuleType('pointnet2_ops'); p2u = types.ModuleType('pointnet2_ops.pointnet2_utils')
    p2.pointnet2_utils = p2u
    tg = types.ModuleType('torch_geometric'); tgn = types.ModuleType('torch_geometric.nn')
    tgn.GATConv = _StubGATConv; tgn.GCNConv = _StubGATConv; tg.nn = tgn
    sys.modules.update({'torchsummary': ts, 'pointnet2_ops': p2, 'pointnet2_ops.pointnet2_utils': p2u,
                        'torch_geometric': tg, 'torch_geometric.nn': tgn})
    sys.path.insert(0, os.path.join(REF, 'src'))
    sys.path.insert(0, REF)
    import aligner.losses as losses
    import aligner.networks.pointnet as pointnet
    import aligner.sg_aligner as sg_aligner
    import utils.alignment as alignment
    return losses, pointnet, sg_aligner, alignment


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **conv)
    print('wrote', path, sum(a.nbytes for a in conv.values()) // 1024, 'KiB')


def index_sets(rng, counts, n_anchor):
    """Index arrays built the way scan3r.py:102-107,142-173 builds them (anchors = first ids of the
    common set, negatives = the rest, e2* offset by N_src, then batch offsets)."""
    e1i, e2i, e1j, e2j, off = [], [], [], [], 0
    for (ns, nr), a in zip(counts, n_anchor):
        perm_s = rng.permutation(ns)[:a]
        perm_r = rng.permutation(nr)[:a]
        e1i += list(perm_s + off)
        e2i += list(perm_r + ns + off)
        e1j += [i + off for i in range(ns) if i not in set(perm_s)]
        e2j += [i + ns + off for i in range(nr) if i not in set(perm_r)]
        off += ns + nr
    f = lambda x: np.asarray(x, dtype=np.int32)
    return dict(e1i=f(e1i), e2i=f(e2i), e1j=f(e1j), e2j=f(e2j)), off


# ---------------------------------------------------------------------------- generators
def gen_pointnet(pointnet):
    for tag, (t, p) in {'small': (8, 64), 'ragged': (5, 37)}.items():
        torch.manual_seed(7)
        net = pointnet.PointNetfeat(global_feat=True, batch_norm=True, point_size=3, input_transform=False,
                                    feature_transform=False, out_size=256)
        with torch.no_grad():                 # non-zero biases so the bias path is exercised
            for c in (net.conv1, net.conv2, net.conv3):
                c.bias.normal_(0, 0.1)
        x = torch.randn(t, 3, p)
        cot = torch.randn(t, 256)
        net.train()
        y = net(x)
        (y * cot).sum().backward()
        net.eval()
        y_eval = net(x)
        npz(f'pointnet_{tag}', x=x, cot=cot, y=y, y_eval=y_eval,
            w1=net.conv1.weight, b1=net.conv1.bias, w2=net.conv2.weight, b2=net.conv2.bias,
            w3=net.conv3.weight, b3=net.conv3.bias,
            gw1=net.conv1.weight.grad, gb1=net.conv1.bias.grad, gw2=net.conv2.weight.grad,
            gb2=net.conv2.bias.grad, gw3=net.conv3.weight.grad, gb3=net.conv3.bias.grad,
            rm1=net.bn1.running_mean, rv1=net.bn1.running_var, rm2=net.bn2.running_mean,
            rv2=net.bn2.running_var, rm3=net.bn3.running_mean, rv3=net.bn3.running_var,
            bn_w_grad_is_none=np.array([net.bn1.weight.grad is None]))


def gen_fusion(sg_aligner):
    for m in (2, 3, 4):
        torch.manual_seed(10 + m)
        fus = sg_aligner.MultiModalFusion(modal_num=m, with_weight=1)
        with torch.no_grad():
            fus.weight.normal_(1.0, 0.5)
        embs = [torch.randn(13, 100, requires_grad=True) for _ in range(m)]
        cot = torch.randn(13, 100 * m)
        j = fus(embs)
        (j * cot).sum().backward()
        arrs = dict(weight=fus.weight, cot=cot, joint=j, gweight=fus.weight.grad)
        for i, e in enumerate(embs):
            arrs[f'emb{i}'] = e
            arrs[f'gemb{i}'] = e.grad
        npz(f'fusion_m{m}', **arrs)


def gen_losses(losses):
    rng = np.random.default_rng(3)
    for tag, counts, n_anchor, mods in (('b1', [(9, 7)], [3], ['point', 'gat']),
                                        ('b2', [(12, 10), (8, 11)], [4, 3], ['point', 'gat', 'rel']),
                                        ('b4', [(6, 5), (7, 9), (10, 6), (5, 5)], [2, 2, 3, 2],
                                         ['point', 'gat', 'rel', 'attr'])):
        idx, t = index_sets(rng, counts, n_anchor)
        torch.manual_seed(100 + t)
        m = len(mods)
        out = {k: (0.5 * torch.randn(t, 100)).requires_grad_(True) for k in mods}
        out['joint'] = (0.3 * torch.randn(t, 100 * m)).requires_grad_(True)
        dd = dict(idx)
        ial_layer = losses.CustomMultiLossLayer(loss_num=m)
        icl_layer = losses.CustomMultiLossLayer(loss_num=m)
        with torch.no_grad():
            ial_layer.log_vars.normal_(0, 0.3)
            icl_layer.log_vars.normal_(0, 0.3)
        ol = losses.OverallLoss(ial_layer, icl_layer, 'cpu', {'zoom': 0.1, 'wt_align_loss': 1.0,
                                                             'wt_contrastive_loss': 1.0, 'modules': mods})
        res = ol(out, dd)
        res['loss'].backward()
        arrs = dict(idx)
        arrs.update(modules=np.array(mods), loss=res['loss'], icl_uni=res['icl_loss_unimodal'],
                    icl_multi=res['icl_loss_multimodal'], ial=res['ial_loss'],
                    lv_ial=ial_layer.log_vars, lv_icl=icl_layer.log_vars,
                    g_lv_ial=ial_layer.log_vars.grad, g_lv_icl=icl_layer.log_vars.grad)
        for k, v in out.items():
            arrs['emb_' + k] = v
            arrs['g_' + k] = v.grad
        # individual pieces
        e = torch.nn.functional.normalize(out[mods[0]].detach(), dim=1)
        q = losses.calculate_prob_dist(e[idx['e1i']], e[idx['e2i']], e[idx['e1j']], e[idx['e2j']], 0.1)
        arrs['q_first_t01'] = q
        arrs['icl_first'] = losses.ICLLoss('cpu')(out[mods[0]].detach(), dd)
        arrs['ial_first'] = losses.IALLoss('cpu')(out[mods[0]].detach(), out['joint'].detach(), dd)
        npz(f'losses_{tag}', **arrs)
    # M == 1 branch of OverallLoss (losses.py:137-138,146)
    idx, t = index_sets(rng, [(11, 9)], [4])
    torch.manual_seed(5)
    emb = torch.randn(t, 100, requires_grad=True)
    ol = losses.OverallLoss(losses.CustomMultiLossLayer(1), losses.CustomMultiLossLayer(1), 'cpu',
                            {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': ['point']})
    res = ol({'point': emb}, dict(idx))
    res['loss'].backward()
    npz('losses_m1', emb_point=emb, g_point=emb.grad, loss=res['loss'], **idx)


def fps(points, npoint, rng):
    """Restatement of utils/point_cloud.py:61-89 with an explicit rng (the reference uses np.random)."""
    n = points.shape[0]
    if n < npoint:
        return points[rng.choice(n, npoint)]
    cent = np.zeros(npoint, dtype=np.int64)
    dist = np.full(n, 1e10)
    far = int(rng.integers(0, n))
    for i in range(npoint):
        cent[i] = far
        d = ((points - points[far]) ** 2).sum(-1)
        dist = np.minimum(dist, d)
        far = int(np.argmax(dist))
    return points[cent]


def gen_example_pair(losses, sg_aligner, alignment):
    """c1: example_data scene_1/scene_2, 256 pts/object, modules=['point'] (SURVEY.md 8c recipe)."""
    rng = np.random.default_rng(42)
    scenes = []
    for s in ('scene_1', 'scene_2'):
        d = np.load(os.path.join(REF, 'example_data', s, 'data.npy'))
        xyz = np.stack([d['x'], d['y'], d['z']], 1).astype(np.float64)
        objs = {}
        for oid in np.unique(d['objectId']):
            pts = xyz[d['objectId'] == oid]
            if pts.shape[0] >= 50:                                   # preprocess.py:90
                objs[int(oid)] = fps(pts, 256, rng)
        scenes.append((xyz, objs))
    (sxyz, sobj), (rxyz, robj) = scenes
    center = sxyz.mean(0)                                            # scan3r.py:76
    sid, rid = sorted(sobj), sorted(robj)
    anchors = [i for i in sid if i != 0 and i in robj]               # scan3r.py:86-87
    pts = np.concatenate([np.stack([sobj[i] for i in sid]), np.stack([robj[i] for i in rid])]) - center
    ns, nr = len(sid), len(rid)
    e1i = np.array([sid.index(a) for a in anchors], dtype=np.int32)
    e2i = np.array([rid.index(a) + ns for a in anchors], dtype=np.int32)
    e1j = np.array([k for k, i in enumerate(sid) if i not in anchors], dtype=np.int32)
    e2j = np.array([k + ns for k, i in enumerate(rid) if i not in anchors], dtype=np.int32)
    dd = {'tot_obj_pts': torch.from_numpy(pts).float(), 'batch_size': 1,
          'tot_bow_vec_object_attr_feats': torch.zeros(ns + nr, 164, dtype=torch.float64),
          'tot_bow_vec_object_edge_feats': torch.zeros(ns + nr, 41, dtype=torch.float64),
          'tot_rel_pose': torch.zeros(ns + nr, 3, dtype=torch.float64),
          'e1i': e1i, 'e2i': e2i, 'e1j': e1j, 'e2j': e2j}
    torch.manual_seed(42)
    model = sg_aligner.MultiModalEncoder(modules=['point'], rel_dim=41, attr_dim=164)
    model.train()
    out = model(dd)
    ol = losses.OverallLoss(losses.CustomMultiLossLayer(1), losses.CustomMultiLossLayer(1), 'cpu',
                            {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': ['point']})
    res = ol(out, dd)
    res['loss'].backward()
    emb = out['point'].detach()
    e = emb / emb.norm(dim=1)[:, None]                               # inference_align_reg.py:126-128
    sim = 1 - e @ e.t()
    rank_list = torch.argsort(sim, dim=1)
    mrr = alignment.compute_mean_reciprocal_rank(rank_list, e1i, e2i, [])
    hits = [alignment.compute_hits_k(rank_list, e1i, e2i, k)[0] for k in (1, 2, 3, 4, 5)]
    sgar = alignment.compute_sgar(sim, rank_list, e1i, e2i, ['2', '50', '100'])
    corrs = alignment.compute_node_corrs(rank_list, ns, 2)
    score = alignment.compute_alignment_score(rank_list, ns, nr)
    sd = model.state_dict()
    arrs = {'sd__' + k: v for k, v in sd.items()}
    arrs.update({'grad__' + k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    npz('example_pair_point', pts=dd['tot_obj_pts'], e1i=e1i, e2i=e2i, e1j=e1j, e2j=e2j,
        counts=np.array([ns, nr]), emb=emb, loss=res['loss'], mrr=np.array(mrr), hits=np.array(hits),
        sgar=np.array([sgar['2'], sgar['50'], sgar['100']]), sim=sim, node_corrs=np.array(corrs),
        align_score=np.array(score), **arrs)


def gen_alignment(alignment):
    """utils/alignment.py on hand-made distance matrices, incl. a row whose self entry is NOT rank 0."""
    rng = np.random.default_rng(11)
    n, ns = 9, 5
    sim = rng.random((n, n))
    sim = (sim + sim.T) / 2
    np.fill_diagonal(sim, 0.0)
    sim[2, 2] = 0.7                      # self not at rank 0 for row 2
    sim_t = torch.from_numpy(sim)
    rank_list = torch.argsort(sim_t, dim=1)
    e1i = np.array([0, 2, 3], dtype=np.int32)
    e2i = np.array([6, 5, 8], dtype=np.int32)
    mrr = alignment.compute_mean_reciprocal_rank(rank_list, e1i, e2i, [])
    hits = [alignment.compute_hits_k(rank_list, e1i, e2i, k)[0] for k in (1, 2, 3, 4, 5)]
    sgar = alignment.compute_sgar(sim_t, rank_list, e1i, e2i, ['2', '50', '100'])
    corrs = alignment.compute_node_corrs(rank_list, ns, 3)
    score = alignment.compute_alignment_score(rank_list, ns, n - ns)
    npz('alignment_handmade', sim=sim, e1i=e1i, e2i=e2i, mrr=np.array(mrr), hits=np.array(hits),
        sgar=np.array([sgar['2'], sgar['50'], sgar['100']]), node_corrs=np.array(corrs),
        align_score=np.array(score), src_count=np.array(ns))


def gen_full_multimodal(losses, sg_aligner):
    """(7) full P+S+R+A encoder + OverallLoss through the reference's own orchestration, with the
    GAT layer supplied by the stub -> GAT-UNPINNED.  Pins sg_aligner.py:71-137 dispatch / ordering,
    the state_dict key set, and the loss wiring end to end."""
    sys.path.insert(0, ROOT)
    from sgaligner_amd.synthetic import make_batch
    dd = make_batch(n_pairs=2, n_obj=(7, 6), n_pts=48, seed=3, device='cpu', ragged=True)
    mods = ['point', 'gat', 'rel', 'attr']
    torch.manual_seed(42)
    model = sg_aligner.MultiModalEncoder(modules=mods, rel_dim=41, attr_dim=164)
    with torch.no_grad():
        for c in (model.object_encoder.conv1, model.object_encoder.conv2, model.object_encoder.conv3):
            c.bias.normal_(0, 0.05)
        for l in model.structure_encoder.layer_stack:
            l.bias.normal_(0, 0.05)
        model.fusion.weight.normal_(1.0, 0.3)
    model.train()
    out = model(dd)
    ial_layer, icl_layer = losses.CustomMultiLossLayer(4), losses.CustomMultiLossLayer(4)
    ol = losses.OverallLoss(ial_layer, icl_layer, 'cpu', {'zoom': 0.1, 'wt_align_loss': 1.0,
                                                         'wt_contrastive_loss': 1.0, 'modules': mods})
    res = ol(out, dd)
    res['loss'].backward()
    arrs = {'sd__' + k: v for k, v in model.state_dict().items()}
    arrs.update({'grad__' + k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    arrs.update({'out__' + k: v for k, v in out.items()})
    for k in ('tot_obj_pts', 'tot_bow_vec_object_attr_feats', 'tot_bow_vec_object_edge_feats', 'tot_rel_pose',
              'edges', 'e1i', 'e2i', 'e1j', 'e2j', 'graph_per_obj_count', 'graph_per_edge_count',
              'tot_obj_count', 'e1i_count'):
        arrs['dd__' + k] = dd[k]
    npz('full_multimodal_gat_unpinned', loss=res['loss'], icl_uni=res['icl_loss_unimodal'],
        icl_multi=res['icl_loss_multimodal'], ial=res['ial_loss'], g_lv_ial=ial_layer.log_vars.grad,
        g_lv_icl=icl_layer.log_vars.grad, sd_keys=np.array(list(model.state_dict().keys())), **arrs)


def gen_fps():
    """Per-object farthest-point sampling: utils/point_cloud.py:61-89 run as is, with np.random.randint
    patched to hand back a recorded start index.  cv2 / open3d.ml.torch / trimesh (imported at the top of that file,
    unused by this function) are stubbed."""
    import importlib.util
    for name in ('cv2', 'open3d', 'open3d.ml', 'open3d.ml.torch', 'trimesh'):
        sys.modules.setdefault(name, types.ModuleType(name))
    spec = importlib.util.spec_from_file_location('ref_point_cloud', os.path.join(REF, 'utils', 'point_cloud.py'))
    ref_pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_pc)
    from oracle import fps_oracle
    rng = np.random.default_rng(7)
    cases = {}
    # (N, npoint): tiny, ragged, duplicates (exact ties -> first arg-max), one > 2048 and one > 8192 points
    specs = [(5, 5), (37, 16), (300, 64), (1000, 256), (2500, 128), (9000, 64)]
    real = np.load(os.path.join(REF, 'example_data', 'scene_1', 'data.npy'))
    for k, (n, m) in enumerate(specs):
        pts = (rng.standard_normal((n, 3)) * np.array([2.0, 1.0, 0.3])).astype(np.float32)
        if k == 2:
            pts[100:200] = pts[0:100]                     # exact duplicates
        if k == 3:
            pts = np.round(pts * 4) / 4                    # a coarse lattice: many equal distances
        start = int(rng.integers(0, n))
        orig = np.random.randint
        np.random.randint = lambda lo, hi=None, *a, **kw: start
        try:
            sampled, idx = ref_pc.pcl_farthest_sample(pts, m, return_idxs=True)
        finally:
            np.random.randint = orig
        assert np.array_equal(idx, fps_oracle.farthest_point_sample_idx(pts, m, start)), 'oracle != reference'
        assert np.array_equal(sampled, pts[idx])
        cases[f'pts{k}'] = pts; cases[f'start{k}'] = np.int32(start); cases[f'idx{k}'] = idx.astype(np.int32)
    # a real object of the example scan (f4 vertices as the preprocessing sees them)
    obj = real[real['objectId'] == np.bincount(real['objectId'].astype(np.int64)).argmax()]
    pts = np.stack([obj['x'], obj['y'], obj['z']], 1).astype(np.float32)
    start = 3
    orig = np.random.randint
    np.random.randint = lambda lo, hi=None, *a, **kw: start
    try:
        _, idx = ref_pc.pcl_farthest_sample(pts, 256, return_idxs=True)
    finally:
        np.random.randint = orig
    assert np.array_equal(idx, fps_oracle.farthest_point_sample_idx(pts, 256, start))
    k = len(specs)
    cases[f'pts{k}'] = pts; cases[f'start{k}'] = np.int32(start); cases[f'idx{k}'] = idx.astype(np.int32)
    npz('fps_cases', n_cases=np.int32(k + 1), **cases)


def gen_hull():
    """Per-object convex-hull barycentre (preprocessing/scan3r/preprocess.py:93-96).  The four lines live
    inside process_scan and cannot be imported; they are scipy.spatial.ConvexHull + three np.mean calls, restated in
    oracle/hull_oracle.py and evaluated here on the objects of the reference's own example scan (as preprocess.py selects
    them: objects with >= min_obj_points = 50 points, :90) plus synthetic shapes with coplanar / collinear structure."""
    from oracle import hull_oracle
    real = np.load(os.path.join(REF, 'example_data', 'scene_1', 'data.npy'))
    ids, cnt = np.unique(real['objectId'], return_counts=True)
    ids = ids[cnt >= 50]
    cases = {}
    k = 0
    for oid in ids[:6]:
        obj = real[real['objectId'] == oid]
        pts = np.stack([obj['x'], obj['y'], obj['z']], 1)
        bc, verts = hull_oracle.hull_barycenter(pts)
        cases[f'pts{k}'] = pts; cases[f'bc{k}'] = bc; cases[f'verts{k}'] = verts.astype(np.int32)
        k += 1
    rng = np.random.default_rng(11)
    cube = rng.integers(0, 4, size=(400, 3)).astype(np.float64)                    # lattice cube: coplanar and collinear points
    sphere = rng.standard_normal((3000, 3)); sphere /= np.linalg.norm(sphere, axis=1, keepdims=True)
    blob = rng.standard_normal((4000, 3)) * np.array([2.0, 1.0, 0.3])
    for pts in (cube, sphere, blob, blob[:5]):
        bc, verts = hull_oracle.hull_barycenter(pts)
        cases[f'pts{k}'] = pts; cases[f'bc{k}'] = bc; cases[f'verts{k}'] = verts.astype(np.int32)
        k += 1
    npz('hull_cases', n_cases=np.int32(k), **cases)


def gen_dataset():
    """Dataset/collate: the reference's Scan3RDataset (src/datasets/scan3r.py) run on a synthetic
    on-disk dataset written by sgaligner_amd.datasets.synthetic_scan3r (deterministic in its seed; the test re-writes the
    same files).  `plyfile` (imported by utils/scan3r.py, unused here) is stubbed."""
    import tempfile
    sys.modules.setdefault('plyfile', types.ModuleType('plyfile'))
    sys.modules['plyfile'].PlyData = object
    sys.path.insert(0, os.path.join(REF, 'src'))
    sys.path.insert(0, REF)
    import datasets.scan3r as ref_ds
    from sgaligner_amd.datasets import synthetic_scan3r as S
    root = tempfile.mkdtemp(prefix='sga_scan3r_')
    S.write_dataset(root, n_pairs=6, seed=5)
    for split, kw in (('val', {}), ('train', {}), ('val', {'overlap_low': 0.3, 'overlap_high': 0.8})):
        cfg = S.make_cfg(root, pc_res=64, **kw)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            ds = ref_ds.Scan3RDataset(cfg, split)
        np.random.seed(123)
        batch = ds.collate_fn([ds[i] for i in range(len(ds))])
        arrs = {}
        for k, v in batch.items():
            if k == 'batch_size':
                arrs[k] = np.int64(v)
            elif k == 'scene_ids':
                arrs[k] = np.asarray(v).astype('U16')
            else:
                arrs[k] = v
        tag = split + ('_overlap' if kw else '')
        npz(f'scan3r_collate_{tag}', n_items=np.int64(len(ds)), **arrs)


def gen_pct():
    """NaivePCT, eval mode: the reference module itself (src/aligner/networks/pct.py) with
    randomised BatchNorm statistics/affines; checks oracle/pct_oracle.py against it and stores inputs, state_dict, outputs."""
    import_reference()
    import aligner.networks.pct as ref_pct
    from oracle import pct_oracle
    torch.manual_seed(0)
    m = ref_pct.NaivePCT()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
    m.eval()
    arrs = {'sd__' + k: v for k, v in m.state_dict().items()}      # one set of weights (5 MB), three inputs
    for tag, (T, N) in {'small': (3, 96), 'p512': (2, 512), 'ragged': (5, 45)}.items():
        x = torch.randn(T, 3, N) * torch.tensor([1.0, 0.6, 0.3]).reshape(1, 3, 1)
        with torch.no_grad():
            y = m(x)
            yo = pct_oracle.naive_pct_forward(x, m.state_dict())
        assert (y - yo).abs().max() < 1e-5 * max(1.0, y.abs().max().item()), (tag, (y - yo).abs().max())
        arrs['x_' + tag] = x
        arrs['y_' + tag] = y
    npz('pct_eval', **arrs)

    # train mode (batch statistics, running-stat updates), Dropout p = 0, on the SAME weights: the reference module's
    # output, a subset of its parameter gradients (the full set is 5 MB) and the updated running statistics; the
    # train-mode oracle is checked against ALL gradients here.
    m.train()
    m.dp1.p = 0.0
    m.dp2.p = 0.0
    torch.manual_seed(3)
    T, N = 6, 80
    x = torch.randn(T, 3, N) * torch.tensor([1.0, 0.6, 0.3]).reshape(1, 3, 1)
    cot = torch.randn(T, 256)
    sd_before = {k: v.clone() for k, v in m.state_dict().items()}
    m.zero_grad()
    y = m(x)
    (y * cot).sum().backward()
    sd_o = {k: v.clone() for k, v in sd_before.items()}
    leaves = {}
    for name, p_ in m.named_parameters():
        leaves[name] = sd_o[name].clone().requires_grad_(True)
        sd_o[name] = leaves[name]
    for sa in ('sa1', 'sa2', 'sa3', 'sa4'):                     # q_conv.weight IS k_conv.weight (pct.py:199)
        sd_o[sa + '.k_conv.weight'] = sd_o[sa + '.q_conv.weight']      # named_parameters() lists the tied tensor as q_conv.weight
    yo, ns = pct_oracle.naive_pct_forward_train(x, sd_o)
    (yo * cot).sum().backward()
    assert (y - yo).abs().max() < 1e-4 * max(1.0, y.abs().max().item()), (y - yo).abs().max()
    gmax = max(p_.grad.abs().max().item() for p_ in m.parameters())
    for name, p_ in m.named_parameters():
        g_ref, g_o = p_.grad, leaves[name].grad
        # (a bias in front of a train-mode BatchNorm has an exactly-zero gradient: both sides hold rounding noise there)
        assert (g_ref - g_o).abs().max() < 2e-4 * max(g_ref.abs().max().item(), 1e-2 * gmax), (name, (g_ref - g_o).abs().max(), g_ref.abs().max())
    sd_after = m.state_dict()
    for k, v in ns.items():
        assert (sd_after[k] - v).abs().max() < 1e-5 * max(1.0, v.abs().max().item()), k
    tr = {'x': x, 'cot': cot, 'y': y}
    keep = ['embedding.conv1.weight', 'embedding.conv2.weight', 'embedding.bn1.weight', 'embedding.bn2.bias', 'sa1.q_conv.weight',
            'sa1.v_conv.weight', 'sa1.v_conv.bias', 'sa3.trans_conv.weight', 'sa4.after_norm.weight', 'sa4.after_norm.bias',
            'linear.1.weight', 'linear2.weight', 'linear2.bias', 'bn1.weight', 'bn2.bias']
    named = dict(m.named_parameters())
    for k in keep:
        tr['g__' + k] = named[k].grad
    for k, v in sd_after.items():
        if 'running_' in k or 'num_batches' in k:
            tr['after__' + k] = v
    npz('pct_train', **tr)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'pct':
        gen_pct()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'hull':
        gen_hull()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'fps':
        gen_fps()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'dataset':
        gen_dataset()
        return
    losses, pointnet, sg_aligner, alignment = import_reference()
    gen_fps()
    gen_hull()
    gen_dataset()
    gen_pct()
    gen_pointnet(pointnet)
    gen_fusion(sg_aligner)
    gen_losses(losses)
    gen_example_pair(losses, sg_aligner, alignment)
    gen_alignment(alignment)
    gen_full_multimodal(losses, sg_aligner)


if __name__ == '__main__':
    main()
