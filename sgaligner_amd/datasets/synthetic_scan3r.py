"""Write a small synthetic dataset in the reference's on-disk Scan3R layout (see datasets/scan3r.py's header), so the
dataset/collate path can be exercised -- and pinned against the reference's own Scan3RDataset -- without 3RScan.

Schema sources: preprocessing/scan3r/preprocess.py:195-211 (pkl keys), :321,357 (bag-of-words features),
utils/scan3r.py:143-144 (data.npy vertex dtype), the anchors json as read by src/datasets/scan3r.py:30-41,60-64,84.
Deterministic in `seed`."""
from __future__ import annotations

import json
import os
import os.path as osp
import pickle

import numpy as np

VERTEX_DTYPE = [('x', 'f4'), ('y', 'f4'), ('z', 'f4'), ('red', 'u1'), ('green', 'u1'), ('blue', 'u1'),
                ('objectId', 'h'), ('globalId', 'h'), ('NYU40', 'u1'), ('Eigen13', 'u1'), ('RIO27', 'u1')]


def _scan(rng, scan_id, object_ids, resolutions, rel_dim, attr_dim, shared):
    n = len(object_ids)
    centers = rng.uniform(-3, 3, (n, 3))
    scales = rng.uniform(0.1, 0.5, (n, 3))
    for k, oid in enumerate(object_ids):                  # objects shared between two scans have the same shape
        if oid in shared:
            centers[k], scales[k] = shared[oid]
        else:
            shared[oid] = (centers[k].copy(), scales[k].copy())
    obj_points = {}
    for res in resolutions:
        obj_points[res] = centers[:, None, :] + scales[:, None, :] * rng.standard_normal((n, res, 3))
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
    m = ii != jj
    edges = np.stack([ii[m], jj[m]], 1).astype(np.int64)
    rel = np.zeros((n, rel_dim))
    for k in range(n):
        hits = rng.integers(1, rel_dim, 3)
        np.add.at(rel[k], hits, 1.0)
        rel[k, 0] = max(0.0, (n - 1) - rel[k, 1:].sum())
    attr = (rng.random((n, attr_dim)) < 0.03).astype(np.float64)
    pkl = {
        'scan_id': scan_id, 'objects_id': np.array(object_ids), 'global_objects_id': np.array(object_ids) + 100,
        'objects_cat': np.array(object_ids) + 100, 'edges': edges, 'obj_points': obj_points, 'objects_count': n,
        'edges_count': len(edges), 'object_id2idx': {int(o): k for k, o in enumerate(object_ids)},
        'rel_trans': centers[0][None, :] - centers, 'root_obj_id': int(object_ids[0]),
        'bow_vec_object_edge_feats': rel, 'bow_vec_object_attr_feats': attr,
    }
    nv = 40 * n
    verts = np.zeros(nv, dtype=VERTEX_DTYPE)
    owner = rng.integers(0, n, nv)
    p = centers[owner] + scales[owner] * rng.standard_normal((nv, 3))
    verts['x'], verts['y'], verts['z'] = p[:, 0].astype('f4'), p[:, 1].astype('f4'), p[:, 2].astype('f4')
    verts['objectId'] = np.array(object_ids)[owner].astype('h')
    return pkl, verts


def write_dataset(root, n_pairs=6, seed=0, resolutions=(32, 64), rel_dim=41, attr_dim=164, modes=('orig',),
                  splits=('train', 'val'), anchor_type_name=''):
    """Returns the list of (src, ref) scan ids.  Each pair shares about half of its objects; anchor lists contain a few
    ids that are filtered out by the loader (0, ids missing on one side) to exercise scan3r.py:88-89."""
    rng = np.random.default_rng(seed)
    pairs, anchors = [], []
    for k in range(n_pairs):
        n_src, n_ref = int(rng.integers(5, 12)), int(rng.integers(5, 12))
        pool = rng.permutation(np.arange(1, 40))
        n_common = min(n_src, n_ref) // 2 + 1
        common = pool[:n_common]
        src_ids = rng.permutation(np.concatenate([common, pool[n_common:n_common + n_src - n_common]]))
        ref_ids = rng.permutation(np.concatenate([common, pool[20:20 + n_ref - n_common]]))
        shared = {}
        sid, rid = f'scan{k:03d}_0', f'scan{k:03d}_1'
        for scan_id, ids in ((sid, src_ids), (rid, ref_ids)):
            pkl, verts = _scan(rng, scan_id, [int(i) for i in ids], resolutions, rel_dim, attr_dim, shared)
            os.makedirs(osp.join(root, 'scans', scan_id), exist_ok=True)
            np.save(osp.join(root, 'scans', scan_id, 'data.npy'), verts)
            for mode in modes:
                os.makedirs(osp.join(root, 'files', mode, 'data'), exist_ok=True)
                with open(osp.join(root, 'files', mode, 'data', scan_id + '.pkl'), 'wb') as fh:
                    pickle.dump(pkl, fh, protocol=pickle.HIGHEST_PROTOCOL)
        listed = [int(i) for i in rng.permutation(common)] + [0, int(src_ids[-1]) if src_ids[-1] not in common else 0]
        anchors.append({'src': sid, 'ref': rid, 'overlap': float(np.round(rng.uniform(0.1, 0.9), 3)), 'anchorIds': listed})
        pairs.append((sid, rid))
    for mode in modes:
        for split in splits:
            with open(osp.join(root, 'files', mode, f'anchors{anchor_type_name}_{split}.json'), 'w') as fh:
                json.dump(anchors, fh, indent=4)
    return pairs


def make_cfg(root, pc_res=64, scan_type='subscan', data_mode='orig', overlap_low=0.0, overlap_high=0.0,
             modules=('point', 'gat', 'rel', 'attr'), batch_size=4, max_epoch=2, lr=1e-3, output_dir=None):
    """The slice of the reference's yacs config the dataset and the trainer read (configs/scan3r/*.yaml), as plain
    namespaces."""
    from types import SimpleNamespace as NS
    return NS(model_name='sgaligner', scan_type=scan_type, modules=list(modules), seed=42, num_workers=0,
              output_dir=output_dir or osp.join(root, 'output'),
              data=NS(root_dir=root, subscan_dir=root),
              preprocess=NS(anchor_type_name=''),
              model=NS(rel_dim=41, attr_dim=164),
              loss=NS(zoom=0.1, alignment_loss_weight=1.0, constrastive_loss_weight=1.0),
              optim=NS(lr=lr, weight_decay=0.0, max_epoch=max_epoch, grad_acc_steps=1),
              train=NS(pc_res=pc_res, use_augmentation=False, rot_factor=1.0, augmentation_noise=0.005, batch_size=batch_size),
              val=NS(pc_res=pc_res, data_mode=data_mode, overlap_low=overlap_low, overlap_high=overlap_high, batch_size=batch_size))
