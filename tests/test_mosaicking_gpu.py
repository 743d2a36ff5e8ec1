"""GPU: the hand-built `data_dict` of the reference's mosaicking tester (src/inference/sgaligner/inference_mosaicking.py:20-66) --
ONE sub-scan pair, `batch_size` 1, 2-D [[n_src, n_ref]] count arrays, BoW features and relative poses left in the pickle's float64,
no e1i/e2i/e1j/e2j sets, extra keys (`src_objects_idxs`, `center`, ...) -- goes through `test_step` and the pairwise-alignment
arithmetic of `run_pairwise_alignment` (:129-149: joint embedding -> row-normalise -> 1 - E E^T -> argsort -> node correspondences,
their object ids, alignment score) and gives what the oracle gives on the same dict and weights."""
import os.path as osp
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mosaicking_dict(root, src_id, ref_id, pc_res):
    """The dict `load_subscan_pair` assembles (inference_mosaicking.py:20-66), from the same files."""
    verts = np.load(osp.join(root, 'scans', src_id, 'data.npy'))
    center = np.mean(np.stack([verts['x'], verts['y'], verts['z']]).transpose((1, 0)), axis=0)
    with open(osp.join(root, 'files', 'orig', 'data', src_id + '.pkl'), 'rb') as fh:
        src = pickle.load(fh)
    with open(osp.join(root, 'files', 'orig', 'data', ref_id + '.pkl'), 'rb') as fh:
        ref = pickle.load(fh)
    cat = lambda k: torch.cat([torch.from_numpy(src[k]), torch.from_numpy(ref[k])])
    sp, rp = src['obj_points'][pc_res] - center, ref['obj_points'][pc_res] - center
    src_idxs = np.array([src['object_id2idx'][o] for o in src['objects_id']])
    return {
        'obj_ids': np.concatenate([src['objects_id'], ref['objects_id']]),
        'tot_obj_pts': torch.cat([torch.from_numpy(sp), torch.from_numpy(rp)]).type(torch.FloatTensor),
        'src_objects_idxs': src_idxs, 'src_objects_counts': src_idxs.shape[0],
        'tot_obj_count': sp.shape[0] + rp.shape[0],
        'graph_per_obj_count': np.array([[sp.shape[0], rp.shape[0]]]),
        'graph_per_edge_count': np.array([[src['edges'].shape[0], ref['edges'].shape[0]]]),
        'tot_bow_vec_object_attr_feats': cat('bow_vec_object_attr_feats'),         # float64, never cast by the tester
        'tot_bow_vec_object_edge_feats': cat('bow_vec_object_edge_feats'),
        'tot_rel_pose': cat('rel_trans'),
        'edges': cat('edges'),
        'global_obj_ids': np.concatenate((src['objects_cat'], ref['objects_cat'])),
        'scene_ids': [src_id, ref_id], 'center': center, 'batch_size': 1,
    }


@pytest.mark.parametrize('modules', [['point', 'gat', 'rel', 'attr'], ['point', 'gat'], ['point']])
def test_mosaicking_data_dict_through_test_step(tmp_path, modules):
    from oracle import sga_oracle as O
    from sgaligner_amd.datasets import synthetic_scan3r as S
    from sgaligner_amd.synthetic import to_device
    from sgaligner_amd.trainer import AlignerSteps
    from sgaligner_amd.utils import alignment
    root = str(tmp_path)
    pairs = S.write_dataset(root, n_pairs=3, seed=11)
    steps = AlignerSteps(modules, device='cuda:0', seed=7)
    params = {k: v.detach().cpu().clone() for k, v in steps.model.state_dict().items() if 'num_batches' not in k}
    key = 'joint' if len(modules) > 1 else modules[0]
    for sid, rid in pairs:
        dd = _mosaicking_dict(root, sid, rid, 64)
        assert dd['tot_bow_vec_object_edge_feats'].dtype == torch.float64 and 'e1i' not in dd
        with torch.no_grad():
            ref = O.encoder_forward(params, dd, modules)[key]
        out = steps.test_step(0, to_device(dd, 'cuda:0'))[key]
        assert not out.requires_grad
        assert (out.cpu() - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))
        # run_pairwise_alignment's arithmetic on both sides (inference_mosaicking.py:137-147)
        n_src, n_ref = int(dd['graph_per_obj_count'][0][0]), int(dd['graph_per_obj_count'][0][1])
        res = []
        for e in (out.cpu(), ref):
            e = e / e.norm(dim=1)[:, None]
            rank_list = torch.argsort(1 - torch.mm(e, e.transpose(0, 1)), dim=1)
            corrs = alignment.compute_node_corrs(rank_list, n_src, k=1)
            res.append((corrs, alignment.get_node_corrs_objects_ids(corrs, dd['obj_ids'], 0),
                        alignment.compute_alignment_score(rank_list, n_src, n_ref)))
        assert res[0] == res[1]
        assert res[0][0] == O.node_corrs(1 - torch.mm(ref / ref.norm(dim=1)[:, None], (ref / ref.norm(dim=1)[:, None]).t()), n_src, k=1)
