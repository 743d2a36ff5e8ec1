"""Alignment metrics -- the reference's utils/alignment.py API (functions over a rank list, kept for
callers that already hold one) plus the fused device path `evaluate_batch` that the testers use.

Reference: utils/alignment.py:3-11 (MRR), :13-25 (Hits@k), :27-57 (SGAR), :59-70 (node correspondences),
:79-89 (alignment score); eval_step arithmetic src/inference/sgaligner/inference_align_reg.py:98-143."""
import numpy as np
import torch

from .. import ops


# ---- rank-list API (host side, same signatures as the reference) -------------------------------------
def _np_rank(rank_list):
    return rank_list.detach().cpu().numpy() if isinstance(rank_list, torch.Tensor) else np.asarray(rank_list)


def _others(row, self_idx):
    row = np.asarray(row)
    hit = np.nonzero(row == self_idx)[0]
    return np.delete(row, hit[0]) if hit.size else row          # list.remove(): first occurrence by value


def compute_mean_reciprocal_rank(rank_list, e1i_idxs, e2i_idxs, mrr_arr):
    rl = _np_rank(rank_list)
    for a, b in zip(e1i_idxs, e2i_idxs):
        mrr_arr.append(1.0 / (int(np.nonzero(_others(rl[a], a) == b)[0][0]) + 1))
    return mrr_arr


def compute_hits_k(rank_list, e1i_idxs, e2i_idxs, k=1):
    rl = _np_rank(rank_list)
    correct = sum(int(b in _others(rl[a], a)[:k]) for a, b in zip(e1i_idxs, e2i_idxs))
    return correct, e1i_idxs.shape[0]


def _sgar_from(pred, sim_top1, gt, modes):
    order = np.argsort(np.asarray(sim_top1), kind='stable')
    out = {}
    for mode in modes:
        sel = order[:2] if mode == '2' else (order[:len(order) // 2] if mode == '50' else order)
        out[mode] = 0.0 if any(pred[i] != gt[i] for i in sel) else 1.0
    return out


def compute_sgar(sim, rank_list, e1i_idxs, e2i_idxs, modes):
    rl = _np_rank(rank_list)
    s = sim.detach().cpu().numpy() if isinstance(sim, torch.Tensor) else np.asarray(sim)
    pred = [int(_others(rl[a], a)[0]) for a in e1i_idxs]
    return _sgar_from(pred, [s[a][p] for a, p in zip(e1i_idxs, pred)], [int(b) for b in e2i_idxs], modes)


def compute_node_corrs(rank_list, src_objects_count, k=1):
    rl = _np_rank(rank_list)
    return [(i, int(j)) for i in range(src_objects_count) for j in _others(rl[i], i)[:k] if j >= src_objects_count]


def get_node_corrs_objects_ids(node_corrs, objects_ids, batch_offset):
    return [(objects_ids[a + batch_offset], objects_ids[b + batch_offset]) for a, b in node_corrs]


def compute_alignment_score(rank_list, src_objects_count, ref_objects_count):
    rl = _np_rank(rank_list)
    aligned = sum(int(_others(rl[i], i)[0] >= src_objects_count) for i in range(src_objects_count))
    return aligned / ref_objects_count


# ---- fused device path ------------------------------------------------------------------------------
def evaluate_batch(embedding, data_dict, all_k=(1, 2, 3, 4, 5), recall_modes=('2', '50', '100'), reg_k=0):
    """The alignment block of AlignerRegTester.eval_step for a whole batch: the per-pair E E^T blocks on the matrix cores
    with rank / top-K in the epilogue (sga_simrank), Hits@K counts and SGAR per pair on the device (sga_pair_metrics), ONE
    device->host copy of the small result arrays, and vectorised formatting -- no per-pair host loop.
    Returns the reference's meter dict: {'mrr': [...], k: {'correct','total'}, 'sgar': {mode: [...]}}
    (+ 'node_corrs': per-pair list of (src, ref) pair-local index tuples when reg_k > 0)."""
    counts = np.asarray(data_dict['tot_obj_count']).reshape(-1)
    e1c = np.asarray(data_dict['e1i_count']).reshape(-1)
    e1i = np.asarray(data_dict['e1i'])
    e2i = np.asarray(data_dict['e2i'])
    res = {'mrr': [], 'sgar': {m: [] for m in recall_modes}}
    for k in all_k:
        res[k] = {'correct': 0, 'total': 0}
    if len(e1i):
        rank, tki, tks, lay = ops.simrank(embedding, counts, e1i, e2i, 1)
        pm = ops.pair_metrics(rank, tki, tks, e2i, lay, e1c)
        host = torch.cat([rank.float(), pm.reshape(-1)]).cpu().numpy()          # one read-back
        rk, pm = host[:len(e1i)], host[len(e1i):].reshape(-1, 12)
        res['mrr'] = (1.0 / rk).tolist()
        has = e1c > 0
        for k in all_k:
            res[k]['correct'] = int(pm[:, k - 1].sum()) if 1 <= k <= 5 else int((rk <= k).sum())
            res[k]['total'] = int(e1c.sum())
        col = {'2': 7, '50': 8, '100': 9}
        for m in recall_modes:
            res['sgar'][m] = pm[has, col[m]].astype(float).tolist()
    if reg_k > 0:
        gpc = np.asarray(data_dict['graph_per_obj_count'])
        offs = np.concatenate([[0], np.cumsum(counts)])
        ns_all = gpc[:, 0].astype(np.int64)
        # src objects of every pair: offs[b] + arange(ns_b), vectorised
        rep = np.repeat(np.arange(len(counts)), ns_all)
        loc = np.arange(int(ns_all.sum())) - np.repeat(np.concatenate([[0], np.cumsum(ns_all)])[:-1], ns_all)
        qi = (offs[rep] + loc).astype(np.int32)
        _, tk, _, _ = ops.simrank(embedding, counts, qi, None, reg_k)
        tk = tk.cpu().numpy()
        keep = tk >= ns_all[rep][:, None]                                        # top-k non-self entries in the ref half (:68)
        res['node_corrs'] = []
        p = 0
        for b in range(len(counts)):
            ns = int(ns_all[b])
            rows, cols = np.nonzero(keep[p:p + ns])
            res['node_corrs'].append(list(zip(rows.tolist(), tk[p:p + ns][rows, cols].tolist())))
            p += ns
    return res
