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
    """The alignment block of AlignerRegTester.eval_step for a whole batch in two kernel launches.
    Returns the reference's meter dict: {'mrr': [...], k: {'correct','total'}, 'sgar': {mode: [...]}}
    (+ 'node_corrs': per-pair list of (src, ref) pair-local index tuples when reg_k > 0)."""
    counts = np.asarray(data_dict['tot_obj_count']).reshape(-1)
    e1c = np.asarray(data_dict['e1i_count']).reshape(-1)
    q_pair = np.repeat(np.arange(len(counts)), e1c)
    e1i = np.asarray(data_dict['e1i'])
    e2i = np.asarray(data_dict['e2i'])
    rank, tki, tks = ops.simrank(embedding, counts, q_pair, e1i, e2i, 1)
    rank = rank.cpu().numpy()
    top1 = tki.cpu().numpy()[:, 0] if len(rank) else np.zeros(0, dtype=np.int64)
    top1s = tks.cpu().numpy()[:, 0] if len(rank) else np.zeros(0)
    offs = np.concatenate([[0], np.cumsum(counts)])
    res = {'mrr': [], 'sgar': {m: [] for m in recall_modes}}
    for k in all_k:
        res[k] = {'correct': 0, 'total': 0}
    a0 = 0
    for b, na in enumerate(e1c):
        if na:
            r = rank[a0:a0 + na]
            res['mrr'] += [1.0 / int(x) for x in r]
            for k in all_k:
                res[k]['correct'] += int((r <= k).sum())
                res[k]['total'] += int(na)
            gt = (e2i[a0:a0 + na] - offs[b]).tolist()
            sg = _sgar_from(top1[a0:a0 + na].tolist(), top1s[a0:a0 + na].tolist(), gt, recall_modes)
            for m in recall_modes:
                res['sgar'][m].append(sg[m])
        a0 += na
    if reg_k > 0:
        gpc = np.asarray(data_dict['graph_per_obj_count'])
        qp = np.repeat(np.arange(len(counts)), gpc[:, 0])
        qi = np.concatenate([np.arange(offs[b], offs[b] + gpc[b, 0]) for b in range(len(counts))]) if len(counts) else np.zeros(0)
        _, tk, _ = ops.simrank(embedding, counts, qp, qi, None, reg_k)
        tk = tk.cpu().numpy()
        res['node_corrs'] = []
        p = 0
        for b in range(len(counts)):
            ns = int(gpc[b, 0])
            res['node_corrs'].append([(i, int(j)) for i in range(ns) for j in tk[p + i] if j >= ns])
            p += ns
    return res
