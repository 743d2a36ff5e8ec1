"""CPU ORACLE for the SGAligner embedding + matching hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain torch ops on CPU tensors, fp32 or fp64) of the
reference algorithm named by BASELINE.json:north_star.  It is the checker, never the product:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Nothing
under sgaligner_amd/ imports it, and the product path raises if the HIP library is missing.

Pinning status
  * PointNet / fusion / Q / ICL / IAL / multi-loss / overall loss / eval similarity / metrics:
    PINNED -- checked against outputs of the reference itself, imported in the build container by
    oracle/make_golden.py; vectors are committed under tests/golden/ (tests/test_oracle_golden.py).
  * GATConv arithmetic (torch-geometric==2.2.0, req.yml:259, NOT vendored in /root/reference):
    PARITY UNPINNED.  `gat_conv` restates the published PyG 2.2.0 GATConv algorithm
    (defaults concat=True, negative_slope=0.2, add_self_loops=True, bias=True, shared lin_src/lin_dst)
    anchored on the reference call sites src/aligner/networks/gat.py:36-37,44 and
    src/aligner/sg_aligner.py:86-110.  It is cross-checked against an independent dense-mask
    formulation (tests/test_oracle_golden.py::test_gat_edge_vs_dense).

Every function cites the reference file:line (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------------
def init_params(modules: Sequence[str], rel_dim: int = 41, attr_dim: int = 164, emb_dim: int = 100,
                pt_out_dim: int = 256, hidden_units=(3, 128, 128), heads=(2, 2), seed: int = 42,
                dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Flat parameter dict with the reference's state_dict key names (sg_aligner.py:38-69).

    Init distributions follow the reference (conv: xavier_normal gain 1 / bias 0, base.py:24-43;
    Linear: torch default; GAT: glorot, zeros bias) but the draw order is the oracle's own -- tests
    always share explicit tensors between oracle and product, never a seed.
    """
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}

    def xavier_normal(shape, fan_in, fan_out):
        std = math.sqrt(2.0 / (fan_in + fan_out))
        return torch.randn(shape, generator=g, dtype=torch.float64) * std

    def linear(prefix, out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        p[prefix + '.weight'] = (torch.rand((out_f, in_f), generator=g, dtype=torch.float64) * 2 - 1) * bound
        p[prefix + '.bias'] = (torch.rand((out_f,), generator=g, dtype=torch.float64) * 2 - 1) * bound

    def glorot(shape, fan_in, fan_out):
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * a

    linear('meta_embedding_rel', emb_dim, rel_dim)
    linear('meta_embedding_attr', emb_dim, attr_dim)
    chans = [3, 64, 128, pt_out_dim]
    for i in range(3):
        p[f'object_encoder.conv{i+1}.weight'] = xavier_normal((chans[i + 1], chans[i], 1), chans[i], chans[i + 1])
        p[f'object_encoder.conv{i+1}.bias'] = torch.zeros(chans[i + 1], dtype=torch.float64)
        p[f'object_encoder.bn{i+1}.weight'] = torch.ones(chans[i + 1], dtype=torch.float64)
        p[f'object_encoder.bn{i+1}.bias'] = torch.zeros(chans[i + 1], dtype=torch.float64)
        p[f'object_encoder.bn{i+1}.running_mean'] = torch.zeros(chans[i + 1], dtype=torch.float64)
        p[f'object_encoder.bn{i+1}.running_var'] = torch.ones(chans[i + 1], dtype=torch.float64)
    linear('object_embedding', emb_dim, pt_out_dim)
    n_layers = len(hidden_units) - 1
    for i in range(n_layers):
        in_c = hidden_units[i] * heads[i - 1] if i else hidden_units[i]   # gat.py:35
        out_c, h = hidden_units[i + 1], heads[i]
        pre = f'structure_encoder.layer_stack.{i}'
        w = glorot((h * out_c, in_c), in_c, h * out_c)
        p[pre + '.lin_src.weight'] = w
        p[pre + '.lin_dst.weight'] = w          # shared tensor in PyG 2.2.0 (in_channels is an int)
        p[pre + '.att_src'] = glorot((1, h, out_c), h, out_c)
        p[pre + '.att_dst'] = glorot((1, h, out_c), h, out_c)
        p[pre + '.bias'] = torch.zeros(h * out_c, dtype=torch.float64)
    linear('structure_embedding', emb_dim, hidden_units[-1] * heads[-1])
    p['fusion.weight'] = torch.ones((len(modules), 1), dtype=torch.float64)
    return {k: v.to(dtype) for k, v in p.items()}


# --------------------------------------------------------------------------------------------------
# row P : PointNetfeat.forward   (src/aligner/networks/pointnet.py:120-175)
# --------------------------------------------------------------------------------------------------
def pointnet_feat(x_t3p: torch.Tensor, w1, b1, w2, b2, w3, b3, return_argmax: bool = False):
    """x_t3p [T,3,P].  y[t,c] = max_p relu(W3 relu(W2 relu(W1 x + b1) + b2) + b3).

    pointnet.py:140-159: conv1/2/3 are Conv1d(k=1) == per-point Linear; the three BatchNorm calls
    (`self.bn1(x)` :141-142, :154-155, :158-159) DISCARD their result, so they do not enter the output.
    pointnet.py:161: torch.max over the point axis.
    """
    h = torch.einsum('oc,tcp->top', w1.reshape(w1.shape[0], -1), x_t3p) + b1[None, :, None]
    h = torch.relu(h)
    h = torch.einsum('oc,tcp->top', w2.reshape(w2.shape[0], -1), h) + b2[None, :, None]
    h = torch.relu(h)
    h = torch.einsum('oc,tcp->top', w3.reshape(w3.shape[0], -1), h) + b3[None, :, None]
    h = torch.relu(h)
    y, idx = torch.max(h, dim=2)
    return (y, idx) if return_argmax else y


def pointnet_bn_batch_stats(x_t3p, w1, b1, w2, b2, w3, b3):
    """Side effect of the discarded BN calls in train mode (pointnet.py:141-142,154-155,158-159):
    running_mean <- 0.9 rm + 0.1 mean, running_var <- 0.9 rv + 0.1 unbiased var of the PRE-ReLU conv
    outputs over (T,P).  Returns [(mean, unbiased_var)] for the three layers."""
    out = []
    z = torch.einsum('oc,tcp->top', w1.reshape(w1.shape[0], -1), x_t3p) + b1[None, :, None]
    out.append((z.mean(dim=(0, 2)), z.var(dim=(0, 2), unbiased=True)))
    z = torch.einsum('oc,tcp->top', w2.reshape(w2.shape[0], -1), torch.relu(z)) + b2[None, :, None]
    out.append((z.mean(dim=(0, 2)), z.var(dim=(0, 2), unbiased=True)))
    z = torch.einsum('oc,tcp->top', w3.reshape(w3.shape[0], -1), torch.relu(z)) + b3[None, :, None]
    out.append((z.mean(dim=(0, 2)), z.var(dim=(0, 2), unbiased=True)))
    return out


# --------------------------------------------------------------------------------------------------
# row G : GATConv (PyG 2.2.0, un-vendored) and MultiGAT.forward (src/aligner/networks/gat.py:40-48)
# --------------------------------------------------------------------------------------------------
def _canon_edges(edge_index: torch.Tensor, n: int):
    """PyG GATConv.forward, add_self_loops=True branch: remove_self_loops(edge_index) then
    add_self_loops(edge_index, num_nodes=N) -- every node gets exactly one self loop; duplicate
    (j->i) edges keep their multiplicity."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    keep = src != dst
    loop = torch.arange(n, dtype=torch.long)
    return torch.cat([src[keep], loop]), torch.cat([dst[keep], loop])


def gat_conv(x, edge_index, lin_w, att_src, att_dst, bias, negative_slope: float = 0.2):
    """One GATConv layer.  x [N,F]; edge_index [2,E] (row 0 = source j, row 1 = target i);
    lin_w [H*C,F]; att_* [1,H,C]; bias [H*C].  Returns [N,H*C]."""
    n = x.shape[0]
    h_, c_ = att_src.shape[1], att_src.shape[2]
    h = (x @ lin_w.t()).view(n, h_, c_)                    # lin_src == lin_dst (shared)
    a_s = (h * att_src).sum(-1)                            # [N,H]
    a_d = (h * att_dst).sum(-1)
    src, dst = _canon_edges(edge_index, n)
    e = F.leaky_relu(a_s[src] + a_d[dst], negative_slope)  # [E',H]
    # torch_geometric.utils.softmax: subtract per-target max, exp, divide by (scatter_sum + 1e-16)
    emax = torch.full((n, h_), -float('inf'), dtype=e.dtype)
    emax = emax.scatter_reduce(0, dst[:, None].expand(-1, h_), e, reduce='amax', include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros((n, h_), dtype=e.dtype).index_add(0, dst, ex) + 1e-16
    alpha = ex / den[dst]
    out = torch.zeros((n, h_, c_), dtype=x.dtype).index_add(0, dst, h[src] * alpha[:, :, None])
    return out.reshape(n, h_ * c_) + bias


def gat_conv_dense(x, edge_index, lin_w, att_src, att_dst, bias, negative_slope: float = 0.2):
    """Independent dense-count formulation of gat_conv (cross-check only)."""
    n = x.shape[0]
    h_, c_ = att_src.shape[1], att_src.shape[2]
    h = (x @ lin_w.t()).view(n, h_, c_)
    a_s = (h * att_src).sum(-1)
    a_d = (h * att_dst).sum(-1)
    src, dst = _canon_edges(edge_index, n)
    cnt = torch.zeros((n, n), dtype=x.dtype)
    cnt.index_put_((dst, src), torch.ones(src.shape[0], dtype=x.dtype), accumulate=True)   # cnt[i,j]
    e = F.leaky_relu(a_d[:, None, :] + a_s[None, :, :], negative_slope)       # [i,j,H]
    e_m = torch.where(cnt[:, :, None] > 0, e, torch.full_like(e, -float('inf')))
    m = e_m.max(dim=1, keepdim=True).values
    ex = torch.exp(e_m - m) * cnt[:, :, None]
    alpha = ex / (ex.sum(dim=1, keepdim=True) + 1e-16)
    out = torch.einsum('ijh,jhc->ihc', alpha, h)
    return out.reshape(n, h_ * c_) + bias


def multi_gat(x, edge_index, layers: List[Dict[str, torch.Tensor]], conv=gat_conv):
    """gat.py:40-48: dropout(p=0) no-op, GATConv, ELU between layers (not after the last)."""
    for i, lp in enumerate(layers):
        x = conv(x, edge_index, lp['lin_w'], lp['att_src'], lp['att_dst'], lp['bias'])
        if i + 1 < len(layers):
            x = F.elu(x)
    return x


def _gat_layers(params, n_layers=2):
    return [dict(lin_w=params[f'structure_encoder.layer_stack.{i}.lin_src.weight'],
                 att_src=params[f'structure_encoder.layer_stack.{i}.att_src'],
                 att_dst=params[f'structure_encoder.layer_stack.{i}.att_dst'],
                 bias=params[f'structure_encoder.layer_stack.{i}.bias']) for i in range(n_layers)]


# --------------------------------------------------------------------------------------------------
# row F : MultiModalFusion.forward (src/aligner/sg_aligner.py:30-35)
# --------------------------------------------------------------------------------------------------
def fusion(embs: List[torch.Tensor], weight: torch.Tensor):
    w = torch.softmax(weight, dim=0)                       # sg_aligner.py:32
    return torch.cat([w[i] * F.normalize(e) for i, e in enumerate(embs)], dim=1)   # :33-34


# --------------------------------------------------------------------------------------------------
# row E : MultiModalEncoder.forward (src/aligner/sg_aligner.py:71-137)
# --------------------------------------------------------------------------------------------------
def encoder_forward(params: Dict[str, torch.Tensor], data_dict: dict, modules: Sequence[str]):
    dt = params['object_embedding.weight'].dtype
    pts = data_dict['tot_obj_pts'].to(dt).permute(0, 2, 1)                       # :72
    attr = data_dict['tot_bow_vec_object_attr_feats'].to(dt)                     # :73
    rel = data_dict['tot_bow_vec_object_edge_feats'].to(dt)                      # :74
    pose = data_dict['tot_rel_pose'].to(dt)                                      # :75
    embs = {}
    for m in modules:
        if m == 'gat':                                                           # :84-112
            layers = _gat_layers(params)
            outs, so, se = [], 0, 0
            for b in range(int(data_dict['batch_size'])):
                for side in range(2):
                    n = int(data_dict['graph_per_obj_count'][b][side])
                    ne = int(data_dict['graph_per_edge_count'][b][side])
                    ei = data_dict['edges'][se:se + ne].t()                      # :100,103 (graph-local ids)
                    outs.append(multi_gat(pose[so:so + n], ei, layers))
                    so += n
                    se += ne
            emb = F.linear(torch.cat(outs), params['structure_embedding.weight'], params['structure_embedding.bias'])
        elif m == 'point':                                                       # :114-116
            y = pointnet_feat(pts, params['object_encoder.conv1.weight'], params['object_encoder.conv1.bias'],
                              params['object_encoder.conv2.weight'], params['object_encoder.conv2.bias'],
                              params['object_encoder.conv3.weight'], params['object_encoder.conv3.bias'])
            emb = F.linear(y, params['object_embedding.weight'], params['object_embedding.bias'])
        elif m == 'rel':                                                         # :118-119
            emb = F.linear(rel, params['meta_embedding_rel.weight'], params['meta_embedding_rel.bias'])
        elif m == 'attr':                                                        # :121-122
            emb = F.linear(attr, params['meta_embedding_attr.weight'], params['meta_embedding_attr.bias'])
        else:
            raise NotImplementedError(m)                                         # :124-125 ('pct' is out of scope)
        embs[m] = emb
    if len(modules) > 1:                                                         # :129-135
        embs['joint'] = fusion([embs[m] for m in modules], params['fusion.weight'])
    return embs


# --------------------------------------------------------------------------------------------------
# rows Q / ICL / IAL / ML / OL : src/aligner/losses.py
# --------------------------------------------------------------------------------------------------
def calculate_prob_dist(e1i, e2i, e1j, e2j, temp):
    """losses.py:5-15 -- note the GLOBAL scalar sums (.sum() with no dim) at :10-11."""
    d12 = torch.exp(e1i @ e2i.t() / temp)
    s11 = torch.exp(e1i @ e1j.t() / temp).sum()
    s12 = torch.exp(e1i @ e2j.t() / temp).sum()
    a = d12 / (s11 + 1e-9)
    b = d12 / (s12 + 1e-9)
    q_inv = 1.0 + 1.0 / (a + 1e-9) + 1.0 / (b + 1e-9)
    return 1.0 / (q_inv + 1e-9)


def _idx(data_dict, key):
    return torch.as_tensor(np.asarray(data_dict[key]), dtype=torch.long)


def icl_loss(emb, data_dict, temp: float = 0.1, alpha: float = 0.5):
    """losses.py:43-58 (temperature hard-coded 0.1 at :39; mean over the WHOLE A x A matrix :57)."""
    emb = F.normalize(emb, dim=1)
    e1i, e2i, e1j, e2j = (emb[_idx(data_dict, k)] for k in ('e1i', 'e2i', 'e1j', 'e2j'))
    qa = calculate_prob_dist(e1i, e2i, e1j, e2j, temp)
    qb = calculate_prob_dist(e2i, e1i, e2j, e1j, temp)          # indexed [i,j] UN-transposed (:54-56)
    return -torch.log(alpha * qa + (1 - alpha) * qb).mean()


def ial_loss(src_emb, ref_emb, data_dict, temp: float = 1.0, alpha: float = 0.5, zoom: float = 0.1):
    """losses.py:68-97 (temp 1.0 :63, zoom 0.1 :66).  KLDivLoss(reduction='sum', log_target=True)
    (input=log qm, target=qo) == sum exp(qo) * (qo - log qm)  (:92-94; the probability is fed as a
    log-probability -- reproduced as-is)."""
    src = F.normalize(src_emb, dim=1)
    ref = F.normalize(ref_emb, dim=1)
    ix = {k: _idx(data_dict, k) for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    qo_a = calculate_prob_dist(src[ix['e1i']], src[ix['e2i']], src[ix['e1j']], src[ix['e2j']], temp)
    qo_b = calculate_prob_dist(src[ix['e2i']], src[ix['e1i']], src[ix['e2j']], src[ix['e1j']], temp)
    qm_a = calculate_prob_dist(ref[ix['e1i']], ref[ix['e2i']], ref[ix['e1j']], ref[ix['e2j']], temp)
    qm_b = calculate_prob_dist(ref[ix['e2i']], ref[ix['e1i']], ref[ix['e2j']], ref[ix['e1j']], temp)
    loss_a = (torch.exp(qo_a) * (qo_a - qm_a.log())).sum()
    loss_b = (torch.exp(qo_b) * (qo_b - qm_b.log())).sum()
    return zoom * (alpha * loss_a + (1 - alpha) * loss_b)


def multi_loss(losses: List[torch.Tensor], log_vars: torch.Tensor):
    """CustomMultiLossLayer.forward, losses.py:28-34."""
    prec = torch.exp(-log_vars)
    out = 0
    for i, l in enumerate(losses):
        out = out + prec[i] * l + log_vars[i]
    return out


def overall_loss(output_dict, data_dict, modules: Sequence[str], log_vars_ial=None, log_vars_icl=None,
                 zoom: float = 0.1):
    """OverallLoss.forward, losses.py:114-152."""
    if len(modules) > 1:
        ial = multi_loss([ial_loss(output_dict[m], output_dict['joint'], data_dict) for m in modules],
                         log_vars_ial) * zoom                                             # :119-126
        icl_uni = multi_loss([icl_loss(output_dict[m], data_dict) for m in modules], log_vars_icl)  # :129-135
        icl_multi = icl_loss(output_dict['joint'], data_dict)                              # :140-141
        loss = ial + icl_uni + icl_multi                                                   # :143-144
    else:
        ial, icl_multi = 0.0, 0.0
        icl_uni = icl_loss(output_dict[modules[0]], data_dict)                             # :137-138
        loss = icl_uni                                                                     # :146
    return {'loss': loss, 'icl_loss_unimodal': icl_uni, 'icl_loss_multimodal': icl_multi, 'ial_loss': ial}


# --------------------------------------------------------------------------------------------------
# row S : per-pair similarity + ranking (src/inference/sgaligner/inference_align_reg.py:125-128)
# row K : utils/alignment.py
# --------------------------------------------------------------------------------------------------
def pair_similarity(emb_pair: torch.Tensor):
    emb = emb_pair / emb_pair.norm(dim=1)[:, None]          # :126 (no eps)
    sim = 1 - emb @ emb.t()                                  # :127
    return sim, torch.argsort(sim, dim=1, stable=True)       # :128 (reference sort is unstable; ties unspecified)


def rank_of(sim_row: np.ndarray, self_idx: int, tgt_idx: int) -> int:
    """1-based rank of tgt in the ascending-distance list of `self_idx`'s row with the self entry
    removed by value (alignment.py:6-8,16-18).  Ties broken by index (stable order)."""
    order = [int(i) for i in np.argsort(sim_row, kind='stable') if int(i) != self_idx]
    return order.index(tgt_idx) + 1


def alignment_metrics(sim: torch.Tensor, e1i: np.ndarray, e2i: np.ndarray, ks=(1, 2, 3, 4, 5),
                      modes=('2', '50', '100')):
    """alignment.py:3-57 on ONE pair (pair-local indices).  Returns dict(mrr list, hits{k:(c,t)}, sgar)."""
    s = sim.detach().cpu().numpy()
    ranks, pred, psim = [], [], []
    for a, b in zip(e1i, e2i):
        order = [int(i) for i in np.argsort(s[a], kind='stable') if int(i) != int(a)]
        ranks.append(order.index(int(b)) + 1)
        pred.append(order[0])
        psim.append(s[a][order[0]])
    hits = {k: (int(sum(r <= k for r in ranks)), len(ranks)) for k in ks}   # :13-25
    mrr = [1.0 / r for r in ranks]                                           # :3-11
    srt = np.argsort(psim, kind='stable')                                    # :39
    sgar = {}
    for mode in modes:                                                       # :42-55
        sel = srt[:2] if mode == '2' else (srt[:len(srt) // 2] if mode == '50' else srt)
        sgar[mode] = 0.0 if any(pred[i] != int(e2i[i]) for i in sel) else 1.0
    return {'mrr': mrr, 'hits': hits, 'sgar': sgar, 'ranks': ranks}


def node_corrs(sim: torch.Tensor, src_count: int, k: int = 1):
    """alignment.py:59-70."""
    s = sim.detach().cpu().numpy()
    out = []
    for i in range(src_count):
        order = [int(j) for j in np.argsort(s[i], kind='stable') if int(j) != i][:k]
        for j in order:
            if j >= src_count:
                out.append((i, j))
    return out


def evaluate_batch(embedding: torch.Tensor, data_dict: dict, ks=(1, 2, 3, 4, 5)):
    """AlignerRegTester.eval_step alignment block, inference_align_reg.py:98-143, restated without
    the in-place index mutation (:119-120)."""
    res = {'mrr': [], 'hits': {k: [0, 0] for k in ks}, 'sgar': {'2': [], '50': [], '100': []}}
    o = a = 0
    for b in range(int(data_dict['batch_size'])):
        n = int(data_dict['tot_obj_count'][b])
        na = int(data_dict['e1i_count'][b])
        e1 = np.asarray(data_dict['e1i'][a:a + na]) - o
        e2 = np.asarray(data_dict['e2i'][a:a + na]) - o
        if na:
            sim, _ = pair_similarity(embedding[o:o + n])
            m = alignment_metrics(sim, e1, e2, ks)
            res['mrr'] += m['mrr']
            for k in ks:
                res['hits'][k][0] += m['hits'][k][0]
                res['hits'][k][1] += m['hits'][k][1]
            for mode in res['sgar']:
                res['sgar'][mode].append(m['sgar'][mode])
        o += n
        a += na
    return res


# --------------------------------------------------------------------------------------------------
# whole step (what bench.py's cpu_baseline leg times): encoder fwd + OverallLoss fwd + backward
# --------------------------------------------------------------------------------------------------
def train_step(params: Dict[str, torch.Tensor], data_dict: dict, modules: Sequence[str],
               log_vars_ial: Optional[torch.Tensor] = None, log_vars_icl: Optional[torch.Tensor] = None):
    """Returns (output_dict, loss_dict, grads) like Trainer.train_step + backward
    (src/trainers/trainval_sgaligner.py:71-74, src/engine/epoch_based_trainer.py:91-93)."""
    names = [k for k in params if 'bn' not in k and not k.endswith('lin_dst.weight')]
    leaves = {k: params[k].detach().clone().requires_grad_(True) for k in names}
    p = dict(params)
    p.update(leaves)
    for k in list(p):
        if k.endswith('lin_dst.weight'):
            p[k] = p[k.replace('lin_dst', 'lin_src')]
    m = len(modules)
    dt = params['object_embedding.weight'].dtype
    lv_ial = (log_vars_ial if log_vars_ial is not None else torch.zeros(m, dtype=dt)).detach().clone().requires_grad_(True)
    lv_icl = (log_vars_icl if log_vars_icl is not None else torch.zeros(m, dtype=dt)).detach().clone().requires_grad_(True)
    out = encoder_forward(p, data_dict, modules)
    loss = overall_loss(out, data_dict, modules, lv_ial, lv_icl)
    loss['loss'].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    grads['log_vars_ial'] = lv_ial.grad if lv_ial.grad is not None else torch.zeros_like(lv_ial)
    grads['log_vars_icl'] = lv_icl.grad if lv_icl.grad is not None else torch.zeros_like(lv_icl)
    return out, loss, grads
