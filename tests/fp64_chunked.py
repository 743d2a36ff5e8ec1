"""TEST INFRASTRUCTURE (never imported by the product): the batch-global OverallLoss of the reference and its gradient in fp64, evaluated in
CHUNKS with plain torch ops on whatever device the tables live on -- so that it finishes at BASELINE configs[2] (4096 pairs x 128 objects:
A = 155 648 anchors, 368 640 negatives per side), where the oracle's dense restatement (oracle/sga_oracle.py: overall_loss) would need terabytes.
It calls none of the library's kernels: it is the independent checker of the loss gradient at the headline size (round-3 review, item 3).

Follows src/aligner/losses.py line by line, like the oracle:
  calculate_prob_dist :5-15   d12 = exp(e1i e2i^T / t), s11 = sum exp(e1i e1j^T / t), s12 = sum exp(e1i e2j^T / t) (GLOBAL sums),
                              q = 1 / (1 + 1/(d12/(s11+1e-9)+1e-9) + 1/(d12/(s12+1e-9)+1e-9) + 1e-9)
  ICLLoss :43-58              t = 0.1; qa = Q(e1i,e2i,e1j,e2j), qb = Q(e2i,e1i,e2j,e1j) (indexed [i,j] un-transposed); -mean log(a qa + (1-a) qb)
  IALLoss :68-97              t = 1.0; qo from the unimodal table, qm from the joint table; 0.1 (a sum exp(qo_a)(qo_a - log qm_a) + (1-a) sum exp(qo_b)(...))
  CustomMultiLossLayer :28-34 sum exp(-lv_i) L_i + lv_i
  OverallLoss :114-152        ial = ML_ial([IAL(m, joint)]) * zoom; icl_uni = ML_icl([ICL(m)]); icl_multi = ICL(joint); loss = ial + icl_uni + icl_multi
and MultiModalFusion (sg_aligner.py:30-35) for the joint table.  tests/test_fp64_chunked_gpu.py pins it on the oracle at sizes the oracle runs.

Structure: (1) the global sums of every table / temperature / family by chunked matmul + exp (no graph); (2) the anchors x anchors terms one
anchor-row chunk at a time under autograd, with the normalised anchor rows and the 16 sums per table as leaves (each chunk's graph is freed by
its own backward); (3) the gradient that arrives through the sums, dX += (g/t) exp(X N^T / t) N and dN += (g/t) exp(.)^T X, chunked, no graph;
(4) normalisation, fusion and the row gathers by one small autograd graph."""
import numpy as np
import torch
import torch.nn.functional as F

TAU_ICL, TAU_IAL, ALPHA, IAL_ZOOM = 0.1, 1.0, 0.5, 0.1


def _q(d, sa, sb):
    a = d / (sa + 1e-9)
    b = d / (sb + 1e-9)
    return 1.0 / (1.0 + 1.0 / (a + 1e-9) + 1.0 / (b + 1e-9) + 1e-9)


def overall_loss_fp64(tables, fusion_weight, lv_ial, lv_icl, data_dict, zoom=0.1, rows=1024, rows_aa=128, want_grad=True, device=None, timings=None):
    """tables: list of M fp32/fp64 [T, D] tensors (module order), fusion_weight [M, 1], lv_ial / lv_icl [M].
    Returns dict(loss, icl_uni, icl_multi, ial) as python floats and, if want_grad, dE (list of M [T, D] fp64), dw [M, 1], dlv_ial, dlv_icl.
    timings: an optional dict that receives the wall seconds of the four phases (synchronised)."""
    import time

    def _tick(name, t0):
        if timings is not None:
            torch.cuda.synchronize()
            timings[name] = timings.get(name, 0.0) + time.time() - t0
        return time.time()
    t_ph = time.time()
    dev = device or tables[0].device
    M = len(tables)
    assert M > 1
    idx = {k: torch.as_tensor(np.asarray(data_dict[k]), dtype=torch.long, device=dev) for k in ('e1i', 'e2i', 'e1j', 'e2j')}
    A, J1, J2 = len(idx['e1i']), len(idx['e1j']), len(idx['e2j'])
    E = [t.detach().to(dev, torch.float64).requires_grad_(want_grad) for t in tables]
    w = fusion_weight.detach().to(dev, torch.float64).requires_grad_(want_grad)
    l1 = lv_ial.detach().to(dev, torch.float64).requires_grad_(want_grad)
    l2 = lv_icl.detach().to(dev, torch.float64).requires_grad_(want_grad)
    # the M + 1 normalised tables (graph kept for step 4)
    sw = torch.softmax(w, dim=0)
    joint = torch.cat([sw[i] * F.normalize(e) for i, e in enumerate(E)], dim=1)           # sg_aligner.py:32-34
    Zn = [F.normalize(e, dim=1) for e in E] + [F.normalize(joint, dim=1)]               # losses.py:44,73-74
    nt = M + 1
    parts = [[z[idx[k]] for k in ('e1i', 'e2i', 'e1j', 'e2j')] for z in Zn]              # X1, X2, N1, N2 (graph)
    P = [[p.detach() for p in tp] for tp in parts]
    temps = (TAU_ICL, TAU_IAL)

    # (1) global sums [table][temp][family]: s11 = X1.N1, s12 = X1.N2, s22 = X2.N2, s21 = X2.N1
    fam = ((0, 2), (0, 3), (1, 3), (1, 2))
    sums = torch.zeros(nt, 2, 4, dtype=torch.float64, device=dev)
    with torch.no_grad():
        for t in range(nt):
            for f, (xi, ni) in enumerate(fam):
                X, N = P[t][xi], P[t][ni]
                for lo in range(0, A, rows):
                    S = X[lo:lo + rows] @ N.t()
                    for ti, tau in enumerate(temps):
                        sums[t, ti, f] += torch.exp(S / tau).sum()      # (two passes per temperature; kept literal: this is the checker)
    s_leaf = sums.clone().requires_grad_(want_grad)
    t_ph = _tick('1_global_sums', t_ph)

    # (2) anchors x anchors terms, one anchor-row chunk at a time
    Xl = [[P[t][0].clone().requires_grad_(want_grad), P[t][1].clone().requires_grad_(want_grad)] for t in range(nt)]
    acc = {'icl': torch.zeros(nt, dtype=torch.float64, device=dev), 'ial': torch.zeros(M, dtype=torch.float64, device=dev)}
    n_el = float(A) * float(A)
    w_ial = (zoom * torch.exp(-l1.detach())).clone()
    w_icl = torch.cat([torch.exp(-l2.detach()), torch.ones(1, dtype=torch.float64, device=dev)])

    def aa_chunk(x1c, x2c, x1, x2, s_l):
        """The terms of one anchor-row chunk: (ICL per table [nt], IAL per modality [M], their weighted sum).  Plain torch ops, literal formulas
        (through torch.compile the same function measured 3 x SLOWER at 1024 pairs -- fp64 codegen + recompiles -- and is not used)."""
        q = {}
        for t in range(nt):
            S12 = x1c[t] @ x2[t].t()                          # e1i[i] . e2i[j]
            S21 = x2c[t] @ x1[t].t()                          # e2i[i] . e1i[j]  (qb is indexed [i, j] un-transposed)
            for ti, tau in enumerate(temps):
                qa = _q(torch.exp(S12 / tau), s_l[t, ti, 0], s_l[t, ti, 1])
                qb = _q(torch.exp(S21 / tau), s_l[t, ti, 2], s_l[t, ti, 3])
                q[(t, ti)] = (qa, qb)
        chunk_icl = [-(torch.log(ALPHA * q[(t, 0)][0] + (1 - ALPHA) * q[(t, 0)][1])).sum() / n_el for t in range(nt)]
        qm_a, qm_b = q[(M, 1)]
        chunk_ial = []
        for m in range(M):
            qo_a, qo_b = q[(m, 1)]
            la = (torch.exp(qo_a) * (qo_a - qm_a.log())).sum()
            lb = (torch.exp(qo_b) * (qo_b - qm_b.log())).sum()
            chunk_ial.append(IAL_ZOOM * (ALPHA * la + (1 - ALPHA) * lb))
        ci, ca = torch.stack(chunk_icl), torch.stack(chunk_ial)
        # d loss / d (terms) is constant: loss = zoom sum_m e^{-l1_m} IAL_m + sum_m e^{-l2_m} ICL_m + ICL_joint (+ the log_vars themselves)
        return ci, ca, (w_ial * ca).sum() + (w_icl * ci).sum()
    for lo in range(0, A, rows_aa):
        hi = min(A, lo + rows_aa)
        args = ([Xl[t][0][lo:hi] for t in range(nt)], [Xl[t][1][lo:hi] for t in range(nt)], [Xl[t][0] for t in range(nt)], [Xl[t][1] for t in range(nt)], s_leaf)
        ci, ca, contrib = aa_chunk(*args)
        acc['icl'] += ci.detach()
        acc['ial'] += ca.detach()
        if want_grad:
            contrib.backward()
        del ci, ca, contrib, args
    t_ph = _tick('2_anchors_x_anchors', t_ph)
    ial = (torch.exp(-l1) * acc['ial'] + l1).sum() * zoom
    icl_uni = (torch.exp(-l2) * acc['icl'][:M] + l2).sum()
    icl_multi = acc['icl'][M]
    loss = ial + icl_uni + icl_multi
    out = {'loss': float(loss.detach()), 'ial': float(ial.detach()), 'icl_uni': float(icl_uni.detach()), 'icl_multi': float(icl_multi.detach())}
    if not want_grad:
        return out

    # (3) the gradient through the global sums
    gs = s_leaf.grad                                           # [nt, 2, 4]
    dP = [[Xl[t][0].grad.clone(), Xl[t][1].grad.clone(), torch.zeros_like(P[t][2]), torch.zeros_like(P[t][3])] for t in range(nt)]
    with torch.no_grad():
        for t in range(nt):
            for f, (xi, ni) in enumerate(fam):
                X, N = P[t][xi], P[t][ni]
                for lo in range(0, A, rows):
                    S = X[lo:lo + rows] @ N.t()
                    C = (gs[t, 0, f] / temps[0]) * torch.exp(S / temps[0]) + (gs[t, 1, f] / temps[1]) * torch.exp(S / temps[1])
                    dP[t][xi][lo:lo + rows] += C @ N
                    dP[t][ni] += C.t() @ X[lo:lo + rows]
                    del S, C
    t_ph = _tick('3_gradient_through_sums', t_ph)
    # (4) gathers, normalisation, fusion; the log_vars see the term values
    tot = sum((parts[t][k] * dP[t][k]).sum() for t in range(nt) for k in range(4))
    head = (torch.exp(-l1) * acc['ial'] + l1).sum() * zoom + (torch.exp(-l2) * acc['icl'][:M] + l2).sum()
    (tot + head).backward()
    t_ph = _tick('4_gathers', t_ph)
    out.update(dE=[e.grad for e in E], dw=w.grad, dlv_ial=l1.grad, dlv_icl=l2.grad,
               dZ=[[d for d in dP[t]] for t in range(nt)])
    return out
