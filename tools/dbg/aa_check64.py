"""fp64 reference of two A x A terms (joint ICL, IAL_a of table 0) at any size, from the kernels' own global sums (fp64, saved by the
autograd node), against the classic forward kernel and the one-pass (backward-with-terms) kernel.  python tools/dbg/aa_check64.py B N"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch_fast
B, N = int(sys.argv[1]), int(sys.argv[2])
dd = make_batch_fast(B, N, 4, seed=3, device='cuda')
T = int(dd['tot_obj_pts'].shape[0])
g = torch.Generator(device='cuda').manual_seed(0)
tabs = [torch.randn(T, 100, device='cuda', generator=g).requires_grad_(True) for _ in range(3)]
w = torch.ones(3, 1, device='cuda', requires_grad=True)
M = 3
hint = (torch.rand(3 * M + 1, device='cuda', generator=g) + 0.5) * 1e-6
res = {}
for name, h in (('classic', None), ('onepass', hint)):
    sums, s = ops.fused_contrastive_terms(tabs, w, dd, coef_hint=h)
    res[name] = sums.detach().double().clone()
    gsums = sums.grad_fn.saved_tensors[0].double().clone()          # [(M+1), 8] global sums: fam*2 + temp
    del sums
torch.cuda.synchronize()
print('classic', res['classic'].tolist())
print('onepass', res['onepass'].tolist())
with torch.no_grad():
    Z = [torch.nn.functional.normalize(t.detach().double(), dim=1) for t in tabs]
    e1i = torch.from_numpy(np.asarray(dd['e1i'])).cuda().long(); e2i = torch.from_numpy(np.asarray(dd['e2i'])).cuda().long()
    beta = torch.full((3,), 1.0 / 3.0, dtype=torch.float64, device='cuda')
    def q(d, sa, sb):
        u = d / (sa + 1e-9) + 1e-9; v = d / (sb + 1e-9) + 1e-9
        return 1.0 / (1.0 + 1.0 / u + 1.0 / v + 1e-9)
    A = len(e1i)
    icl_j = ial_a0 = icl_0 = 0.0
    X1 = [z[e1i] for z in Z]; X2 = [z[e2i] for z in Z]
    for c in range(0, A, 512):
        Sx = [X1[m][c:c + 512] @ X2[m].t() for m in range(3)]        # S_m[i, j] = x1_i . x2_j
        Sy = [X2[m][c:c + 512] @ X1[m].t() for m in range(3)]        # S_m[j, i] seen from row i: x2_i . x1_j
        SJx = sum(beta[m] * Sx[m] for m in range(3)); SJy = sum(beta[m] * Sy[m] for m in range(3))
        gs = gsums
        # ICL joint (tau 0.1): sums row M, temp 0 -> indices fam*2 + 0
        qa = q(torch.exp(SJx / 0.1), gs[3, 0], gs[3, 2]); qb = q(torch.exp(SJy / 0.1), gs[3, 4], gs[3, 6])
        icl_j += (-torch.log(0.5 * qa + 0.5 * qb)).sum().item()
        qa0 = q(torch.exp(Sx[0] / 0.1), gs[0, 0], gs[0, 2]); qb0 = q(torch.exp(Sy[0] / 0.1), gs[0, 4], gs[0, 6])
        icl_0 += (-torch.log(0.5 * qa0 + 0.5 * qb0)).sum().item()
        # IAL_a table 0 (tau 1): qo from table 0, qm from the joint, direction A (e1i -> e2i): temp index 1
        qo = q(torch.exp(Sx[0]), gs[0, 1], gs[0, 3]); qm = q(torch.exp(SJx), gs[3, 1], gs[3, 3])
        ial_a0 += (torch.exp(qo) * (qo - torch.log(qm))).sum().item()
    print(f'fp64: ICL_0 {icl_0:.6e}  ICL_joint {icl_j:.6e}  IAL_a0 {ial_a0:.6e}')
    for name in res:
        r = res[name]
        print(f'{name}: ICL_0 rel err {abs(r[0].item() - icl_0) / icl_0:.2e}  ICL_joint {abs(r[3].item() - icl_j) / icl_j:.2e}  IAL_a0 {abs(r[4].item() - ial_a0) / ial_a0:.2e}')
