import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgaligner_amd import ops, _lib
from sgaligner_amd.ops import _p, _ptr_array, _stream
from sgaligner_amd.synthetic import make_batch_fast
L = _lib.lib()
dd = make_batch_fast(1, 20, 4, seed=3, device='cuda')
T = int(dd['tot_obj_pts'].shape[0])
s = ops.IndexSets.of(dd, 'cuda', T)
g = torch.Generator(device='cuda').manual_seed(0)
M = 3
zs = []
for m in range(M):
    e = torch.randn(T, 100, device='cuda', generator=g)
    z = torch.zeros((s.R + 32, 104), device='cuda'); nrm = torch.empty(s.R, device='cuda')
    L.sga_loss_gather(_p(e), T, 100, _p(s.idx), s.R, _p(z), 104, _p(nrm), _stream())
    zs.append(z)
beta = torch.tensor([0.2, 0.5, 0.3], device='cuda')
slots = 1 + L.sga_loss_slots()
nb = L.sga_loss_split_bytes(s.A, s.J1, s.J2)
zbs = []
for z in zs:
    zb = torch.empty(nb, device='cuda', dtype=torch.uint8)
    _lib.check(L.sga_loss_split_tables(_p(z), s.A, s.J1, s.J2, _p(zb), _stream()), 'split')
    zbs.append(zb)
gs = torch.ones(M + 1, 8, device='cuda', dtype=torch.float64)
dzs = [torch.zeros((s.R, 104), device='cuda') for _ in range(M)]
gam = torch.empty((slots, M), device='cuda', dtype=torch.float64)
_lib.check(L.sga_loss_multi_grad_bf16x3(_ptr_array(zbs), M, _p(beta), s.A, s.J1, s.J2, 0.1, 1.0, _p(gs), _ptr_array(dzs), _p(gam), 0, s.A, _stream()), 'gb')
torch.cuda.synchronize()
A, J1 = s.A, s.J1
for m in range(M):
    S = zs[m][:A] @ zs[m][2 * A:2 * A + J1].t()
    got = dzs[m][:A, :J1]
    print('table', m, 'max err', (S - got).abs().max().item())
    if m == 1:
        print(np.round(S.cpu().numpy()[:3, :14], 4)); print(np.round(got.cpu().numpy()[:3, :14], 4))
if os.environ.get('DBG_GAMMA'):
    # fp64 reference of gamma_m = sum over the four anchors x negatives products of dL/dS_J * S_m with gs = 1
    Z = [z[:s.R].double() for z in zs]
    A, J1, J2 = s.A, s.J1, s.J2
    X1 = [z[:A] for z in Z]; X2 = [z[A:2 * A] for z in Z]; N1 = [z[2 * A:2 * A + J1] for z in Z]; N2 = [z[2 * A + J1:] for z in Z]
    b = beta.double()
    ref = torch.zeros(M, dtype=torch.float64, device='cuda')
    for own, oth in ((X1, N1), (X1, N2), (X2, N2), (X2, N1)):
        Sm = [own[m] @ oth[m].t() for m in range(M)]
        SJ = sum(b[m] * Sm[m] for m in range(M))
        cJ = 10.0 * torch.exp(10.0 * SJ) + 1.0 * torch.exp(SJ)
        for m in range(M):
            ref[m] += (cJ * Sm[m]).sum()
    dzs2 = [torch.zeros((s.R, 104), device='cuda') for _ in range(M)]
    gam2 = torch.empty((slots, M), device='cuda', dtype=torch.float64)
    _lib.check(L.sga_loss_multi_grad(_ptr_array(zs), M, 100, _p(beta), s.A, s.J1, s.J2, 0.1, 1.0, _p(gs), _ptr_array(dzs2), _p(gam2), 0, s.A, _stream()), 'ga')
    torch.cuda.synchronize()
    print('gamma fp64 ref', ref.cpu().numpy(), '\n fp32 kernel  ', gam2[0].cpu().numpy(), '\n bf16x3 kernel', gam[0].cpu().numpy())
    # dZ reference for table 0, x1 rows and n1 rows
    m = 0
    dref = torch.zeros((s.R, 104), dtype=torch.float64, device='cuda')
    segs = {'x1': (0, A), 'x2': (A, 2 * A), 'n1': (2 * A, 2 * A + J1), 'n2': (2 * A + J1, s.R)}
    for (own, oth, ko, kt) in ((X1, N1, 'x1', 'n1'), (X1, N2, 'x1', 'n2'), (X2, N2, 'x2', 'n2'), (X2, N1, 'x2', 'n1')):
        Sm = [own[k] @ oth[k].t() for k in range(M)]
        SJ = sum(b[k] * Sm[k] for k in range(M))
        C = 10.0 * torch.exp(10.0 * Sm[m]) + torch.exp(Sm[m]) + b[m] * (10.0 * torch.exp(10.0 * SJ) + torch.exp(SJ))
        dref[segs[ko][0]:segs[ko][1]] += C @ oth[m]
        dref[segs[kt][0]:segs[kt][1]] += C.t() @ own[m]
    for name, d in (('fp32', dzs2[0]), ('bf16x3', dzs[0])):
        print(name, {k: float((d[lo:hi].double() - dref[lo:hi]).abs().max()) for k, (lo, hi) in segs.items()}, 'scale', float(dref.abs().max()))
if os.environ.get('DBG_CJ'):
    Z = [z[:s.R].double() for z in zs]
    A, J1 = s.A, s.J1
    b = beta.double()
    G, SG = int(os.environ['DBG_G']), int(os.environ['DBG_SG'])
    J2 = s.J2
    own = slice(0, A) if G == 0 else slice(A, 2 * A)
    n1, n2 = slice(2 * A, 2 * A + J1), slice(2 * A + J1, 2 * A + J1 + J2)
    oth = (n1, n2)[SG] if G == 0 else (n2, n1)[SG]
    SJ = sum(b[m] * (Z[m][own] @ Z[m][oth].t()) for m in range(M))
    cJ = 10.0 * torch.exp(10.0 * SJ) + torch.exp(SJ)
    got = dzs[0][own, :cJ.shape[1]]
    print('G', G, 'SG', SG, 'cj max err', float((cJ - got.double()).abs().max()), 'okf rows', dzs[1][own][:2, :16].cpu().numpy(), 'c0M', dzs[2][own][:1, :1].cpu().numpy())
if os.environ.get('DBG_GAM'):
    Z = [z[:s.R].double() for z in zs]
    A, J1, J2 = s.A, s.J1, s.J2
    b = beta.double()
    n1, n2 = slice(2 * A, 2 * A + J1), slice(2 * A + J1, 2 * A + J1 + J2)
    for G, own in ((0, slice(0, A)), (1, slice(A, 2 * A))):
        ref = torch.zeros((M, A, 4), dtype=torch.float64, device='cuda')
        for oth in ((n1, n2) if G == 0 else (n2, n1)):
            Sm = [Z[m][own] @ Z[m][oth].t() for m in range(M)]
            SJ = sum(b[m] * Sm[m] for m in range(M))
            cJ = 10.0 * torch.exp(10.0 * SJ) + torch.exp(SJ)
            for m in range(M):
                t = torch.zeros((A, 32), dtype=torch.float64, device='cuda'); t[:, :Sm[m].shape[1]] = cJ * Sm[m]
                ref[m] += t.reshape(A, 4, 8).sum(2)
        for m in range(M):
            got = dzs[m][own, :4].double()
            print('G', G, 'm', m, 'per-lane gamma max err', float((ref[m] - got).abs().max()), 'ref sum', float(ref[m].sum()), 'got sum', float(got.sum()))
    print('kernel gamma', gam[0].cpu().numpy())
