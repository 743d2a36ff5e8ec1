"""Fused loss on the HIP path (symmetric / ordered anchors x anchors walk) against the fp64 oracle on the same tables and fusion weights:
table gradients, and their COLUMN SUMS (what a bias gradient upstream is made of: the cancellation-heavy quantity).
  python tools/dbg/sym_vs_fp64.py [pairs=192] [objects=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import sga_oracle as O
from sgaligner_amd import ops
from sgaligner_amd.aligner import losses as L
from sgaligner_amd.aligner.sg_aligner import MultiModalFusion
from sgaligner_amd.synthetic import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mods = ['point', 'gat', 'rel']
dd = make_batch(B, N, 1, seed=3, ragged=True)
T = int(dd['tot_obj_count'].sum())
torch.manual_seed(0)
base = {k: torch.randn(T, 100, dtype=torch.float64) for k in mods}
w0 = torch.randn(3, 1, dtype=torch.float64) * 0.5
lv1, lv2 = 0.3 * torch.randn(3, dtype=torch.float64), 0.3 * torch.randn(3, dtype=torch.float64)
# fp64 oracle
e64 = {k: base[k].clone().requires_grad_(True) for k in mods}
w64 = w0.clone().requires_grad_(True)
out64 = dict(e64); out64['joint'] = O.fusion([e64[k] for k in mods], w64)
ref = O.overall_loss(out64, dd, mods, lv1.clone().requires_grad_(True), lv2.clone().requires_grad_(True))
ref['loss'].backward()
print(f'{B} pairs x {N} objects: T = {T}, anchors = {len(dd["e1i"])}; fp64 loss {ref["loss"].item():.10e}')


def run(sym):
    ops.AA_SYMMETRIC = sym
    e = {k: base[k].float().cuda().requires_grad_(True) for k in mods}
    fus = MultiModalFusion(3).cuda()
    ial, icl = L.CustomMultiLossLayer(3).cuda(), L.CustomMultiLossLayer(3).cuda()
    with torch.no_grad():
        fus.weight.copy_(w0.float()); ial.log_vars.copy_(lv1.float()); icl.log_vars.copy_(lv2.float())
    out = dict(e); out['joint'] = fus([e[k] for k in mods])
    fn = L.OverallLoss(ial, icl, 'cuda', {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
    res = fn(out, dd)
    res['loss'].backward()
    torch.cuda.synchronize()
    return res['loss'].item(), {k: e[k].grad.double().cpu() for k in mods}, fus.weight.grad.double().cpu()


for name, sym in (('symmetric', True), ('ordered', False)):
    l, g, gw = run(sym)
    errs, cerrs = [], []
    for k in mods:
        r = e64[k].grad
        errs.append(((g[k] - r).abs().max() / r.abs().max()).item())
        cerrs.append(((g[k].sum(0) - r.sum(0)).abs().max() / r.sum(0).abs().max()).item())
    print(f'{name:10s}: loss rel err {abs(l - ref["loss"].item()) / abs(ref["loss"].item()):.2e}; table gradients {max(errs):.2e} of their maximum; '
          f'their column sums {max(cerrs):.2e} of the largest column sum; fusion weights {((gw - w64.grad).abs().max() / w64.grad.abs().max()).item():.2e}')
ops.AA_SYMMETRIC = True
