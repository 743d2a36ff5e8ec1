"""Loss gradients when one table's rows are nearly identical (what meta_embedding_rel produces from bag-of-words rows that are almost all the
same: S ~ 1 for every pair, the gradient wrt the un-normalised rows is the small tangential remainder of a large radial sum): exact-fp32 sweeps
vs the split-fp16 sweeps vs the fp64 oracle.  python tools/dbg/f16x2_degenerate.py [pairs=16] [objects=40] [spread=1e-3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import sga_oracle as O
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
spread = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
mods = ['point', 'gat', 'rel']
dd = make_batch(B, N, 1, seed=3)
T = int(dd['tot_obj_count'].sum())
g = torch.Generator().manual_seed(0)
base = [torch.randn(T, 100, generator=g, dtype=torch.float64) for _ in mods]
v0 = torch.randn(1, 100, generator=g, dtype=torch.float64)
base[2] = v0 + spread * torch.randn(T, 100, generator=g, dtype=torch.float64)          # 'rel': nearly identical rows
base = [b.float().double() for b in base]                                             # fp32-representable inputs
w0 = torch.tensor([[0.7], [1.2], [0.9]], dtype=torch.float64)
lv1 = torch.tensor([0.1, -0.2, 0.05], dtype=torch.float64)
lv2 = torch.tensor([-0.1, 0.15, 0.0], dtype=torch.float64)
eo = {k: base[i].clone().requires_grad_(True) for i, k in enumerate(mods)}
wo = w0.clone().requires_grad_(True)
out_o = dict(eo)
out_o['joint'] = O.fusion([eo[k] for k in mods], wo)
ref = O.overall_loss(out_o, dd, mods, lv1.clone().requires_grad_(True), lv2.clone().requires_grad_(True))
ref['loss'].backward()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from test_fullsize_gpu import _run_overall
res = {}
for mode in ('f32', 'f16x2'):
    ops.set_mfma_mode(mode)
    res[mode] = _run_overall([b.float().cuda() for b in base], dd, mods, w0.float().cuda(), lv1.float().cuda(), lv2.float().cuda(), fused=True)
ops.set_mfma_mode('f32')
print(f'{B} pairs x {N} objects, rel rows = v0 + {spread} noise; loss {ref["loss"].item():.6f}  f32 {res["f32"][0]:.6f}  f16x2 {res["f16x2"][0]:.6f}')
for k in mods:
    gref = eo[k].grad
    for mode in ('f32', 'f16x2'):
        gg = res[mode][1][k].cpu().double()
        e = (gg - gref).abs().max().item() / gref.abs().max().item()
        cs = (gg.sum(0) - gref.sum(0)).abs().max().item() / max(1e-300, gref.sum(0).abs().max().item())
        print(f'  dE[{k}] {mode:6s}: max err / max |dE| {e:.3e}   column-sum err / max |column sum| {cs:.3e}   (max |dE| {gref.abs().max().item():.3e}, max |colsum| {gref.sum(0).abs().max().item():.3e})')
