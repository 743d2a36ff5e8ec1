"""BASELINE.json configs[4] (stress shape) on one GPU: scenes of 256 objects x 2048 points, 1024-d embeddings (general per-table
loss kernels on the 1024-d tables + 3072-d joint), fwd + loss + bwd, and the ranking on exact-fp32 vs fp16-input MFMA.
  python tools/bench_c5.py [pairs=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgaligner_amd import ops
from sgaligner_amd.aligner.losses import CustomMultiLossLayer, OverallLoss
from sgaligner_amd.aligner.sg_aligner import MultiModalEncoder
from sgaligner_amd.synthetic import make_batch_fast
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = sys.argv[2] if len(sys.argv) > 2 else 'f32'          # f32 | f16 (fp16-input loss GEMMs, csrc/wide16.hip)
ops.set_mfma_mode(MODE)
mods = ['point', 'gat', 'rel']
torch.manual_seed(2)
model = MultiModalEncoder(modules=mods, rel_dim=41, attr_dim=164, emb_dim=1024).cuda()
loss_fn = OverallLoss(CustomMultiLossLayer(3).cuda(), CustomMultiLossLayer(3).cuda(), 'cuda',
                      {'zoom': 0.1, 'wt_align_loss': 1.0, 'wt_contrastive_loss': 1.0, 'modules': mods})
dd = make_batch_fast(B, 256, 2048, seed=4, device='cuda')
params = list(model.parameters()) + list(loss_fn.parameters())


def step():
    for p in params:
        p.grad = None
    res = loss_fn(model(dd), dd)
    res['loss'].backward()
    return res


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for _ in range(n):
    res = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'[{MODE}] configs[4] shape, {B} pairs x 256 objects x 2048 pts, D = 1024, P+S+R: {dt * 1e3:.1f} ms/step = {B / dt:.1f} pairs/s, '
      f'peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, loss {float(res["loss"].detach()):.4e}')
ddv = make_batch_fast(B, 256, 8, seed=4, device='cuda', anchors='val')
with torch.no_grad():
    emb = model({**ddv, 'tot_obj_pts': dd['tot_obj_pts']})['joint']
for f16 in (False, True):
    ops.simrank(emb, ddv['tot_obj_count'], ddv['e1i'], ddv['e2i'], 1, f16=f16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        r = ops.simrank(emb, ddv['tot_obj_count'], ddv['e1i'], ddv['e2i'], 1, f16=f16)
    torch.cuda.synchronize()
    print(f'  ranking of {len(ddv["e1i"])} anchors on the 3072-d joint table, {"fp16-input" if f16 else "exact fp32"} MFMA: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms')
