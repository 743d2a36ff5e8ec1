import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch, to_device
from sgaligner_amd.trainer import AlignerSteps
dd = to_device(make_batch(3, 20, 96, seed=8, ragged=True), 'cuda')
steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=3)
enc = steps.model.object_encoder
ws = [enc.conv1.weight.reshape(64, 3).contiguous(), enc.conv1.bias, enc.conv2.weight.reshape(128, 64).contiguous(), enc.conv2.bias,
      enc.conv3.weight.reshape(256, 128).contiguous(), enc.conv3.bias]
ws = [w.detach() for w in ws]
x = dd['tot_obj_pts']
res = {}
for mode in ('f32', 'f16x2'):
    ops.set_mfma_mode(mode)
    y, am = ops.pointnet_forward(x, *ws, want_argmax=True)
    res[mode] = (y.clone(), am.clone())
ops.set_mfma_mode('f32')
y0, a0 = res['f32']; y1, a1 = res['f16x2']
print('T', x.shape, 'max |dy|', (y0 - y1).abs().max().item(), 'argmax equal', (a0 == a1).float().mean().item(), 'differ where y>0:', ((a0 != a1) & (y0 > 0)).sum().item(), 'of', (y0 > 0).sum().item())
d = (a0 != a1) & (y0 > 0)
if d.any():
    t, c = d.nonzero()[0].tolist()
    print('example object', t, 'channel', c, 'argmax', a0[t, c].item(), a1[t, c].item(), 'y', y0[t, c].item(), y1[t, c].item())
    print('points equal?', torch.equal(x[t, a0[t, c]], x[t, a1[t, c]]), x[t, a0[t, c]].tolist(), x[t, a1[t, c]].tolist())
