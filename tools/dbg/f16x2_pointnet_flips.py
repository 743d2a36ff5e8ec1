"""How many arg-max points / ReLU masks of the object encoder's forward differ between the exact-fp32 mode and 'f16x2' (split + near-tie re-run)
on the bench's own batch and weights.   python tools/dbg/f16x2_pointnet_flips.py [pairs=4096] [objects=128] [eps ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops, _lib
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nobj = int(sys.argv[2]) if len(sys.argv) > 2 else 128
eps_list = [float(e) for e in sys.argv[3:]] or [None]
dd = make_batch_fast(pairs, nobj, 512, seed=42, device='cuda')
steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
enc = steps.model.object_encoder
ws = [w.detach() for w in (enc.conv1.weight.reshape(64, 3).contiguous(), enc.conv1.bias, enc.conv2.weight.reshape(128, 64).contiguous(), enc.conv2.bias,
                           enc.conv3.weight.reshape(256, 128).contiguous(), enc.conv3.bias)]
x = dd['tot_obj_pts']
ops.set_mfma_mode('f32')
y0, a0 = ops.pointnet_forward(x, *ws, want_argmax=True)
print('objects', tuple(x.shape), 'y > 0 fraction', float((y0 > 0).float().mean()))
for eps in eps_list:
    ops.POINTNET_TIE_EPS = -1.0 if eps is None else eps
    for mode in ('f16x2p', 'f16x2'):
        ops.set_mfma_mode(mode)
        y1, a1 = ops.pointnet_forward(x, *ws, want_argmax=True)
        da = (a0 != a1) & ((y0 > 0) | (y1 > 0))
        dm = (y0 > 0) != (y1 > 0)
        extra = ''
        if mode == 'f16x2':
            extra = f'; eps {ops.POINTNET_TIE_EPS if ops.POINTNET_TIE_EPS >= 0 else 2.0 ** -17:.3g}: {int(ops.POINTNET_LAST_REDO[0]) / x.shape[0] * 100:.2f} % of the objects re-run'
        print(f'{mode}: arg-max differs (live channels) {int(da.sum())}, ReLU mask differs {int(dm.sum())} of {a0.numel()}; max |dy| {float((y0 - y1).abs().max()):.3e} '
              f'(max |y| {float(y0.abs().max()):.3e}){extra}')
        if mode == 'f16x2' and da.any():
            t, c = da.nonzero()[0].tolist()
            print('   e.g. object', t, 'channel', c, 'points', int(a0[t, c]), int(a1[t, c]), 'y', float(y0[t, c]), float(y1[t, c]),
                  'same coordinates:', torch.equal(x[t, a0[t, c]], x[t, a1[t, c]]))
ops.set_mfma_mode('f32')
