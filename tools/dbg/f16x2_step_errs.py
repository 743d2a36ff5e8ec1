"""Small train step in f16x2 mode (and with pieces of it switched off) against the CPU oracle: per-parameter errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import sga_oracle as O
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch, to_device
from sgaligner_amd.trainer import AlignerSteps
mods = ['point', 'gat', 'rel']
dd = make_batch(3, 20, 96, seed=8, ragged=True)
steps = AlignerSteps(mods, device='cuda', seed=3)
params = {k: v.detach().cpu().clone() for k, v in steps.model.state_dict().items() if 'num_batches' not in k}
out_o, loss_o, grads_o = O.train_step(params, dd, mods)
for mode in ('f32', 'f16x2'):
    ops.set_mfma_mode(mode)
    out, loss = steps.forward_backward(to_device(dd, 'cuda'))
    torch.cuda.synchronize()
    print(mode, 'loss', loss['loss'].item(), loss_o['loss'].item(), 'emb err', max((out[k].detach().cpu() - out_o[k].detach()).abs().max().item() for k in out_o))
    for name, p in steps.model.named_parameters():
        if name in grads_o and p.grad is not None:
            ref = grads_o[name]
            print(f'   {name:45s} err {(p.grad.cpu() - ref).abs().max().item():.3e}  max |ref| {ref.abs().max().item():.3e}')
ops.set_mfma_mode('f32')
