import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 64)
steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
dd = make_batch_fast(B, N, 512, seed=43, device='cuda')
def run(mode):
    ops.set_mfma_mode(mode)
    out, loss = steps.forward_backward(dd)
    torch.cuda.synchronize()
    g = {n: p.grad.detach().double().clone() for n, p in steps.model.named_parameters() if p.grad is not None}
    for tag, layer in (('ial', steps.multi_loss_layer_ial), ('icl', steps.multi_loss_layer_icl)):
        g['log_vars_' + tag] = next(layer.parameters()).grad.detach().double().clone()
    return float(loss['loss']), g, {k: v.detach().clone() for k, v in out.items()}
l0, g0, o0 = run('f32'); l1, g1, o1 = run('f32'); l2, g2, o2 = run('bf16x3')
ops.set_mfma_mode('f32')
print('loss', l0, l1, l2)
for k in o0:
    print('emb', k, 'f32 rerun', float((o0[k] - o1[k]).abs().max()), 'bf16x3', float((o0[k] - o2[k]).abs().max()), 'scale', float(o0[k].abs().max()))
for n in g0:
    sc = g0[n].abs().max().item()
    print(f'{n:55s} max|g| {sc:10.3e}  f32 rerun {float((g0[n]-g1[n]).abs().max())/max(sc,1e-30):9.2e}  bf16x3 {float((g0[n]-g2[n]).abs().max())/max(sc,1e-30):9.2e}')
