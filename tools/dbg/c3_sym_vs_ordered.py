"""configs[2] on one GPU: one training step with the symmetric anchors x anchors walk against the same step with the ordered walk (and
against the two-pass path): loss and every parameter gradient.  python tools/dbg/c3_sym_vs_ordered.py [pairs=4096] [objects=128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
dd = make_batch_fast(B, N, 512, seed=5, device='cuda')


def run(sym, onepass=True):
    ops.AA_SYMMETRIC, ops.FUSED_AA_ONEPASS = sym, onepass
    for p in steps.model.parameters():
        p.grad = None
    for p in steps.loss_func.parameters():
        p.grad = None
    _, loss = steps.forward_backward(dd)
    torch.cuda.synchronize()
    named = list(steps.model.named_parameters()) + [('loss.' + n, p) for n, p in steps.loss_func.named_parameters()]
    return float(loss['loss']), {n: p.grad.detach().clone() for n, p in named if p.grad is not None}


ref_l, ref = run(False)
for name, (sym, one) in {'symmetric': (True, True), 'ordered again (rerun noise)': (False, True), 'two-pass': (False, False)}.items():
    l, g = run(sym, one)
    worst = max(((g[k] - ref[k]).abs().max().item() / max(1e-30, ref[k].abs().max().item()), k) for k in ref)
    print(f'{name:28s}: loss rel diff {abs(l - ref_l) / abs(ref_l):.2e}; worst parameter-gradient difference {worst[0]:.2e} of its own maximum ({worst[1]})')
ops.AA_SYMMETRIC, ops.FUSED_AA_ONEPASS = True, True
