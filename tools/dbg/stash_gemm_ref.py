"""How far are the stash GEMMs (N = 104) from what the vendor library does on the same shapes?  torch.matmul (hipBLASLt / rocBLAS, fp32) is
only a yardstick here -- the product path does not call it.  python tools/dbg/stash_gemm_ref.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import ops
torch.backends.cuda.matmul.allow_tf32 = False
A, ns = 155648, 4608
m1 = torch.randn(A, ns, device='cuda')
x1 = torch.randn(ns, 104, device='cuda'); x2 = torch.randn(A, 104, device='cuda')


def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


fl = 2.0 * A * ns * 104
o1 = torch.zeros(A, 104, device='cuda'); o2 = torch.zeros(ns, 104, device='cuda')
for name, f in (('NN  [A, ns] x [ns, 104]  sga_gemm', lambda: ops.gemm(m1, x1, False, False, A, 104, ns, out=o1, accumulate=True)),
                ('NN  torch.matmul', lambda: torch.matmul(m1, x1)),
                ('TN  [A, ns]^T x [A, 104] sga_gemm', lambda: ops.gemm(m1, x2, True, False, ns, 104, A, out=o2, accumulate=True)),
                ('TN  torch.matmul', lambda: torch.matmul(m1.t(), x2))):
    ms = t(f)
    print(f'{name:36s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  ({m1.numel() * 4 / ms / 1e9:.2f} TB/s of stash reads)')
