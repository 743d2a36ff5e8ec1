"""Cost of the BatchNorm-statistics epilogue of the NT GEMM (fp64 atomics per column and workgroup) against the plain GEMM and against a
separate statistics pass, at the PCT encoder's shapes.  python tools/dbg/bnstats_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import _lib, ops
from sgaligner_amd.ops import _p, _stream
L = _lib.lib(); st = _stream()
for r, k, n in ((164000, 128, 128), (164000, 128, 256), (164000, 512, 1024), (164000, 3, 128)):
    x = torch.randn(r, k, device='cuda'); w = torch.randn(n, k, device='cuda') * 0.1
    y = torch.empty(r, n, device='cuda'); sums = torch.empty(2 * n, device='cuda', dtype=torch.float64)
    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    t_plain = t(lambda: ops.gemm(x, w, False, True, r, n, k, out=y))
    rc = [0]
    def withstats():
        rc[0] = L.sga_gemm_bnstats(r, n, k, _p(x), x.stride(0), _p(w), w.stride(0), _p(y), y.stride(0), None, _p(sums), st)
    t_stats = t(withstats)
    print(f'[{r} x {k}] x [{k} x {n}]: plain {t_plain:.1f} us, with statistics epilogue {t_stats:.1f} us (rc {rc[0]})')
