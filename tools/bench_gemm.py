"""Exact-fp32 GEMM (sga_gemm_ex: C = act(A W^T + b)) at the per-point layer shapes of the PCT encoder and the linears of the path.
  python tools/bench_gemm.py [rows=2097152]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgaligner_amd.aligner.networks.pct import _gemm_ex

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
shapes = [(128, 128), (128, 32), (512, 1024), (64, 128), (256, 100), (1024, 512), (512, 256)]
for K, N in shapes:
    m = M if K * N < 512 * 1024 else M // 4
    a = torch.randn(m, K, device='cuda')
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    b = torch.randn(N, device='cuda')
    out = torch.empty(m, N, device='cuda')
    for _ in range(2):
        _gemm_ex(a, w, b, act=1, out=out)
    ev = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _gemm_ex(a, w, b, act=1, out=out); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = float(np.median([x.elapsed_time(y) for x, y in ev]))
    fl = 2.0 * m * K * N
    by = 4.0 * (m * K + m * N + K * N)
    ref = torch.relu(a[:4096].double() @ w.double().T + b.double())
    err = float((out[:4096].double() - ref).abs().max())
    print(f'M={m:8d} K={K:4d} N={N:4d}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s ({fl / ms / 1e9 / 157.3:.2f} of fp32 MFMA peak)  {by / ms / 1e6:6.0f} GB/s  max err {err:.1e}')
