"""Micro-benchmark of the PointNet forward kernel (HIP events on the launch stream): exact fp32 (v_mfma_f32_32x32x2_f32) and the default
three exact bf16 planes, with the latter's difference from the former (values, arg-max points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [torch.randn(64, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1,
     torch.randn(128, 64, device='cuda') * 0.1, torch.randn(128, device='cuda') * 0.1,
     torch.randn(256, 128, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1]
ref = None
for mode in ('f32', 'bf16x6'):
    old = ops.set_mfma_mode(mode)
    for am in (False, True):
        for _ in range(2):
            y, a = ops.pointnet_forward(x, *w, want_argmax=am)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 5
        for _ in range(n):
            y, a = ops.pointnet_forward(x, *w, want_argmax=am)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        fl = 82304.0 * T * P
        extra = ''
        if mode == 'f32' and am:
            ref = (y.clone(), a.clone())
        if mode != 'f32' and am:
            err = (y - ref[0]).abs().max().item()
            rel = err / ref[0].abs().max().item()
            same = (a == ref[1]).float().mean().item()
            extra = f'  max|y - y_fp32| {err:.3e} (rel {rel:.2e}), same arg-max point {same * 100:.5f} % ({int((a != ref[1]).sum())} of {a.numel()} differ, {int(((a != ref[1]) & (ref[0] > 0)).sum())} of them with y > 0)'
        print(f'pointnet_fwd mode={mode} argmax={am} T={T} P={P}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s algorithmic '
              f'({fl/ms/1e9/157.3*100:.1f}% of the fp32 MFMA peak){extra}')
    ops.set_mfma_mode(old)
