"""Micro-benchmark of the PointNet forward kernel (HIP events on the launch stream): exact fp32 and the opt-in split modes
(bf16x3; f16x2 = fp16 hi + lo, fp32-faithful), with their error against the former."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import _lib, ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [torch.randn(64, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1,
     torch.randn(128, 64, device='cuda') * 0.1, torch.randn(128, device='cuda') * 0.1,
     torch.randn(256, 128, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1]
ref = None
EPS = [float(e) for e in os.environ.get('PN_TIE_EPS', '').split(',') if e] or [None]
NAMES = {0: 'f32', 1: 'bf16x3', 3: 'f16x2', 4: 'f16x2p'}
for mode, eps in [(0, None), (1, None), (4, None)] + [(3, e) for e in EPS]:
    ops.POINTNET_TIE_EPS = -1.0 if eps is None else eps
    ops.set_mfma_mode(NAMES[mode])
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
        if mode == 0 and am:
            ref = (y.clone(), a.clone())
        if mode in (1, 3, 4) and am:
            err = (y - ref[0]).abs().max().item()
            rel = err / ref[0].abs().max().item()
            same = (a == ref[1]).float().mean().item()
            extra = f'  max|y - y_fp32| {err:.3e} (rel {rel:.2e}), same arg-max point {same * 100:.5f} % ({int((a != ref[1]).sum())} of {a.numel()} differ, {int(((a != ref[1]) & (ref[0] > 0)).sum())} of them with y > 0)'
            if mode == 3:
                extra += f'; tie eps {ops.POINTNET_TIE_EPS if ops.POINTNET_TIE_EPS >= 0 else 2.0 ** -17:.3g}: {int(ops.POINTNET_LAST_REDO[0]) / T * 100:.2f} % of the objects re-run in fp32'
        print(f'pointnet_fwd mode={ {0: "fp32", 1: "bf16x3", 3: "f16x2", 4: "f16x2p"}[mode]} argmax={am} T={T} P={P}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s algorithmic '
              f'({fl/ms/1e9/157.3*100:.1f}% of the fp32 MFMA peak){extra}')
ops.set_mfma_mode('f32')
