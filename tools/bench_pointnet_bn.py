"""PointNet training forward, with and without the fused BatchNorm statistics (sga_pointnet_fwd_bn), in the exact-fp32 kernel ('f32') and on
three exact bf16 planes ('bf16x6', the default), HIP events on the launch stream; the default's outputs against the fp32 kernel's.
  python tools/bench_pointnet_bn.py [T=131072] [P=512]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
x = torch.randn(T, P, 3, device='cuda')
w = [torch.randn(64, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1,
     torch.randn(128, 64, device='cuda') * 0.1, torch.randn(128, device='cuda') * 0.1,
     torch.randn(256, 128, device='cuda') * 0.1, torch.randn(256, device='cuda') * 0.1]
sums = torch.empty(265 + 512, device='cuda', dtype=torch.float64)
ref = {}
for mode in ('f32', 'bf16x6'):
    ops.set_mfma_mode(mode)
    for tag, bn in (('plain', None), ('with BN sums', sums)):
        for _ in range(2):
            y, am = ops.pointnet_forward(x, *w, want_argmax=True, bn_sums=bn)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 5
        for _ in range(n):
            y, am = ops.pointnet_forward(x, *w, want_argmax=True, bn_sums=bn)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        extra = ''
        if mode == 'f32':
            ref[tag] = (y.clone(), am.clone(), sums.clone())
        else:
            ry, ra, rs = ref[tag]
            extra = f'  max|y - y_f32| {(y - ry).abs().max().item():.2e} (max |y| {ry.abs().max().item():.2e}); arg-max differs at {int((am != ra).sum())} of {am.numel()}'
            if bn is not None:
                extra += f'; BN sums rel diff {((sums - rs).abs() / rs.abs().clamp_min(1e-30)).max().item():.2e}'
        print(f'{mode:7s} T={T} P={P} {tag:14s} {ms:8.3f} ms   {82304.0 * T * P / ms / 1e9:.1f} TFLOP/s (forward FLOPs only){extra}')
