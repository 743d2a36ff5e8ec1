"""Time the negatives' backward sweep (sga_loss_multi_grad) alone on configs[1]-shaped index sets (or configs[2] shard: pass pairs, objects).
  python tools/bench_sweep.py [pairs=512] [objects=64] [reps=5] [tables=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch_fast
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
MT = int(sys.argv[4]) if len(sys.argv) > 4 else 3          # modality tables (4 = point + gat + rel + attr: the paired-wave sweep16x2 kernels)
dd = make_batch_fast(B, N, 4, seed=3, device='cuda')
T = int(dd['tot_obj_pts'].shape[0])
g = torch.Generator(device='cuda').manual_seed(0)
tabs = [torch.randn(T, 100, device='cuda', generator=g).requires_grad_(True) for _ in range(MT)]
w = torch.ones(MT, 1, device='cuda', requires_grad=True)
ops.KERNEL_EVENTS = {}
for _ in range(reps + 2):
    sums, s = ops.fused_contrastive_terms(tabs, w, dd)
    sums.sum().backward()
torch.cuda.synchronize()
sfx = {'bf16x6': '_bf16x6', 'f16': '_bf16x6'}.get(ops.get_mfma_mode(), '')
ev = ops.KERNEL_EVENTS['loss_multi_grad' + sfx][2:]
ms = [a.elapsed_time(b) for a, b, _ in ev]
ns, A, J1, J2, M = ev[0][2]
alg = 2.0 * (2.0 * 200 * MT * 2.0 * ns * (J1 + J2))
evf = ops.KERNEL_EVENTS.get('loss_multi_sums' + sfx, [])[2:]
msf = [a.elapsed_time(b) for a, b, _ in evf] or [float('nan')]
print(f'sums {np.median(msf):.3f} ms | sweep grad: median {np.median(ms):.3f} ms (min {min(ms):.3f})  A={A} J={J1 + J2}  algorithmic {alg / np.median(ms) / 1e9:.1f} TFLOP/s = {alg / np.median(ms) / 1e9 / 157.3:.3f} of fp32 MFMA peak; grad checksum {float(tabs[0].grad.abs().sum()):.6e}')
# accuracy of the opt-in mode against the exact-fp32 sweeps on the same inputs (run with SGA_BENCH_SWEEP_COMPARE=1)
if os.environ.get('SGA_BENCH_SWEEP_COMPARE'):
    res = {}
    other = 'bf16x6'
    for mode in ('f32', other):
        ops.set_mfma_mode(mode)
        for t in tabs:
            t.grad = None
        w.grad = None
        sums, s = ops.fused_contrastive_terms(tabs, w, dd)
        sums.sum().backward()
        torch.cuda.synchronize()
        res[mode] = (sums.detach().clone(), [t.grad.clone() for t in tabs], w.grad.clone())
    ops.set_mfma_mode('f32')
    a, b = res['f32'], res[other]
    print('loss terms rel err', ((a[0] - b[0]).abs() / a[0].abs().clamp_min(1e-30)).max().item())
    for k in range(3):
        print(f'dE[{k}] max abs err {(a[1][k] - b[1][k]).abs().max().item():.3e} / max |dE| {a[1][k].abs().max().item():.3e}')
    print('dw max abs err', (a[2] - b[2]).abs().max().item(), '/', a[2].abs().max().item())
