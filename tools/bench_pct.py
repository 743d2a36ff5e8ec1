"""NaivePCT inference throughput (eval mode) and the attention kernel's MFMA rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd import _lib
from sgaligner_amd.aligner.networks.pct import NaivePCT
from sgaligner_amd.ops import _p, _stream

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = 512
torch.manual_seed(0)
m = NaivePCT().cuda().eval()
x = torch.randn(T, 3, N, device='cuda')
with torch.no_grad():                    # inference: the chunked eval path (with gradients enabled the module keeps activations for backward)
    for _ in range(2):
        y = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = m(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
# per object: convs 2*N*(3*128 + 128*128 + 4*(128*32 + 2*128*128) + 512*1024) + attention 4 * 2*N*N*(32 + 128) + head
conv = 2.0 * N * (3 * 128 + 128 * 128 + 4 * (128 * 32 + 2 * 128 * 128) + 512 * 1024) + 2.0 * (1024 * 512 + 512 * 256)
attn = 4 * 2.0 * N * N * (32 + 128)
print(f'NaivePCT eval forward: T={T} objects x {N} pts: {dt*1e3:.2f} ms = {T/dt:.0f} objects/s, '
      f'{(conv + attn) * T / dt / 1e12:.1f} TFLOP/s algorithmic ({(conv + attn) / 1e9:.2f} GFLOP/object, attention {attn / 1e9:.2f})')
Ta = min(T, 4096)                          # the attention kernel alone: at most 4096 objects (2 x 17 GB of operands at 65 536)
q = torch.randn(Ta * N, 32, device='cuda'); v = torch.randn(Ta * N, 128, device='cuda')
st = torch.empty(2 * Ta * N, device='cuda'); xs = torch.empty(Ta * N, 128, device='cuda')
L = _lib.lib()
L.sga_pct_attention(_p(q), 32, _p(v), 128, Ta, N, _p(st), _p(xs), 128, _stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    L.sga_pct_attention(_p(q), 32, _p(v), 128, Ta, N, _p(st), _p(xs), 128, _stream())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
alg = 2.0 * N * N * (32 + 128) * Ta
print(f'sga_pct_attention: {ms:.3f} ms per SA layer, {alg / ms / 1e9:.1f} TFLOP/s algorithmic '
      f'({alg * (32 * 2 + 128) / (32 + 128) / ms / 1e9:.1f} executed: the energy tiles are computed in both passes) of 157.3 peak')
# CPU baseline: the torch oracle (pinned to the reference module) on a bounded sample, 32 threads
from oracle import pct_oracle
torch.set_num_threads(32)
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
xc = x[:64].cpu()
pct_oracle.naive_pct_forward(xc[:8], sd)
t0 = time.perf_counter()
for _ in range(3):
    yc = pct_oracle.naive_pct_forward(xc, sd)
cpu = (time.perf_counter() - t0) / 3
err = (yc - y[:64].cpu()).abs().max().item()
print(f'CPU oracle (32 threads, 64 objects): {cpu*1e3:.1f} ms = {64/cpu:.0f} objects/s; max |GPU - oracle| on those objects {err:.2e}')
# training step of the encoder alone (forward + backward, batch-statistic BatchNorm), reference-like batch: 1024 objects
Tt = 1024
mt = NaivePCT().cuda().train()
xt = torch.randn(Tt, 3, N, device='cuda')
cot = torch.randn(Tt, 256, device='cuda')
for _ in range(2):
    mt.zero_grad(set_to_none=True)
    (mt(xt) * cot).sum().backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    mt.zero_grad(set_to_none=True)
    (mt(xt) * cot).sum().backward()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f'NaivePCT train step (fwd+bwd, T={Tt} x {N} pts): {dt*1e3:.1f} ms = {Tt/dt:.0f} objects/s '
      f'({3 * (conv + attn) * Tt / dt / 1e12:.1f} TFLOP/s at 3x the forward FLOPs), peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
