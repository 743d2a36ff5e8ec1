"""The fp16-input tile core of csrc/wide16.hip, call by call, at the configs[4] loss shape (A anchors, J1 = J2 negatives, Dp columns):
  python tools/bench_wide16.py [A=4864] [J=11520] [Dp=1024]
times sga_loss_neg_sums_f16, sga_loss_neg_grad_f16, the anchors x anchors forward / backward with the similarity blocks formed on the core,
and sga_loss_stash_grad_f16; prints TFLOP/s of the products each call executes against the 2.5 PFLOP/s dense fp16 peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as ct
import torch
from sgaligner_amd import _lib
A = int(sys.argv[1]) if len(sys.argv) > 1 else 4864
J = int(sys.argv[2]) if len(sys.argv) > 2 else 11520
Dp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
L = _lib.lib()
dev = torch.device('cuda:0')
R = 2 * A + 2 * J
torch.manual_seed(0)
z = torch.nn.functional.normalize(torch.randn(R, Dp, device=dev), dim=1).contiguous()
p = lambda t: ct.c_void_p(t.data_ptr())
st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
ldt = int(L.sga_wide16_ldt(A, J, J))
zh = torch.empty((R, Dp), device=dev, dtype=torch.float16)
zt = torch.empty((Dp, ldt), device=dev, dtype=torch.float16)
_lib.check(L.sga_wide16_prepare(p(z), Dp, A, J, J, p(zh), p(zt), st), 'prepare')
slots = 1 + L.sga_loss_slots()
sums = torch.empty((slots * 8,), device=dev, dtype=torch.float64)
gs = torch.full((8,), 1e-3, device=dev, dtype=torch.float64)
dz = torch.zeros((R, Dp), device=dev, dtype=torch.float32)
need = int(L.sga_loss_neg_grad_f16_bytes(A, J, J))
stash = torch.empty((need,), device=dev, dtype=torch.uint8)
unit = 2.0 * Dp * 2 * A * 2 * J          # one pass over the four (anchors x negatives) blocks of a table


def timed(name, fn, flops, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'{name:52s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s executed = {flops / ms / 1e9 / 2500:.3f} of the fp16 peak')


timed('sga_loss_neg_sums_f16 (S + exp sums)', lambda: _lib.check(L.sga_loss_neg_sums_f16(p(zh), Dp, A, J, J, 0.1, 1.0, p(sums), 0, A, st), 'sums'), unit)
timed('sga_loss_neg_grad_f16 (S -> C, C^T; two GEMMs)', lambda: _lib.check(L.sga_loss_neg_grad_f16(p(zh), p(zt), Dp, A, J, J, 0.1, 1.0, p(gs), p(dz), p(stash), need, 0, A, st), 'grad'), 3 * unit)
# anchors x anchors: one table + a joint of the same width (NT = 2)
nt = 2
zs = (ct.c_void_p * nt)(z.data_ptr(), z.data_ptr())
zhs = (ct.c_void_p * nt)(zh.data_ptr(), zh.data_ptr())
dps = (ct.c_int * nt)(Dp, Dp)
sm = torch.rand((nt, 8), device=dev, dtype=torch.float64) * 1e4 + 1e3
out = torch.empty((slots * (nt + 2),), device=dev, dtype=torch.float64)
ws = torch.empty((int(L.sga_loss_anchor_f16_ws_bytes(nt, A, A)),), device=dev, dtype=torch.uint8)
aa = 2.0 * Dp * A * A * 2 * nt
timed('sga_loss_anchor_fwd_f16 (tile core + epilogue kernel)', lambda: _lib.check(L.sga_loss_anchor_fwd_f16(zs, zhs, dps, nt, A, p(sm), 0.5, 0.1, 1.0, p(out), 0, A, p(ws), ws.numel(), st), 'afwd'), aa)
timed('sga_loss_anchor_fwd_f16 (one-kernel form)', lambda: _lib.check(L.sga_loss_anchor_fwd_f16(zs, zhs, dps, nt, A, p(sm), 0.5, 0.1, 1.0, p(out), 0, A, None, 0, st), 'afwd'), aa)
coef = torch.ones((nt + 2,), device=dev, dtype=torch.float32)
m1 = [torch.empty((A * A,), device=dev, dtype=torch.float32) for _ in range(nt)]
m1a = (ct.c_void_p * nt)(*[t.data_ptr() for t in m1])
gsa = torch.empty((slots, nt, 8), device=dev, dtype=torch.float64)
timed('sga_loss_anchor_bwd_f16 (tile core + epilogue kernel)', lambda: _lib.check(L.sga_loss_anchor_bwd_f16(zs, zhs, dps, nt, A, p(sm), 0.5, 0.1, 1.0, p(coef), m1a, p(gsa), 0, A, p(ws), ws.numel(), st), 'abwd'), aa)
timed('sga_loss_anchor_bwd_f16 (one-kernel form)', lambda: _lib.check(L.sga_loss_anchor_bwd_f16(zs, zhs, dps, nt, A, p(sm), 0.5, 0.1, 1.0, p(coef), m1a, p(gsa), 0, A, None, 0, st), 'abwd'), aa)
ws16 = torch.empty((int(L.sga_loss_stash_grad_f16_bytes(A, A)),), device=dev, dtype=torch.uint8)
timed('sga_loss_stash_grad_f16 (max + convert + two GEMMs)', lambda: _lib.check(L.sga_loss_stash_grad_f16(p(m1[0]), p(zt), Dp, A, J, J, p(dz), 0, A, p(ws16), ws16.numel(), st), 'sg16'), 2 * 2.0 * Dp * A * A)
timed('sga_loss_stash_grad (fp32 GEMMs)', lambda: _lib.check(L.sga_loss_stash_grad(p(m1[0]), p(z), A, Dp, p(dz), 0, A, st), 'sg32'), 2 * 2.0 * Dp * A * A)
