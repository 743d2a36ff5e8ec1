"""Where a wave of sweep3_kernel spends its cycles (build with `python -m sgaligner_amd._build -DS3_DBG_TIMING` first; s_memtime stamps
cost ~10 % themselves).  python tools/dbg/sweep3_phases.py [pairs=512] [objects=64]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import _lib, ops
from sgaligner_amd.synthetic import make_batch_fast
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
MT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ops.set_mfma_mode('bf16x6')
dd = make_batch_fast(B, N, 4, seed=3, device='cuda')
T = int(dd['tot_obj_pts'].shape[0])
g = torch.Generator(device='cuda').manual_seed(0)
tabs = [torch.randn(T, 100, device='cuda', generator=g).requires_grad_(True) for _ in range(MT)]
w = torch.ones(MT, 1, device='cuda', requires_grad=True)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
sums, s = ops.fused_contrastive_terms(tabs, w, dd)
sums.sum().backward()
torch.cuda.synchronize()
raw.sga_dbg_sweep3(buf)
sums, s = ops.fused_contrastive_terms(tabs, w, dd)
sums.sum().backward()
torch.cuda.synchronize()
raw.sga_dbg_sweep3(buf)
names = ['barrier wait', 'DMA issue', 'S phase (+ edge zeroing)', 'joint coefficient', 'coefficients + gradient GEMM', 'loop overhead', '-', '-']
for k, tag in ((0, 'sums'), (8, 'grad')):
    tot = sum(buf[k + i] for i in range(8))
    print(tag, 'total wave-cycles', tot)
    for i in range(6):
        print(f'   {names[i]:32s} {buf[k + i] / max(tot, 1) * 100:6.2f} %')
