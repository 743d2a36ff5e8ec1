"""Per-phase cycle breakdown of the split-fp16 sweeps (variant built with -DSH_DBG_TIMING: s_memtime stamps at the phase boundaries of every
tile, summed over all waves).  SGA_LIB_PATH=variants/libsga_<tag>.so python tools/dbg/sweeph_phases.py [pairs=512] [objects=64]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd import _lib, ops
from sgaligner_amd.synthetic import make_batch_fast
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ops.set_mfma_mode('f16x2')
dd = make_batch_fast(B, N, 4, seed=3, device='cuda')
T = int(dd['tot_obj_pts'].shape[0])
g = torch.Generator(device='cuda').manual_seed(0)
tabs = [torch.randn(T, 100, device='cuda', generator=g).requires_grad_(True) for _ in range(3)]
w = torch.ones(3, 1, device='cuda', requires_grad=True)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
for rep in range(2):
    sums, s = ops.fused_contrastive_terms(tabs, w, dd)
    sums.sum().backward()
    torch.cuda.synchronize()
    L.sga_dbg_sweeph(buf)
names = ['barrier wait', 'DMA issue', 'S phase (+ edge zeroing)', 'joint epilogue', 'table epilogues + gradient MFMAs', 'loop tail']
for base, tag in ((0, 'sums'), (8, 'grad')):
    tot = sum(buf[base + i] for i in range(8))
    print(tag, 'total wave-cycles', tot, ' '.join(f'| {names[i]} {100.0 * buf[base + i] / max(1, tot):.1f}%' for i in range(6)))
