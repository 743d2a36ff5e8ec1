"""Inference path throughput: encoder forward (no grad) + per-pair similarity ranking / Hits@K on val-style batches (every
common object an anchor) of BASELINE.json configs[1] (default) or configs[2] shape.  Prints one JSON line with the HIP-event
time of the similarity kernel and its roofline (the per-pair E E^T blocks read each pair's table once: HBM/L2-bound).
  python tools/bench_eval.py [pairs] [objects] [--f16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sgaligner_amd import ops
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps

args = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args[0]) if len(args) > 0 else 512
N = int(args[1]) if len(args) > 1 else 64
ops.SIMRANK_F16 = '--f16' in sys.argv
mods = ['point', 'gat', 'rel']
steps = AlignerSteps(mods, device='cuda', seed=42)
steps.model.eval()
dd = make_batch_fast(B, N, 512, seed=5, device='cuda', anchors='val')


def once():
    out = steps.test_step(0, dd)
    return out, steps.eval_step(0, dd, out)


out, m = once(); once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out, m = once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(5):
    out = steps.test_step(0, dd)
torch.cuda.synchronize(); enc = (time.perf_counter() - t1) / 5
# the similarity + ranking kernels alone, HIP events on the launch stream
emb = out['joint'].detach()
counts = np.asarray(dd['tot_obj_count'])
e1i, e2i = np.asarray(dd['e1i']), np.asarray(dd['e2i'])
ev = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.simrank(emb, counts, e1i, e2i, 1)
    b.record()
    ev.append((a, b))
torch.cuda.synchronize()
k_ms = float(np.median([a.elapsed_time(b) for a, b in ev[2:]]))
T, D = emb.shape
alg_bytes = T * D * 4 + len(e1i) * 12                       # every pair's table once + rank/top-1 outputs
rows_with_queries = len(e1i)
flops = 2.0 * sum(int(c) for c in counts) * 0 + 2.0 * D * sum((min(int(na), int(n)) + 15) // 16 * 16 * int(n) for na, n in zip(dd['e1i_count'], counts))
print(json.dumps({
    'metric': 'inference pairs/s (encoder fwd + per-pair similarity ranking + Hits@K/MRR/SGAR)', 'value': round(B / dt, 1), 'unit': 'pairs/s',
    'ms_per_batch': round(dt * 1e3, 3), 'encoder_forward_ms': round(enc * 1e3, 3), 'ranking_and_metrics_ms': round((dt - enc) * 1e3, 3),
    'config': {'workload': f'{B} val-style pairs x {N} objects x 512 pts, {"+".join(mods)}, joint width {D}', 'similarity': 'f16-input MFMA' if ops.SIMRANK_F16 else 'exact fp32 MFMA'},
    'roofline': {'bound': 'hbm', 'achieved': round(alg_bytes / (k_ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                 'frac': round(alg_bytes / (k_ms * 1e-3) / 8e12, 4), 'traffic': None,
                 'kernel': 'sga_simrank (row norms + query scatter + simrank_mfma_kernel)', 'avg_launch_ms': round(k_ms, 4),
                 'algorithmic_bytes_per_launch': alg_bytes, 'mfma_tflops': round(flops / (k_ms * 1e-3) / 1e12, 2)},
    'hits_at_1': m[1]['correct'] / max(1, m[1]['total'])}))
