"""PCIe-inclusive step rate at BASELINE configs[1]: the boundary (`train_step(data_dict)`) receives device tensors, the batch
gets there from pinned host memory.  Three loops over the same 3 host batches:
  resident   the batch is already in HBM (what bench.py's `value` times)
  sync       DeviceBatch(host batch) then the step, on one stream (the reference's loop: to_cuda inside the iteration)
  prefetch   datasets.DevicePrefetcher: batch i+1 uploaded on a second HIP stream under the step of batch i
  python tools/bench_h2d.py [pairs=512] [objects=64] [steps=12]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgaligner_amd.datasets import DeviceBatch, DevicePrefetcher
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
steps = AlignerSteps(['point', 'gat', 'rel'], device='cuda', seed=42)
host = []
for i in range(3):
    dd = make_batch_fast(B, N, 512, seed=43 + i, device='cuda')
    host.append({k: (v.cpu().pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in dd.items()})
    del dd
nbytes = sum(v.numel() * v.element_size() for v in host[0].values() if isinstance(v, torch.Tensor))


def run(batches):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for dd in batches:
        steps.forward_backward(dd)
        n += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = DeviceBatch(host[0])
for _ in range(3):
    steps.forward_backward(res)
t_res = run([res] * K)
t_sync = run(DeviceBatch(host[i % 3]) for i in range(K))
t_pre = run(DevicePrefetcher([host[i % 3] for i in range(K)], 'cuda'))
# copy alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(4):
    d = DeviceBatch(host[i % 3])
torch.cuda.synchronize(); t_copy = (time.perf_counter() - t0) / 4 * 1e3
print(json.dumps({'workload': f'{B} pairs x {N} objects x 512 pts, point+gat+rel', 'host_batch_mb': round(nbytes / 1e6, 1),
                  'upload_ms': round(t_copy, 2), 'upload_gb_s': round(nbytes / t_copy / 1e6, 1),
                  'ms_per_step': {'resident': round(t_res, 2), 'sync_upload': round(t_sync, 2), 'prefetch_stream': round(t_pre, 2)},
                  'pairs_per_s': {'resident': round(B / t_res * 1e3, 1), 'sync_upload': round(B / t_sync * 1e3, 1),
                                  'prefetch_stream': round(B / t_pre * 1e3, 1)}}))
