"""End-to-end check of the one-process-per-GPU path: WORLD_SIZE ranks each take a contiguous block of pairs of ONE
global batch, run AlignerSteps.forward_backward (tables all-gathered, anchor-sharded global loss, flat gradient
all-reduce) and rank 0 compares loss + parameter gradients against the single-process result on the full batch.
  SGA_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dist_check.py
(gloo lets several ranks share one GPU; on a multi-GPU node drop SGA_DIST_BACKEND to use RCCL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from sgaligner_amd import dist as sdist
from sgaligner_amd.synthetic import make_batch, to_device
from sgaligner_amd.trainer import AlignerSteps

rank, world, local = sdist.init_from_env()
dev = torch.device('cuda', torch.cuda.current_device())
mods = ['point', 'gat', 'rel']
full = to_device(make_batch(6, 14, 48, seed=21, ragged=True), dev)
lo, hi = sdist.shard_range(6, rank, world)
mine = sdist.shard_data_dict(full, lo, hi)
steps = AlignerSteps(mods, device=dev, seed=42)
_, loss = steps.forward_backward(mine)
torch.cuda.synchronize()
grads = {n: p.grad.clone() for n, p in steps.model.named_parameters() if p.grad is not None}
lv = [p.grad.clone() for p in steps.multi_loss_layer_ial.parameters()] + [p.grad.clone() for p in steps.multi_loss_layer_icl.parameters()]
ok = True
if rank == 0:
    ref = AlignerSteps(mods, device=dev, seed=42)
    grp, world_saved = dist.group.WORLD, world
    # single-process reference: bypass the distributed branch
    ref.zero_grad()
    out = ref.model(full)
    l = ref.loss_func(out, full)
    l['loss'].backward()
    torch.cuda.synchronize()
    e = abs(l['loss'].item() - loss['loss'].item()) / max(1.0, abs(l['loss'].item()))
    print(f'loss single {l["loss"].item():.6f} vs {world} ranks {loss["loss"].item():.6f}  rel err {e:.2e}')
    ok &= e < 1e-4
    worst = 0.0
    for n, p in ref.model.named_parameters():
        if p.grad is not None:
            err = (p.grad - grads[n]).abs().max().item() / max(1.0, p.grad.abs().max().item())
            worst = max(worst, err)
    for a, b in zip([p.grad for p in ref.multi_loss_layer_ial.parameters()] + [p.grad for p in ref.multi_loss_layer_icl.parameters()], lv):
        worst = max(worst, (a - b).abs().max().item() / max(1.0, a.abs().max().item()))
    print(f'max rel grad err {worst:.2e}')
    ok &= worst < 1e-3
    print('DIST_CHECK', 'PASS' if ok else 'FAIL')
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
