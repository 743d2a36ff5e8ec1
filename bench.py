#!/usr/bin/env python3
"""Headline benchmark: subscan-pairs/sec, fwd + bwd (encoder forward, OverallLoss forward, backward to all
parameter gradients; optimiser step and data loading excluded; inputs resident in HBM) -- BASELINE.json
metric, SURVEY.md 8(d).

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

Workload (config.workload): BASELINE.json configs[1] per GPU -- 512 synthetic subscan pairs x 64 objects x
512 points, modules point+gat+rel (P+S+R), batch-global loss.  For N > 1 every rank holds 512 pairs (weak
scaling) and the loss is the global one over all 512*N pairs (tables all-gathered over RCCL).
Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MODULES = ['point', 'gat', 'rel']
PAIRS_PER_GPU, N_OBJ, N_PTS = 512, 64, 512
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA (= fp32 vector) dense peak
PEAK_HBM_GBS = 8000.0


def cpu_baseline(seconds_budget=20.0):
    """The oracle (CPU restatement pinned to the reference by tests/golden) on this box's host cores:
    fwd + loss + bwd at the reference's native batch size b=2 (configs/scan3r/scan3r_ground_truth.yaml:27)
    with the same (objects, points, modules) as the GPU workload."""
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch
    # Thread count: on the 2 x 64-core EPYC 9575F GPU host a sweep over {4,8,16,32,64,128} threads (tools/cpu_threads.py)
    # peaks at 32 (5.9 pairs/s); all 256 hardware threads are 20x SLOWER (0.26 pairs/s) on these small per-graph ops.
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    b = 2
    dd = make_batch(b, N_OBJ, N_PTS, seed=43, device='cpu')
    params = O.init_params(MODULES, seed=42)
    O.train_step(params, dd, MODULES)                      # warm-up
    times = []
    t_end = time.time() + seconds_budget
    while len(times) < 5 or (time.time() < t_end and len(times) < 40):
        t0 = time.time()
        O.train_step(params, dd, MODULES)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {'value': b / med, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle fwd+loss+bwd, b={b} pairs x {N_OBJ} obj x {N_PTS} pts, {"+".join(MODULES)}, '
                      f'{len(times)} iterations, median {med*1e3:.1f} ms, torch {torch.__version__} CPU, {cores} threads '
                      f'(best of a thread-count sweep; host has {os.cpu_count()} hardware threads)'}


def pmc_traffic_bytes():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload
    (profiles/r01_e_pmc_sweep_multi.csv: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of bench.py, KiB per dispatch;
    gfx950 correction: FETCH_SIZE counts wide coalesced reads at half their size -> x2, MI355X_MICROARCH.md, HBM)."""
    path = os.path.join(ROOT, 'profiles', 'r01_e_pmc_sweep_multi.csv')
    try:
        f = w = None
        for line in open(path):
            if line.startswith('#') or ', true>' not in line:
                continue
            parts = line.strip().split(',')
            name_end = len(parts) - 7
            counter, value = parts[name_end], float(parts[name_end + 1])
            if counter == 'FETCH_SIZE':
                f = value
            elif counter == 'WRITE_SIZE':
                w = value
        if f is None or w is None:
            return None
        return int((2.0 * f + w) * 1024)
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from sgaligner_amd import dist as sdist
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch
    from sgaligner_amd.trainer import AlignerSteps

    rank, world, local = sdist.init_from_env()
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the product path has no CPU fallback)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world != args.gpus and rank == 0:
        print(f'[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)

    steps = AlignerSteps(MODULES, device=dev, seed=42)
    dd = make_batch(PAIRS_PER_GPU, N_OBJ, N_PTS, seed=43 + rank, device=dev, gen_device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        steps.forward_backward(dd)
    barrier()
    ops.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, loss_dict = steps.forward_backward(dd)
    barrier()
    elapsed = time.perf_counter() - t0
    events = ops.KERNEL_EVENTS
    ops.KERNEL_EVENTS = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss_dict['loss'].item())

    if rank == 0:
        total_pairs = PAIRS_PER_GPU * world
        ms = elapsed / args.steps * 1e3
        # dominant kernel: the fused negatives-gradient sweep (sweep_multi_kernel<M,...,true>), one launch per step.
        # Algorithmic FLOPs: backward of the four anchors x negatives products of all M+1 tables
        # = 2 x [ sum_tab 2*D_tab * 2A(J1+J2) ]  with sum_tab D_tab = 100*M + 100*M.
        roof = None
        key = f'sweep_multi_kernel<{len(MODULES)},grad>'
        evs = events.get(key, [])
        if evs:
            durs = [a.elapsed_time(b) for a, b, _ in evs]
            ns, A, J1, J2, M = evs[0][2]          # ns = anchors in this rank's shard (== A on one GPU)
            d_sum = 100 * M + 100 * M
            fwd_flops = 2.0 * d_sum * 2.0 * ns * (J1 + J2)
            alg = 2.0 * fwd_flops
            executed = 2.0 * (2.0 * ns * (J1 + J2)) * 2.0 * M * (104 + 128)      # two sweeps x M x (S K=104 + grad 128 cols)
            avg_ms = float(np.mean(durs))
            ach = alg / (avg_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_TFLOPS, 4), 'traffic': pmc_traffic_bytes() if world == 1 else None,
                    'kernel': f'sweep_multi_kernel<{M},0,{M},true> (loss: negatives backward, all {M}+1 tables)',
                    'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4),
                    'algorithmic_flops_per_launch': alg, 'executed_flops_per_launch': executed,
                    'executed_tflops': round(executed / (avg_ms * 1e-3) / 1e12, 2)}
        line = {
            'metric': 'subscan-pairs/sec (fwd+bwd)', 'value': round(total_pairs * args.steps / elapsed, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE.json configs[1]: {PAIRS_PER_GPU} synthetic subscan pairs/GPU x {N_OBJ} objects x '
                                   f'{N_PTS} pts, modules {"+".join(MODULES)} (P+S+R), batch-global ICL/IAL loss over '
                                   f'{total_pairs} pairs', 'global_pairs': total_pairs, 'objects_per_scene': N_OBJ,
                       'points_per_object': N_PTS, 'modules': MODULES, 'parallelism': f'dp{world}', 'loss': loss_val},
            'roofline': roof,
        }
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
