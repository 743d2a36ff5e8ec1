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

# BASELINE.json configs[1] is P+S+R; SGA_BENCH_MODULES (e.g. point,gat,rel,attr) is for side experiments only
MODULES = os.environ.get('SGA_BENCH_MODULES', 'point,gat,rel').split(',')
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


def pmc_traffic_bytes(kernel_tag):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes of THIS workload
    (profiles/r01_o_pmc_traffic.csv, written by tools/pmc_traffic.sh: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of
    bench.py, KiB per dispatch; gfx950 correction: FETCH_SIZE counts wide coalesced reads at half their size -> x2,
    MI355X_MICROARCH.md, HBM)."""
    path = os.path.join(ROOT, 'profiles', 'r01_o_pmc_traffic.csv')
    try:
        f = w = None
        for line in open(path):
            if line.startswith('#') or not line.startswith(kernel_tag + ','):
                continue
            _, counter, value, _ = line.strip().rsplit(',', 3)
            if counter == 'FETCH_SIZE':
                f = float(value)
            elif counter == 'WRITE_SIZE':
                w = float(value)
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
        # The two kernels that carry the step, each timed with HIP events on its launch stream; `roofline` is the
        # one with the larger per-step time, the other is reported under `roofline_other`.  Both are bound by the
        # exact-fp32 MFMA rate (157.3 TFLOP/s dense; on gfx950 fp32 MFMA and fp32 VALU share the SIMD's FMA datapath --
        # tools/micro/mfma_valu_overlap.hip -- so epilogue VALU work adds to, rather than hides under, the MFMA time).
        roofs = []
        evs = events.get('pointnet_fwd_kernel', [])
        if evs:
            durs = [a.elapsed_time(b) for a, b, _ in evs]
            T, P, C1, C2, C3 = evs[0][2]
            # algorithmic FLOPs: 2 * T * P * (3*C1 + C1*C2 + C2*C3) for the three per-point layers
            alg = 2.0 * T * P * (3 * C1 + C1 * C2 + C2 * C3)
            avg_ms = float(np.mean(durs))
            ach = alg / (avg_ms * 1e-3) / 1e12
            roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': round(ach / PEAK_F32_TFLOPS, 4),
                          'traffic': pmc_traffic_bytes('pointnet_fwd_kernel') if world == 1 else None,
                          'kernel': 'pointnet_fwd_kernel<256,true> (object encoder: 3 per-point layers + max-pool, one wave per object)',
                          'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4),
                          'algorithmic_flops_per_launch': alg})
        evs = events.get('loss_multi_grad', [])
        if evs:
            durs = [a.elapsed_time(b) for a, b, _ in evs]
            ns, A, J1, J2, M = evs[0][2]          # ns = anchors in this rank's shard (== A on one GPU)
            # Algorithmic FLOPs: backward of the four anchors x negatives products of all M+1 tables
            # = 2 x [ sum_tab 2*D_tab * 2A(J1+J2) ]  with sum_tab D_tab = 100*M + 100*M.
            d_sum = 100 * M + 100 * M
            alg = 2.0 * (2.0 * d_sum * 2.0 * ns * (J1 + J2))
            # executed: two sweeps x M tables x (S with K = 100 + gradient GEMM with 112 columns); the joint table is derived
            executed = 2.0 * (2.0 * ns * (J1 + J2)) * 2.0 * M * (100 + 112)
            avg_ms = float(np.mean(durs))
            ach = alg / (avg_ms * 1e-3) / 1e12
            kname = f'sweep16_kernel<{M},true>'
            roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': round(ach / PEAK_F32_TFLOPS, 4), 'traffic': pmc_traffic_bytes(f'sweep16_kernel<{M},true>') if world == 1 else None,
                          'kernel': f'{kname} (loss: negatives backward, all {M}+1 tables)',
                          'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4),
                          'algorithmic_flops_per_launch': alg, 'executed_flops_per_launch': executed,
                          'executed_tflops': round(executed / (avg_ms * 1e-3) / 1e12, 2)})
        roofs.sort(key=lambda r: -r['avg_launch_ms'])
        roof = roofs[0] if roofs else None
        line = {
            'metric': 'subscan-pairs/sec (fwd+bwd)', 'value': round(total_pairs * args.steps / elapsed, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE.json configs[1]: {PAIRS_PER_GPU} synthetic subscan pairs/GPU x {N_OBJ} objects x '
                                   f'{N_PTS} pts, modules {"+".join(MODULES)} (P+S+R), batch-global ICL/IAL loss over '
                                   f'{total_pairs} pairs', 'global_pairs': total_pairs, 'objects_per_scene': N_OBJ,
                       'points_per_object': N_PTS, 'modules': MODULES, 'parallelism': f'dp{world}', 'loss': loss_val},
            'roofline': roof,
            'roofline_other': roofs[1:],
        }
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
