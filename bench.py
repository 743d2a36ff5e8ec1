#!/usr/bin/env python3
"""Headline benchmark: subscan-pairs/sec, fwd + bwd (encoder forward, OverallLoss forward, backward to all
parameter gradients; optimiser step and data loading excluded; inputs resident in HBM) -- BASELINE.json
metric, SURVEY.md 8(d) -- plus the metric's second half, node-match Hits@1 against the oracle on a fixed
val-style subsample.

  python bench.py [--gpus N --steps K --warmup W --config auto|c2|c3]      (N > 1: launched by torch.distributed.run)

Workloads (config.workload):
  c3 = BASELINE.json configs[2] = the configuration north_star quotes its target on ("4096-pair / 128-object / 512-pt synthetic
       batches at 1 GPU" + the 1/2/4/8-GPU series): 4096 pairs x 128 objects x 512 points IN TOTAL, sharded 4096/N pairs per
       GPU, batch-global loss via the table all-gather (strong scaling; N = 1 runs the whole batch on one GPU, 26 GiB).
       DEFAULT at every N: `roofline` is the kernel that carries this step (the loss-gradient sweep), `cpu_baseline` is taken
       at this shape (128 objects per scene, the reference's b = 2).
  c2 = BASELINE.json configs[1]: 512 synthetic subscan pairs PER GPU x 64 objects x 512 points, P+S+R, batch-global
       loss over all 512*N pairs (weak scaling).  `--config c2`; at N = 1 the default line carries it as `extra_c2`.
Rank 0 prints ONE compact JSON line (< 4 KB: the contract's keys, the dominant kernel's `roofline`, `cpu_baseline`, Hits@1) as the LAST line
of stdout; the full record (every roofline object, the CPU sweep, the other arithmetic modes and workloads) goes to bench_extras.json
beside this script (and gpurun_out/ when it exists) and to stderr, one `[bench extra]` line per key."""
import argparse
import hashlib
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

# BASELINE.json configs[1]/[2] are P+S+R; SGA_BENCH_MODULES (e.g. point,gat,rel,attr) is for side experiments only
MODULES = os.environ.get('SGA_BENCH_MODULES', 'point,gat,rel').split(',')
CONFIGS = {
    'c2': {'ref': 'BASELINE.json configs[1]', 'n_obj': 64, 'n_pts': 512, 'pairs_per_gpu': 512, 'scaling': 'weak'},
    'c3': {'ref': 'BASELINE.json configs[2] (north-star target)', 'n_obj': 128, 'n_pts': 512, 'global_pairs': 4096,
           'scaling': 'strong'},
    # configs[4] (stress shape; the config names no pair count; SURVEY 8(d): "B sized to memory (e.g. 64/GPU)": 64 pairs per GPU = 32 768 objects, 67 M points): 1024-d embeddings,
    # loss + ranking GEMMs on fp16-input MFMA (ops.set_mfma_mode('f16'), csrc/wide16.hip); the encoder stays exact fp32
    'c5': {'ref': 'BASELINE.json configs[4] (stress shape)', 'n_obj': 256, 'n_pts': 2048, 'pairs_per_gpu': 64, 'scaling': 'weak',
           'emb_dim': 1024, 'mfma_mode': 'f16'},
}
PEAK_F16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA (= fp32 vector) dense peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16X6_TFLOPS = PEAK_F16_TFLOPS / 6.0  # an fp32 product on three exact bf16 planes = six bf16 MFMAs: the pipe's ceiling in fp32-product terms
PMC_TRAFFIC_FILE = 'r06_pmc_traffic.csv'   # written by tools/pmc_traffic.sh on the GPU box, committed under profiles/
HITS_PAIRS = 8                   # fixed val-style subsample for the Hits@K half of the metric
NOISE_FLOOR = 2e-5               # floor of the rerun-noise unit in default_vs_exact_f32: 5 x the fp32-MFMA step's own table-gradient error against fp64 at configs[2]

COMPACT_LIMIT = 4096             # bytes: the driver keeps only the tail of stdout; the final line must fit and parse (tests/test_bench_line_cpu.py)
EXTRAS_FILE = 'bench_extras.json'


def _clip(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + '...'


def compact_roofline(r):
    """The dominant kernel's roofline object as the contract words it (bound, achieved, peak, unit, frac, traffic) plus frac_useful, the kernel's
    name (<= 120 characters) and its mean launch time; everything else about it goes to the extras file."""
    if not r:
        return None
    out = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac')}
    if r.get('frac_useful') is not None:
        out['frac_useful'] = r['frac_useful']
    out['traffic'] = r.get('traffic')
    out['kernel'] = _clip(r.get('kernel', ''), 120)
    out['avg_launch_ms'] = r.get('avg_launch_ms')
    if r.get('peak_is'):
        out['peak_is'] = _clip(r['peak_is'], 110)
    return out


def split_line(full):
    """(compact, extras): `compact` is the ONE JSON line the driver parses -- the contract's keys, the dominant kernel's roofline, the CPU
    baseline's best point, Hits@1 and the headline numbers of the extras, < COMPACT_LIMIT bytes whatever was measured; `extras` is the whole
    record (every roofline object, the sweeps, the other modes and workloads), written beside the script and echoed on earlier lines."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'median_ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'data')
    c = {k: full[k] for k in keep if k in full}
    c['dtype'] = _clip(full.get('dtype', ''), 160)
    cfg = full.get('config', {})
    c['config'] = {'workload': _clip(cfg.get('workload', ''), 200)}
    for k in ('name', 'global_pairs', 'pairs_per_gpu', 'objects_per_scene', 'points_per_object', 'emb_dim', 'parallelism', 'peak_hbm_gib'):
        if k in cfg:
            c['config'][k] = cfg[k]
    if 'modules' in cfg:
        c['config']['modules'] = '+'.join(cfg['modules'])
    c['roofline'] = compact_roofline(full.get('roofline'))
    others = full.get('roofline_other') or []
    if others:
        c['roofline_next'] = [{'kernel': _clip(r.get('kernel', ''), 48), 'bound': r.get('bound'), 'frac': r.get('frac'), 'step_ms': r.get('step_ms')}
                              for r in others[:3]]
    cb = full.get('cpu_baseline')
    if cb:
        best = next((p for p in cb.get('sweep', []) if p.get('pairs_per_s') == cb.get('value')), {})
        c['cpu_baseline'] = {'value': cb.get('value'), 'unit': cb.get('unit'), 'cores': cb.get('cores'), 'kind': cb.get('kind'),
                             'cpu_model': _clip(cb.get('cpu_model', ''), 48), 'b': best.get('b'), 'threads': best.get('threads'),
                             'sample': _clip(cb.get('sample', ''), 230)}
        c['speedup_vs_cpu_baseline'] = full.get('speedup_vs_cpu_baseline')
    h = full.get('hits_at_1')
    if h:
        c['hits_at_1'] = {k: h.get(k) for k in ('gpu', 'oracle', 'anchors', 'max_abs_embedding_err')}
    for key in ('extra_exact_f32', 'extra_c2', 'extra_pct'):
        e = full.get(key)
        if e:
            c[key] = {'value': e.get('value'), 'ms_per_step': e.get('ms_per_step')} if 'error' not in e else {'error': _clip(e['error'], 80)}
    e = full.get('extra_full_module_list')
    if e and 'error' not in e:
        c['extra_full_module_list'] = {'value': e.get('value'), 'at_configs2_size': (e.get('at_configs2_size') or {}).get('value')}
    d = full.get('default_vs_exact_f32')
    if d:
        c['default_vs_exact_f32'] = {k: d.get(k) for k in ('loss_rel_diff', 'max_grad_diff_rel_to_own_max', 'worst_param')}
    if full.get('collectives') is not None:
        col = full['collectives']
        # the self-certifying part (who took part, what a step moved per kind) stays on the line; per-call timings go to the extras
        c['collectives'] = {'backend': col.get('backend'), 'world_size': col.get('world_size'),
                            'ranks_seen': [{'rank': x.get('rank'), 'device': x.get('device')} for x in (col.get('ranks_seen') or [])][:16],
                            'bytes_per_step_this_rank': {k: col.get(k + '_bytes') for k in ('all_gather', 'reduce_scatter', 'all_reduce')},
                            'ms_alone_per_step': round(sum(t.get('ms_each', 0.0) * t.get('calls_in_timed_steps', 0) for t in col.get('timed_alone', []))
                                                       / max(1, full.get('steps', 1)), 3)}
    w = full.get('weak_scaling_point')
    if w:
        c['weak_scaling_point'] = {k: w.get(k) for k in ('value', 'ms_per_step', 'scaling') if k in w} if 'error' not in w else {'error': _clip(w['error'], 80)}
    c['extras'] = EXTRAS_FILE
    # belt and braces: whatever a future edit adds, the line the driver reads stays under the limit
    for drop in ('roofline_next', 'default_vs_exact_f32', 'collectives', 'weak_scaling_point', 'extra_full_module_list', 'extra_pct', 'extra_c2'):
        if len(json.dumps(c)) < COMPACT_LIMIT:
            break
        c.pop(drop, None)
    return c, full


def emit(full, out=None, root=ROOT):
    """Extras first (a side file + one stderr line per top-level key), then the ONE compact JSON line as the last line of stdout."""
    out = out or sys.stdout
    compact, extras = split_line(full)
    for path in (os.path.join(root, EXTRAS_FILE), os.path.join(root, 'gpurun_out', EXTRAS_FILE)):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, 'w') as f:
                    json.dump(extras, f, indent=1)
        except OSError:
            pass
    for k, v in extras.items():
        if k not in compact or compact[k] != v:
            print('[bench extra] ' + json.dumps({k: v}), file=sys.stderr, flush=True)
    text = json.dumps(compact)
    assert len(text) < COMPACT_LIMIT and '\n' not in text
    print(text, file=out, flush=True)
    return compact


def fp64_evidence():
    """What tests/test_fp64_chunked_gpu.py (the chunked fp64 evaluation of OverallLoss at configs[2]) last measured, from the committed
    report: errors against fp64 of the default step (sweeps on three exact bf16 planes), of the same step with fp32-MFMA sweeps, and of the
    opt-in two-plane fp16 mode, side by side."""
    for name in ('r06_c3_gradient_vs_fp64.json', 'r06_1024_gradient_vs_fp64.json'):
        try:
            r = json.load(open(os.path.join(ROOT, 'profiles', name)))
            e = r['meta_embedding_rel_err_vs_fp64_rel_to_own_max']
            return {'source': f'profiles/{name} (tests/test_fp64_chunked_gpu.py, {r["pairs"]} pairs x 128 objects)',
                    'table_grad_max_err_vs_fp64_rel_to_max': {m: {k: t[k + '_max_err_rel_to_max'] for k in ('bf16x6', 'f32') if k + '_max_err_rel_to_max' in t}
                                                              for m, t in r['tables'].items()},
                    'meta_embedding_rel_err_vs_fp64_rel_to_own_max': {k: e[k] for k in ('bf16x6', 'f32') if k in e},
                    'meta_embedding_rel_rerun_diff_rel_to_own_max': r.get('meta_embedding_rel_rerun_diff_rel_to_own_max')}
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(n_obj, n_pts, seconds_budget=45.0, emb_dim=100):
    """The oracle (CPU restatement pinned to the reference by tests/golden) on this box's host cores: fwd + loss + bwd with the same
    (objects, points, modules) as the GPU workload, over a BUDGETED sweep of batch sizes b (the reference's native 2 and 4,
    configs/scan3r/*.yaml, and a larger one) and thread counts (16, 32, 64): SURVEY 8(d) asks for the reference's best.
    The best point is the baseline; the whole sweep is attached.  The per-graph Python loop of the structure encoder (sg_aligner.py:86-110)
    is what stops it scaling with either."""
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch
    ncpu = os.cpu_count() or 1
    params = O.init_params(MODULES, seed=42, emb_dim=emb_dim)
    # (all hardware threads is NOT in the sweep: on the 2 x 64-core EPYC 9575F GPU host 256 threads run these small per-graph ops at 0.023 pairs/s,
    # 300 x slower than 32 threads -- 171 s per b = 4 step, measured once in round 5 -- and a torch op cannot be interrupted; 96 and 128 threads at
    # b = 16: 1.97 and 1.78 pairs/s against 4.05 at 32, profiles/r06_cpu_threads.txt)
    points = [(2, min(32, ncpu)), (4, min(32, ncpu)), (4, min(64, ncpu)), (16, min(32, ncpu)), (16, min(64, ncpu)), (2, min(16, ncpu))]
    seen, sweep, best = set(), [], None
    t_all = time.time()
    per_point = seconds_budget / len(points)
    batches = {}
    for b, th in points:
        if (b, th) in seen:
            continue
        seen.add((b, th))
        if time.time() - t_all > seconds_budget and best is not None:
            sweep.append({'b': b, 'threads': th, 'skipped': 'time budget'})
            continue
        torch.set_num_threads(th)
        if b not in batches:
            batches[b] = make_batch(b, n_obj, n_pts, seed=43, device='cpu')
        dd = batches[b]
        t0 = time.time()
        O.train_step(params, dd, MODULES)                  # warm-up (also the probe: a point whose single step eats the slice gets one more)
        warm = time.time() - t0
        times = []
        t_end = time.time() + max(0.0, per_point - warm)
        while len(times) < 2 or (time.time() < t_end and len(times) < 20):
            t0 = time.time()
            O.train_step(params, dd, MODULES)
            times.append(time.time() - t0)
            if len(times) >= 2 and warm > per_point:
                break
        med = float(np.median(times))
        rec = {'b': b, 'threads': th, 'pairs_per_s': round(b / med, 3), 'median_ms': round(med * 1e3, 1), 'iterations': len(times)}
        sweep.append(rec)
        if best is None or rec['pairs_per_s'] > best['pairs_per_s']:
            best = rec
    cpu_model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), 'unknown')
    except OSError:
        pass
    return {'value': best['pairs_per_s'], 'unit': 'pairs/s', 'cores': best['threads'], 'kind': 'port', 'cpu_model': cpu_model,
            'sample': f'oracle fwd+loss+bwd, best of a (batch, threads) sweep: b={best["b"]} pairs x {n_obj} obj x {n_pts} pts, emb_dim {emb_dim}, '
                      f'{"+".join(MODULES)}, {best["iterations"]} iterations, median {best["median_ms"]:.1f} ms, torch {torch.__version__} CPU, '
                      f'{best["threads"]} threads (host has {ncpu} hardware threads)',
            'sweep': sweep}


def hits_at_k(steps, n_obj, n_pts, dev):
    """BASELINE metric, second half: node-match Hits@K of the HIP path's embeddings vs the oracle's on identical inputs and
    weights -- a fixed 8-pair val-style batch (every common object an anchor, scan3r.py:83-87), ranking arithmetic of
    inference_align_reg.py:125-143 + utils/alignment.py on both sides.  Outside the timed region."""
    from oracle import sga_oracle as O
    from sgaligner_amd.synthetic import make_batch, to_device
    dd = make_batch(HITS_PAIRS, n_obj, n_pts, seed=1234, device='cpu', anchors='val')
    params = {k: v.detach().cpu().clone() for k, v in steps.model.state_dict().items() if 'num_batches' not in k}
    with torch.no_grad():
        out_o = O.encoder_forward(params, dd, MODULES)
    key = 'joint' if len(MODULES) > 1 else MODULES[0]
    m_o = O.evaluate_batch(out_o[key].detach(), dd)
    ddd = to_device(dd, dev)
    out_g = steps.test_step(0, ddd)
    m_g = steps.eval_step(0, ddd, out_g)
    tot = m_g[1]['total']
    emb_err = float((out_g[key].detach().cpu() - out_o[key].detach()).abs().max())
    return {'gpu': m_g[1]['correct'] / max(1, tot), 'oracle': m_o['hits'][1][0] / max(1, m_o['hits'][1][1]),
            'gpu_hits_1to5': [m_g[k]['correct'] for k in (1, 2, 3, 4, 5)],
            'oracle_hits_1to5': [m_o['hits'][k][0] for k in (1, 2, 3, 4, 5)], 'anchors': tot,
            'mrr_gpu': float(np.mean(m_g['mrr'])), 'mrr_oracle': float(np.mean(m_o['mrr'])),
            'max_abs_embedding_err': emb_err,
            'sample': f'{HITS_PAIRS} val-style pairs x {n_obj} obj x {n_pts} pts, random-init weights (seed 42), same inputs/weights both sides'}


def _finite(loss_dict, what):
    """A measured step whose loss is not finite is not a measurement (round 6: an M = 4 kernel variant returned NaN sums and the line still carried its rate)."""
    v = float(loss_dict['loss'].item())
    if not np.isfinite(v):
        raise FloatingPointError(f'{what}: loss = {v}')
    return v


def _sha16(path):
    try:
        return hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic_bytes(kernel_tag, workload_key, source):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/r04_pmc_traffic.csv, written
    by tools/pmc_traffic.sh: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of THIS script; KiB per dispatch; gfx950
    correction: FETCH_SIZE counts wide coalesced reads at half their size -> x2, MI355X_MICROARCH.md, HBM).
    A row is used only if it was taken on the same workload (`workload_key`) AND the kernel's source file is unchanged
    since (sha256 recorded by the tool): a stale measurement is reported as null, never as a number."""
    path = os.path.join(ROOT, 'profiles', PMC_TRAFFIC_FILE)
    sha = _sha16(os.path.join(ROOT, 'sgaligner_amd', 'csrc', source))
    try:
        f = w = None
        for line in open(path):
            if line.startswith('#') or line.startswith('kernel,'):
                continue
            parts = line.strip().split(';')
            if len(parts) != 6:
                continue
            k, wk, src_sha, counter, value, _ = parts
            if k != kernel_tag or wk != workload_key or src_sha != sha:
                continue
            if counter == 'FETCH_SIZE':
                f = float(value)
            elif counter == 'WRITE_SIZE':
                w = float(value)
        if f is None or w is None:
            return None
        return int((2.0 * f + w) * 1024)
    except OSError:
        return None


def roofline_objects(events, world, workload=None):
    """The kernels that carry the step, each timed with HIP events on its launch stream (ops.KERNEL_EVENTS), sorted by their time per
    step: [0] is the line's `roofline`, the rest `roofline_other`.  All are bound by the exact-fp32 MFMA rate (157.3 TFLOP/s dense;
    on gfx950 fp32 MFMA and fp32 VALU share the SIMD's FMA datapath -- tools/micro/mfma_valu_overlap.hip -- so epilogue VALU work adds
    to, rather than hides under, the MFMA time).  `achieved` = ALGORITHMIC FLOPs per launch / mean launch time."""
    from sgaligner_amd import ops
    roofs = []
    evs = events.get('pointnet_fwd_kernel', [])
    if evs:
        durs = [a.elapsed_time(b) for a, b, _ in evs]
        T, P, C1, C2, C3 = evs[0][2][:5]
        pmode = evs[0][2][5] if len(evs[0][2]) > 5 else 'f32'
        alg = 2.0 * T * P * (3 * C1 + C1 * C2 + C2 * C3)           # three per-point layers
        avg_ms = float(np.mean(durs))
        ach = alg / (avg_ms * 1e-3) / 1e12
        with_bn = bool(evs[0][2][6]) if len(evs[0][2]) > 6 else False
        bn_note = ('; the launch also sums the batch statistics of the reference\'s BatchNorm side effect (pointnet.py:141-159) -- the timed span includes '
                   'its point-moments and reduce kernels') if with_bn else ''
        if pmode == 'bf16x6':
            # three exact bf16 planes: six bf16 MFMAs per product; at C3 = 256 two workgroups share an object and both run layers 1-2
            executed = 2.0 * T * P * (3 * C1 + (2 if C3 == 256 else 1) * C1 * C2 + C2 * C3) * 6.0
            roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(PEAK_BF16X6_TFLOPS, 1), 'unit': 'TFLOP/s',
                          'frac': round(ach / PEAK_BF16X6_TFLOPS, 4),
                          'peak_is': 'dense bf16 MFMA peak (2500 TFLOP/s) / 6: an fp32 product on three exact bf16 planes is six bf16 MFMAs',
                          'achieved_over_fp32_mfma_peak': round(ach / PEAK_F32_TFLOPS, 4),
                          'frac_of_bf16_peak_executed': round(executed / (avg_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                          'traffic': pmc_traffic_bytes('pointnet_fwd_p3_kernel', f'T={T},P={P}', 'pointnet.hip') if world == 1 else None,
                          'kernel': 'pointnet_fwd_p3_kernel<256,true> (object encoder: 3 per-point layers + max-pool, one wave per object and channel half; fp32 operands as '
                                    'three exact bf16 planes, six bf16 MFMAs per product, fp32 accumulate' + bn_note + ')',
                          'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4), 'step_ms': round(avg_ms, 4),
                          'algorithmic_flops_per_launch': alg, 'executed_flops_per_launch': executed})
        else:
            roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': round(ach / PEAK_F32_TFLOPS, 4),
                          'traffic': pmc_traffic_bytes('pointnet_fwd_kernel', f'T={T},P={P}', 'pointnet.hip') if world == 1 else None,
                          'kernel': 'pointnet_fwd_kernel<256,true> (object encoder: 3 per-point layers + max-pool, one wave per object' + bn_note + ')',
                          'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4), 'step_ms': round(avg_ms, 4),
                          'algorithmic_flops_per_launch': alg})
    for key, grad in (('loss_multi_grad_bf16x6', True), ('loss_multi_sums_bf16x6', False)):
        evs = events.get(key, [])
        if not evs:
            continue
        durs = [a.elapsed_time(b) for a, b, _ in evs]
        ns, A, J1, J2, M = evs[0][2][:5]
        lite = (not grad) and len(evs[0][2]) > 5 and bool(evs[0][2][5])      # forward sums from the h and m planes only: 14 of 20 MFMAs per sub-step
        pairs = 2.0 * ns * (J1 + J2)                     # (anchor, negative) pairs of this rank's shard, both sides
        # ALGORITHMIC FLOPs exactly as for the fp32 sweeps (the four anchors x negatives products of all M+1 tables,
        # sum_tab D_tab = 100 M + 100 M; backward = 2 x forward).  The kernel multiplies only the M modality tables (joint derived) and
        # executes every product as six bf16 MFMAs: S is 20 MFMAs per 16 x 16 x 104 tile, the gradient GEMM 42 per 16 x 32 x 112; the
        # backward visits every pair twice (once per owner side).
        # The forward sums are priced on sum D = 100 M -- SURVEY's count with the joint table derived (S_J = sum_m beta_m S_m), which is what the
        # launch multiplies -- so that `frac` stays a fraction of the peak (SURVEY's 200 M would print 1.01 for it).
        d_sum = (100 * M + 100 * M) if grad else 100 * M
        alg = (2.0 if grad else 1.0) * (2.0 * d_sum * pairs)
        mfma_flops = 16 * 16 * 32 * 2.0
        # (M = 4 as well since round 6: ONE launch, the small-product accumulators shared by the tables -- sweep3_kernel's FOLD form)
        executed = (2.0 * pairs / 512.0 * M * 82 * mfma_flops) if grad else (pairs / 512.0 * M * (22 if lite else 40) * mfma_flops)
        # the ceiling in fp32-product terms: six bf16 MFMAs per product; the lite forward sums execute 11/20 of them (3.3 per product)
        peak = PEAK_F16_TFLOPS / (6.0 * (11.0 / 20.0 if lite else 1.0))
        useful = (3.0 if grad else 1.0) * 2.0 * 100 * M * pairs          # S once + the two gradient GEMMs (backward), sum D = 100 M
        avg_ms = float(np.mean(durs))
        ach = alg / (avg_ms * 1e-3) / 1e12
        roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                      'frac': round(ach / peak, 4),
                      'peak_is': ('dense bf16 MFMA peak (2500 TFLOP/s) / 3.3: the forward sums multiply the h and m planes only (h h + h m + m h + the K tail: 11 of the 20 MFMAs of a six-product sub-step)' if lite else
                                  'dense bf16 MFMA peak (2500 TFLOP/s) / 6: an fp32 product on three exact bf16 planes is six bf16 MFMAs'),
                      'frac_useful': round(useful / (avg_ms * 1e-3) / 1e12 / peak, 4),
                      'frac_useful_is': 'S once + two gradient GEMMs over the M modality tables (sum D = 100 M; the joint table is derived), same peak',
                      'achieved_over_fp32_mfma_peak': round(ach / PEAK_F32_TFLOPS, 4),
                      'executed_bf16_tflops': round(executed / (avg_ms * 1e-3) / 1e12, 1),
                      'frac_of_bf16_peak_executed': round(executed / (avg_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                      'traffic': pmc_traffic_bytes(f'sweep3_kernel<{M},{"true" if grad else "false"}>', f'ns={ns},A={A},J={J1 + J2}', 'sweep3.hip') if world == 1 else None,
                      'kernel': f'sweep3_kernel<{M},{"true" if grad else "false"}> ({"loss: negatives backward" if grad else "loss: global sums over anchors x negatives (forward)"}, '
                                f'all {M}+1 tables; ' + ('h and m planes (16 bits, unbiased) of the fp32 operands, fp32 accumulate' if lite else 'fp32 operands as three exact bf16 planes, six bf16 MFMAs per product, fp32 accumulate')
                                + ('; one launch, small-product accumulators shared by the four tables' if (grad and M == 4) else '') + ')',
                      'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4), 'step_ms': round(avg_ms, 4),
                      'algorithmic_flops_per_launch': alg, 'executed_flops_per_launch': executed})
    for key, grad in (('loss_multi_grad', True), ('loss_multi_sums', False)):
        evs = events.get(key, [])
        if not evs:
            continue
        durs = [a.elapsed_time(b) for a, b, _ in evs]
        ns, A, J1, J2, M = evs[0][2]          # ns = anchors in this rank's shard (== A on one GPU)
        # Algorithmic FLOPs of the four anchors x negatives products of all M+1 tables, sum_tab D_tab = 100 M + 100 M:
        # forward = sum_tab 2 D_tab * 2 ns (J1+J2); backward = 2 x that (one GEMM per side).
        d_sum = 100 * M + 100 * M
        alg = (2.0 if grad else 1.0) * (2.0 * d_sum * 2.0 * ns * (J1 + J2))
        avg_ms = float(np.mean(durs))
        info = ops.SWEEP_GRAD_INFO if grad else ops.SWEEP_SUMS_INFO
        executed = info['executed_flops'](ns, J1 + J2, M)
        if not grad:
            # The forward launch multiplies only the M modality tables (the joint similarities are derived: S_J = sum beta_m S_m), i.e.
            # HALF of SURVEY's count for it -- priced on what it executes, so that `frac` stays a fraction of the MFMA peak.
            alg = executed
        ach = alg / (avg_ms * 1e-3) / 1e12
        tag = info['tag'] % M if M != 4 else ('sweep16x2_kernel<true>' if grad else 'sweep16x2_kernel<false>')
        roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
                      'frac': round(ach / PEAK_F32_TFLOPS, 4),
                      'traffic': pmc_traffic_bytes(info['tag'] % M, f'ns={ns},A={A},J={J1 + J2}', 'contrastive.hip') if world == 1 else None,
                      'kernel': f'{tag} ({info["what"]}, all {M}+1 tables)',
                      'launches_timed': len(durs), 'avg_launch_ms': round(avg_ms, 4), 'step_ms': round(avg_ms, 4),
                      'algorithmic_flops_per_launch': alg, 'executed_flops_per_launch': executed,
                      'executed_tflops': round(executed / (avg_ms * 1e-3) / 1e12, 2),
                      'frac_useful': round((3.0 if grad else 1.0) * 2.0 * 100 * M * 2.0 * ns * (J1 + J2) / (avg_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4),
                      'frac_useful_is': 'S once + two gradient GEMMs over the M modality tables (sum D = 100 M; the joint table is derived, and the '
                                        'backward recomputes S in its second owner pass: 4 products executed for 3 useful)'})
    for key, what, mult in (('wide16_grad', 'loss: negatives backward on fp16-input MFMA -- one S pass writing the coefficient tile in both orientations '
                             '+ both gradient GEMMs, every table', 2.0),
                            ('wide16_sums', 'loss: global sums on fp16-input MFMA, every table', 1.0)):
        evs = events.get(key, [])
        if not evs:
            continue
        # one event pair per table per step; algorithmic FLOPs: forward = 2 Dp * 2A (J1 + J2) per table, backward = 2 x that
        total_ms = sum(a_.elapsed_time(b_) for a_, b_, _ in evs)
        widths = [shp[3] for _, _, shp in evs]
        n_steps = max(1, events.get('_steps', 0)) or 1
        alg = sum(mult * 2.0 * dp * 2.0 * A * (J1 + J2) for _, _, (A, J1, J2, dp) in evs) / n_steps
        step_ms = total_ms / n_steps
        ach = alg / (step_ms * 1e-3) / 1e12
        # HBM-side bytes per step from the PMC passes of tools/pmc_traffic_c5.sh (per template; the GEMM template's row also holds the four
        # stash-product launches of the A x A backward): only at the workload the passes ran on
        traffic = None
        if world == 1 and workload == 'c5:64x256x2048':      # configs[4] at 64 pairs: what tools/pmc_traffic_c5.sh ran
            tk = [pmc_traffic_bytes(f'wide16_batch_kernel<{t}>', 'c5:64x256x2048 per step', 'wide16.hip') for t in ((1, 2) if mult == 2.0 else (0,))]
            traffic = None if any(t is None for t in tk) else int(sum(tk))
        roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F16_TFLOPS, 4),
                      'traffic': traffic, 'kernel': f'wide16_kernel ({what}; widths {sorted(set(widths))})', 'launch_groups_timed': len(evs),
                      'avg_launch_ms': round(step_ms, 4), 'step_ms': round(step_ms, 4), 'algorithmic_flops_per_launch': alg,
                      'executed_flops_per_launch': alg * (1.5 if mult == 2.0 else 1.0),
                      'frac_executed': round(ach * (1.5 if mult == 2.0 else 1.0) / PEAK_F16_TFLOPS, 4),
                      'note': 'one entry = the launches of ONE step over all tables (per table backward: 1 coefficient launch + 2 GEMM launches, '
                              'the four sum families batched in each; forward: 1 launch), HIP events around each table\'s group'})
    for key, what, mult in (('wide16_aa_fwd', 'loss: anchors x anchors forward of the wide tables -- X1 X2^T and X2 X1^T blocks on the fp16 tile core + '
                             'the epilogue-only kernel', 2.0),
                            ('wide16_aa_bwd', 'loss: anchors x anchors backward of the wide tables -- the two similarity blocks again, the epilogue-only '
                             'kernel, max + convert passes and both stash products on the fp16 tile core', 4.0)):
        evs = events.get(key, [])
        if not evs:
            continue
        n_steps = max(1, events.get('_steps', 0)) or 1
        step_ms = sum(a_.elapsed_time(b_) for a_, b_, _ in evs) / n_steps
        flops = sum(mult * 2.0 * dsum * A * A for _, _, (A, dsum) in evs) / n_steps        # products of [A, D] x [D, A] / [A, A] x [A, D] shape, all tables
        ach = flops / (step_ms * 1e-3) / 1e12
        roofs.append({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F16_TFLOPS, 4),
                      'traffic': None, 'kernel': f'wide16_kernel + anchor_kernel<.., PRE> ({what})', 'launch_groups_timed': len(evs),
                      'avg_launch_ms': round(step_ms, 4), 'step_ms': round(step_ms, 4), 'algorithmic_flops_per_launch': flops,
                      'executed_flops_per_launch': flops,
                      'note': 'one entry = every launch of ONE step (HIP events around the group); the epilogue kernels are VALU work on A^2 pairs '
                              'per table, not matrix work: the fraction is that of the whole group'})
    # ---- the HBM-bound kernels SURVEY 8(d) names: fusion (2 x 4 T D M bytes per direction) and the GAT message passing (per layer and head-pair
    # launch: 256 x 4 B of features in and out per node; the 16 B/edge list only when the batch is not recognised as complete graphs --
    # then the attention kernels never read it).  `achieved` = algorithmic bytes / mean launch time; one object per kernel, step_ms = all
    # of its launches in a step.
    n_steps = max(1, events.get('_steps', 1))
    for key, what in (('fusion_fwd', 'MultiModalFusion forward (sg_aligner.py:30-35): M x [T, D] in, [T, M D] out'),
                      ('fusion_bwd', 'MultiModalFusion backward: [T, M D] + M x [T, D] in, M x [T, D] out')):
        evs = events.get(key, [])
        if evs:
            t_, d_, m_ = evs[0][2]
            avg_ms = float(np.mean([a_.elapsed_time(b_) for a_, b_, _ in evs]))
            by = (2.0 if key == 'fusion_fwd' else 3.0) * 4.0 * t_ * d_ * m_
            ach = by / (avg_ms * 1e-3) / 1e9
            roofs.append({'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': None,
                          'kernel': f'{key}_kernel ({what})', 'launches_timed': len(evs), 'avg_launch_ms': round(avg_ms, 4),
                          'step_ms': round(avg_ms * len(evs) / n_steps, 4), 'algorithmic_bytes_per_launch': by})
    for key, mult in (('gat_attn_fwd', 2.0), ('gat_attn_bwd', 3.0)):
        evs = events.get(key, [])
        if evs:
            t_, e_, complete = evs[0][2]
            avg_ms = float(np.mean([a_.elapsed_time(b_) for a_, b_, _ in evs]))
            by = mult * 4.0 * t_ * 256 + (0.0 if complete else 16.0 * e_)
            ach = by / (avg_ms * 1e-3) / 1e9
            roofs.append({'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': None,
                          'kernel': f'{key}_kernel (GATConv message passing, one launch per layer: {t_} nodes x 2 heads x 128 channels'
                                    + (', complete graphs: edge list not read' if complete else f', {e_} edges of 16 B') + ')',
                          'launches_timed': len(evs), 'avg_launch_ms': round(avg_ms, 4), 'step_ms': round(avg_ms * len(evs) / n_steps, 4),
                          'algorithmic_bytes_per_launch': by,
                          'note': 'N x N attention per graph is on-chip work (2 N^2 (2 x 128 + 8) FLOP per graph and head): at 128-node graphs the kernel is bound by '
                                  'that LDS / VALU work, not by these bytes'})
    roofs.sort(key=lambda r: -r['step_ms'])
    return roofs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=['auto', 'c2', 'c3', 'c5'], default=os.environ.get('SGA_BENCH_CONFIG', 'auto'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hits', action='store_true')
    ap.add_argument('--no-attr', action='store_true', help='skip the extra point+gat+rel+attr (M = 4) measurement (N = 1)')
    ap.add_argument('--no-attr-c3', action='store_true', help='skip the M = 4 measurement at the configs[2] size (N = 1, default config)')
    ap.add_argument('--no-scale-ref', action='store_true', help='skip the extra weak-scaling point (N > 1, --config auto)')
    ap.add_argument('--no-pct', action='store_true', help='skip the extra pct+gat+rel+attr small-batch measurement (N = 1)')
    ap.add_argument('--no-c2', action='store_true', help='skip the extra BASELINE configs[1] measurement (N = 1)')
    ap.add_argument('--no-exact', action='store_true', help='skip the extra measurement of the same step with fp32-MFMA sweeps (extra_exact_f32)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Launched bare (`python bench.py --gpus N`): create the N ranks ourselves -- one process per GPU under torch.distributed.run,
        # the same command line the driver uses -- instead of silently measuring one GPU.  With fewer GPUs than ranks (the 2-rank test on
        # a one-GPU box) the ranks share devices and talk over gloo: RCCL refuses two ranks on one device.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        env = dict(os.environ)
        if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
            env.setdefault('SGA_DIST_BACKEND', 'gloo')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    from sgaligner_amd import dist as sdist
    from sgaligner_amd import ops
    from sgaligner_amd.synthetic import make_batch_fast
    from sgaligner_amd.trainer import AlignerSteps

    rank, world, local = sdist.init_from_env()
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the product path has no CPU fallback)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world != args.gpus:
        raise SystemExit(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the line would be labelled with a rank count it did not run on')
    cname = args.config if args.config != 'auto' else 'c3'
    cfg = CONFIGS[cname]
    n_obj, n_pts = cfg['n_obj'], cfg['n_pts']
    if cfg['scaling'] == 'weak':
        my_pairs, total_pairs = cfg['pairs_per_gpu'], cfg['pairs_per_gpu'] * world
    else:
        lo, hi = sdist.shard_range(cfg['global_pairs'], rank, world)
        my_pairs, total_pairs = hi - lo, cfg['global_pairs']

    if cfg.get('mfma_mode'):
        ops.set_mfma_mode(cfg['mfma_mode'])
        args.no_c2 = args.no_attr = args.no_exact = True      # the extras belong to the fp32 configurations
    dtype_label = {'f16': 'f16-in/f32-acc (loss + ranking GEMMs on wide tables); f32 encoder',
                   'bf16x6': 'f32 (fp32 operands and fp32 accumulation throughout; the anchors x negatives loss sweeps multiply them as three exact bf16 planes, '
                             'six bf16 MFMAs per product, and so does the PointNet forward -- extra_exact_f32 is the same step with both on the fp32 MFMA)',
                   'f32': 'f32'}.get(ops.get_mfma_mode(), ops.get_mfma_mode())
    mode0 = ops.get_mfma_mode()            # the arithmetic of the headline: ops.DEFAULT_MFMA_MODE unless the configuration / SGA_MFMA_MODE says otherwise
    steps = AlignerSteps(MODULES, device=dev, seed=42, emb_dim=cfg.get('emb_dim', 100))
    dd = make_batch_fast(my_pairs, n_obj, n_pts, seed=43 + rank, device=dev)
    if world > 1 and cfg['scaling'] == 'strong' and cfg['global_pairs'] % world == 0:
        # every rank's shard has the same shape (uniform synthetic scenes): the step needs no layout all-gather / host read-back
        dd['_sga_layout'] = sdist.known_layout(dd['tot_obj_pts'].shape[0], len(dd['e1i']), len(dd['e1j']), len(dd['e2j']), world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(st, batch, warm, n):
        """`warm` untimed steps, then exactly `n` steps between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(warm):
            st.forward_backward(batch)
        barrier()
        evs = []
        t0 = time.perf_counter()
        for _ in range(n):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            _, ld = st.forward_backward(batch)
            e1.record()
            evs.append((e0, e1))
        barrier()
        el = time.perf_counter() - t0
        med = float(np.median([x.elapsed_time(y) for x, y in evs]))
        if world > 1:
            t = torch.tensor([el, med], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, med = float(t[0].item()), float(t[1].item())
        return el, med, ld

    for _ in range(args.warmup):
        steps.forward_backward(dd)
    barrier()
    ops.KERNEL_EVENTS = {}
    sdist.COLLECTIVE_EVENTS = [] if world > 1 else None
    elapsed, med_ms, loss_dict = timed(steps, dd, 0, args.steps)
    ops.KERNEL_EVENTS['_steps'] = args.steps
    events = ops.KERNEL_EVENTS
    ops.KERNEL_EVENTS = None
    coll_events, sdist.COLLECTIVE_EVENTS = sdist.COLLECTIVE_EVENTS, None
    loss_val = _finite(loss_dict, 'headline step')
    peak_gib = torch.cuda.max_memory_allocated() / 2 ** 30
    roofs = roofline_objects(events, world, workload=f'{cname}:{my_pairs}x{cfg["n_obj"]}x{cfg["n_pts"]}')

    # ---- extras, NOT the headline: the same steps in the other arithmetic modes (ops.set_mfma_mode), on the same batch and weights.
    #   extra_exact_f32: the sweeps on the fp32 MFMA (v_mfma_f32_16x16x4_f32, sweep16_kernel) -- the arithmetic every earlier round's headline
    #          ran in; its value, its roofline objects, and how far the DEFAULT step's gradients are from it in units of ITS OWN run-to-run
    #          differences (fp32 atomics: order-dependent sums).  The default (bf16x6: every fp32 operand as three exact bf16 planes, six bf16
    #          MFMAs per product, fp32 accumulate) is fp32 arithmetic on the same operands; tests/test_fp64_chunked_gpu.py holds it to the
    #          fp64 evaluation at this size.
    extra_exact = default_vs_exact = None
    want_exact = not args.no_exact and mode0 != 'f32'
    if want_exact:
        n_x = 2 if cname == 'c3' else max(2, min(args.steps, 10))
        head_grads = {n: p.grad.detach().clone() for n, p in steps.model.named_parameters() if p.grad is not None}
        ops.set_mfma_mode('f32')
        try:
            ops.KERNEL_EVENTS = {}
            el_f, _, ld_f = timed(steps, dd, 1 if cname == 'c3' else min(2, args.warmup), n_x)
            ops.KERNEL_EVENTS['_steps'] = n_x
            ev_f, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            ref_grads = {n: p.grad.detach().clone() for n, p in steps.model.named_parameters() if p.grad is not None}
            noise_runs = []
            for _ in range(2 if cname != 'c3' else 1):
                steps.forward_backward(dd)
                torch.cuda.synchronize()
                noise_runs.append({n: float((p.grad - ref_grads[n]).abs().max()) / max(1e-30, float(ref_grads[n].abs().max()))
                                   for n, p in steps.model.named_parameters() if p.grad is not None and n in ref_grads})
            f32_noise = {n: max(r[n] for r in noise_runs) for n in noise_runs[0]}
            gmax = max(float(g.abs().max()) for g in ref_grads.values())
            if want_exact:
                extra_exact = {'mode': "ops.set_mfma_mode('f32'): the loss sweeps on v_mfma_f32_16x16x4_f32 (sweep16_kernel) and the PointNet forward on v_mfma_f32_32x32x2_f32 "
                                       "(pointnet_fwd_kernel); everything else as in the headline",
                               'value': round(total_pairs * n_x / el_f, 2), 'unit': 'pairs/s', 'ms_per_step': round(el_f / n_x * 1e3, 3), 'steps': n_x,
                               'dtype': 'f32', 'loss': float(ld_f['loss'].item()), 'roofline': roofline_objects(ev_f, world)}
                # The gradient comparison isolates the LOSS arithmetic: one more fp32-MFMA-sweep step with the object encoder's forward as in the
                # headline (three exact bf16 planes), so that both steps route the max-pool's gradients through the same arg-max points (two fp32
                # summation orders pick different points for ~1 in 10^6 near-tied maxima; the encoder's own parity: tests/test_pointnet_gpu.py)
                cmp_grads = ref_grads
                if ops._POINTNET_MODE['f32'] != ops._POINTNET_MODE[mode0]:
                    pn_saved = ops._POINTNET_MODE['f32']
                    ops._POINTNET_MODE['f32'] = ops._POINTNET_MODE[mode0]
                    try:
                        steps.forward_backward(dd)
                        torch.cuda.synchronize()
                        cmp_grads = {n: p.grad.detach().clone() for n, p in steps.model.named_parameters() if p.grad is not None}
                    finally:
                        ops._POINTNET_MODE['f32'] = pn_saved
                worst_ratio, worst_ratio_name, worst_own, worst_own_name = 0.0, None, 0.0, None
                for n, g in head_grads.items():
                    if n not in cmp_grads:
                        continue
                    own = float((g - cmp_grads[n]).abs().max()) / max(1e-30, float(cmp_grads[n].abs().max()))
                    ratio = own / max(f32_noise.get(n, 0.0), NOISE_FLOOR)
                    if own > worst_own:
                        worst_own, worst_own_name = own, n
                    if ratio > worst_ratio:
                        worst_ratio, worst_ratio_name = ratio, n
                default_vs_exact = {'loss_rel_diff': abs(loss_val - float(ld_f['loss'].item())) / max(1e-30, abs(loss_val)),
                                    'max_grad_diff_rel_to_own_max': worst_own, 'worst_param': worst_own_name,
                                    'f32_rerun_diff_rel_to_own_max_same_param': f32_noise.get(worst_own_name),
                                    'max_diff_over_f32_rerun_diff': round(worst_ratio, 3), 'max_diff_over_rerun_param': worst_ratio_name,
                                    'noise_floor_rel_to_own_max': NOISE_FLOOR,
                                    'note': 'headline (default arithmetic) gradients against the fp32-MFMA-sweep step on the same batch and weights, both with the headline\'s '
                                            'object-encoder forward (same max-pool arg-maxes); where the two differ by more than '
                                            'the rerun difference (meta_embedding_rel.*: the default projects the gradient of nearly parallel rows without forming its radial part) '
                                            'the fp64 evaluation decides: fp64_evidence',
                                    'fp64_evidence': fp64_evidence()}
        finally:
            ops.set_mfma_mode(mode0)
            ops.KERNEL_EVENTS = None
        del head_grads, ref_grads

    # ---- extras at N = 1 under --config auto (the headline is configs[2]): BASELINE configs[1] (512 pairs x 64 objects x 512 pts) as
    # its own full measurement with its own roofline object, and the same batch with the reference's full module list
    # (point+gat+rel+attr, M = 4: what every yaml the reference ships trains; SURVEY.md 8(d) "also report +'attr'").  Exact fp32.
    extra_c2 = extra_attr = None
    if world == 1 and not args.no_c2 and (cname != 'c2'):
        try:
            del dd
            torch.cuda.empty_cache()
            c2 = CONFIGS['c2']
            dd2 = make_batch_fast(c2['pairs_per_gpu'], c2['n_obj'], c2['n_pts'], seed=43, device=dev)
            for _ in range(3):
                steps.forward_backward(dd2)
            torch.cuda.synchronize()
            ops.KERNEL_EVENTS = {}
            n2 = max(5, min(args.steps, 20))
            el2, med2, ld2 = timed(steps, dd2, 0, n2)
            _finite(ld2, 'extra_c2')
            ops.KERNEL_EVENTS['_steps'] = n2
            ev2, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            extra_c2 = {'workload': f'{c2["ref"]}: {c2["pairs_per_gpu"]} pairs x {c2["n_obj"]} objects x {c2["n_pts"]} pts, modules '
                                    f'{"+".join(MODULES)}, batch-global loss', 'value': round(c2['pairs_per_gpu'] * n2 / el2, 2), 'unit': 'pairs/s',
                        'ms_per_step': round(el2 / n2 * 1e3, 3), 'median_ms_per_step': round(med2, 3), 'steps': n2, 'warmup': 3,
                        'dtype': dtype_label, 'roofline': roofline_objects(ev2, world)}
        except Exception as e:
            extra_c2 = {'error': f'{type(e).__name__}: {e}'}
            dd2 = None
    elif world == 1 and cname == 'c2':
        dd2 = dd
    else:
        dd2 = None
    if world == 1 and not args.no_attr and 'attr' not in MODULES and dd2 is not None:
        try:
            mods4 = MODULES + ['attr']
            steps4 = AlignerSteps(mods4, device=dev, seed=42)
            n4 = max(3, min(args.steps, 10))
            for _ in range(2):
                steps4.forward_backward(dd2)
            torch.cuda.synchronize()
            ops.KERNEL_EVENTS = {}
            el4, _, ld4 = timed(steps4, dd2, 0, n4)
            _finite(ld4, 'extra_full_module_list')
            ops.KERNEL_EVENTS['_steps'] = n4
            ev4, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            extra_attr = {'modules': mods4, 'workload': 'BASELINE.json configs[1] shape (512 pairs x 64 objects x 512 pts)',
                          'value': round(CONFIGS['c2']['pairs_per_gpu'] * n4 / el4, 2), 'unit': 'pairs/s',
                          'ms_per_step': round(el4 / n4 * 1e3, 3), 'steps': n4, 'warmup': 2,
                          'dtype': dtype_label,
                          'roofline': roofline_objects(ev4, world)}
            # ... and at the HEADLINE's size: configs[2] (4096 pairs x 128 objects x 512 pts on this GPU) with the full module list
            if cname == 'c3' and not args.no_attr_c3:
                dd2 = None
                torch.cuda.empty_cache()
                c3 = CONFIGS['c3']
                dd3 = make_batch_fast(c3['global_pairs'], c3['n_obj'], c3['n_pts'], seed=43, device=dev)
                steps4.forward_backward(dd3)
                torch.cuda.synchronize()
                ops.KERNEL_EVENTS = {}
                el43, _, ld43 = timed(steps4, dd3, 0, 2)
                _finite(ld43, 'extra_full_module_list.at_configs2_size')
                ops.KERNEL_EVENTS['_steps'] = 2
                ev43, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
                extra_attr['at_configs2_size'] = {'workload': f'{c3["ref"]} shape: {c3["global_pairs"]} pairs x {c3["n_obj"]} objects x {c3["n_pts"]} pts, modules {"+".join(mods4)}',
                                                  'value': round(c3['global_pairs'] * 2 / el43, 2), 'unit': 'pairs/s', 'ms_per_step': round(el43 / 2 * 1e3, 1), 'steps': 2,
                                                  'warmup': 1, 'roofline': roofline_objects(ev43, world)[:3]}
                del dd3
            del steps4
        except Exception as e:
            extra_attr = {'error': f'{type(e).__name__}: {e}'} if extra_attr is None else dict(extra_attr, error_at_configs2_size=f'{type(e).__name__}: {e}')
    dd2 = None
    torch.cuda.empty_cache()

    # ---- extra at N = 1: the reference's OWN default configuration -- configs/scan3r/scan3r_ground_truth.yaml: modules pct+gat+rel+attr,
    # 4 pairs per step, ~40 objects per scene, 512 points -- where the 'pct' object encoder is ~95 % of the step.  Exact fp32; the
    # algorithmic rate counts the encoder's FLOPs only (forward x 3 for forward + backward; FLOPs per object stated).
    extra_pct = None
    if world == 1 and not args.no_pct and cname != 'c5':
        try:
            from sgaligner_amd.synthetic import make_batch, to_device
            modsp = ['pct', 'gat', 'rel', 'attr']
            stepsp = AlignerSteps(modsp, device=dev, seed=42)
            ddp = [to_device(make_batch(4, 40, 512, seed=7 + i, ragged=True), dev) for i in range(4)]
            for i in range(6):
                stepsp.forward_backward(ddp[i % 4])
            torch.cuda.synchronize()
            npct = 40
            tp = time.perf_counter()
            for i in range(npct):
                stepsp.forward_backward(ddp[i % 4])
            torch.cuda.synchronize()
            elp = (time.perf_counter() - tp) / npct
            n_objs = float(np.mean([int(d['tot_obj_pts'].shape[0]) for d in ddp]))
            Np = 512
            conv = 2.0 * Np * (3 * 128 + 128 * 128 + 4 * (128 * 32 + 2 * 128 * 128) + 512 * 1024) + 2.0 * (1024 * 512 + 512 * 256)
            attn = 4 * 2.0 * Np * Np * (32 + 128)
            alg = 3.0 * (conv + attn) * n_objs
            extra_pct = {'workload': "reference default (scan3r_ground_truth.yaml): modules pct+gat+rel+attr, 4 pairs x ~40 objects x 512 pts per step",
                         'value': round(4 / elp, 1), 'unit': 'pairs/s', 'ms_per_step': round(elp * 1e3, 3), 'steps': npct, 'dtype': 'f32',
                         'objects_per_step': n_objs, 'encoder_gflop_per_object_forward': round((conv + attn) / 1e9, 3),
                         'encoder_algorithmic_tflops': round(alg / elp / 1e12, 1), 'frac_of_fp32_mfma_peak': round(alg / elp / 1e12 / PEAK_F32_TFLOPS, 3),
                         'note': 'whole-step wall time; FLOPs = 3 x the NaivePCT forward (pct.py:275-317) of the step\'s objects'}
            del stepsp, ddp
        except Exception as e:
            extra_pct = {'error': f'{type(e).__name__}: {e}'}

    # ---- extra at N > 1 under --config auto: the weak-scaling point (BASELINE configs[1] per GPU: 512 pairs x 64 objects on every
    # rank, batch-global loss over 512 N pairs) next to the strong-scaling headline of the same line.
    weak_ref = None
    if world > 1 and args.config == 'auto' and not args.no_scale_ref:
        try:
            c2 = CONFIGS['c2']
            ddw = make_batch_fast(c2['pairs_per_gpu'], c2['n_obj'], c2['n_pts'], seed=143 + rank, device=dev)
            nw = 3
            elw, _, _ = timed(steps, ddw, 1, nw)
            weak_ref = {'workload': f'{c2["ref"]} per GPU: {c2["pairs_per_gpu"]} pairs x {c2["n_obj"]} objects x {c2["n_pts"]} pts on each of {world} '
                                    f'GPUs, batch-global loss over {c2["pairs_per_gpu"] * world} pairs', 'scaling': 'weak',
                        'value': round(c2['pairs_per_gpu'] * world * nw / elw, 2), 'unit': 'pairs/s', 'ms_per_step': round(elw / nw * 1e3, 2),
                        'steps': nw, 'warmup': 1, 'dtype': dtype_label}
            del ddw
        except Exception as e:
            weak_ref = {'error': f'{type(e).__name__}: {e}'}

    collectives = sdist.collective_summary(coll_events, args.steps, dev) if world > 1 else None

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        if cfg.get('mfma_mode') == 'f16':       # the configuration is ABOUT the fp16 GEMMs: they lead, whatever their share of the step
            roofs.sort(key=lambda r: (0 if 'wide16_kernel (loss: negatives backward' in r['kernel'] else 1, -r['step_ms']))
        roof = roofs[0] if roofs else None
        per = f'{my_pairs} pairs/GPU' if world > 1 else f'{my_pairs} pairs'
        line = {
            'metric': 'subscan-pairs/sec (fwd+bwd) + node-match Hits@1 vs reference',
            'value': round(total_pairs * args.steps / elapsed, 2), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'median_ms_per_step': round(med_ms, 3), 'value_median': round(total_pairs / (med_ms * 1e-3), 2),
            'higher_is_better': True, 'scaling': cfg['scaling'], 'vs_baseline': None,
            'dtype': dtype_label, 'data': 'synthetic',
            'config': {'workload': f'{cfg["ref"]}: {total_pairs} synthetic subscan pairs ({per}) x {n_obj} objects x '
                                   f'{n_pts} pts, modules {"+".join(MODULES)} (P+S+R), batch-global ICL/IAL loss over '
                                   f'{total_pairs} pairs', 'name': cname, 'global_pairs': total_pairs, 'pairs_per_gpu': my_pairs,
                       'objects_per_scene': n_obj, 'points_per_object': n_pts, 'emb_dim': cfg.get('emb_dim', 100), 'modules': MODULES,
                       'parallelism': f'dp{world}',
                       'loss': loss_val, 'peak_hbm_gib': round(peak_gib, 2)},
            'roofline': roof,
            'roofline_other': roofs[1:],
        }
        if collectives is not None:
            line['collectives'] = collectives
        if extra_exact is not None:
            line['extra_exact_f32'] = extra_exact
            line['default_vs_exact_f32'] = default_vs_exact
        if extra_c2 is not None:
            line['extra_c2'] = extra_c2
        if extra_attr is not None:
            line['extra_full_module_list'] = extra_attr
        if extra_pct is not None:
            line['extra_pct'] = extra_pct
        if weak_ref is not None:
            line['weak_scaling_point'] = weak_ref
        if not args.no_hits:
            line['hits_at_1'] = hits_at_k(steps, n_obj, n_pts, dev)
        if not args.no_cpu_baseline and world == 1:         # the CPU leg runs on rank 0 of a one-GPU run only
            line['cpu_baseline'] = cpu_baseline(n_obj, n_pts, emb_dim=cfg.get('emb_dim', 100))
            line['speedup_vs_cpu_baseline'] = round(line['value'] / line['cpu_baseline']['value'], 1)
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
