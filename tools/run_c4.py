#!/usr/bin/env python3
"""BASELINE.json configs[3] runner: the reference's alignment evaluation (src/inference/sgaligner/inference_align_reg.py, the
alignment half of `AlignerRegTester`: test_step :74-76, eval_step :98-143, compute_metrics :85-96) on the 3RScan / 3DSSG
validation sub-scan pairs with a released checkpoint, through the HIP path.

The data and the checkpoint are not in the build container (README.md:69,74 of the reference): point the script at them and it
runs as is --

    SGA_3RSCAN_ROOT   directory holding `scans/<scan_id>/data.npy` and `files/<mode>/{data/*.pkl, anchors*_val.json}`
                      (the reference's cfg.data.subscan_dir after preprocessing/scan3r/preprocess.py)
    SGA_CHECKPOINT    a snapshot written by the reference (`epoch-*.pth.tar`: {'model': state_dict, ...}, keys may carry `module.`)
                      or by sgaligner_amd.epoch_trainer; loaded with strict=True after the prefix is stripped
    SGA_MODULES       default point,gat,rel,attr (configs/scan3r/scan3r_ground_truth.yaml:5 uses pct,gat,rel,attr)
    SGA_PC_RES        points per object, default 512 (cfg.val.pc_res)            SGA_DATA_MODE  default orig (cfg.val.data_mode)
    SGA_OVERLAP_LOW / SGA_OVERLAP_HIGH   the val overlap window (both 0: all pairs)      SGA_BATCH  pairs per batch, default 32
    SGA_ANCHOR_TYPE   cfg.preprocess.anchor_type_name, default ''

and prints ONE JSON line: the reference's metrics dict (`hits@_k`, `mrr`, `sgar_2/50/100`, exactly `compute_metrics`' keys and
rounding) + pair / anchor counts and throughput.  `--synthetic [N]` writes an N-pair dataset in the reference's on-disk layout
(sgaligner_amd/datasets/synthetic_scan3r.py) and a random-init snapshot to a temp dir first -- the CI mode
(tests/test_run_c4_gpu.py), where the same numbers are also produced by the oracle on the CPU (`--check-oracle`)."""
import argparse
import json
import os
import os.path as osp
import sys
import tempfile
import time
from collections import OrderedDict

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def new_meter(all_k, recall_modes):
    """alignment_metrics_meter of inference_align_reg.py:35-44."""
    m = {'mrr': [], 'sgar': {r: [] for r in recall_modes}}
    for k in all_k:
        m[k] = {'correct': 0, 'total': 0}
    return m


def accumulate(meter, batch_res, all_k, recall_modes):
    meter['mrr'] += list(batch_res['mrr'])
    for k in all_k:
        meter[k]['correct'] += batch_res[k]['correct']
        meter[k]['total'] += batch_res[k]['total']
    for r in recall_modes:
        meter['sgar'][r] += list(batch_res['sgar'][r])


def compute_metrics(result_dict):
    """inference_align_reg.py:85-96."""
    out = {}
    for key in result_dict:
        if type(key) == int:
            out['hits@_{}'.format(key)] = round(result_dict[key]['correct'] / max(1, result_dict[key]['total']), 5)
        elif type(result_dict[key]) == list:
            out[key] = round(float(np.array(result_dict[key]).mean()), 5) if len(result_dict[key]) else float('nan')
        elif type(result_dict[key]) == dict:
            for mode in result_dict[key]:
                v = result_dict[key][mode]
                out[key + '_' + mode] = round(float(np.array(v).mean()), 5) if len(v) else float('nan')
    return out


def load_checkpoint(model, path):
    state = torch.load(path, map_location='cpu', weights_only=False)
    sd = state['model'] if isinstance(state, dict) and 'model' in state else state
    sd = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    model.load_state_dict(sd, strict=True)                 # a released checkpoint must fit key for key
    return len(sd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, nargs='?', const=24, default=0, help='write an N-pair synthetic dataset + random snapshot and run on it')
    ap.add_argument('--check-oracle', action='store_true', help='also run the CPU oracle on every batch and compare the meters (small data only)')
    ap.add_argument('--max-batches', type=int, default=0)
    args = ap.parse_args()

    from sgaligner_amd.datasets import DeviceBatch, Scan3RDataset, synthetic_scan3r as S
    from sgaligner_amd.trainer import AlignerSteps

    modules = os.environ.get('SGA_MODULES', 'point,gat,rel,attr').split(',')
    pc_res = int(os.environ.get('SGA_PC_RES', '512'))
    batch = int(os.environ.get('SGA_BATCH', '32'))
    all_k, recall_modes = [1, 2, 3, 4, 5], ['2', '50', '100']          # cfg.metrics.all_k; inference_align_reg.py:41
    assert torch.cuda.is_available(), 'run_c4.py needs the HIP device (no CPU path)'
    steps = AlignerSteps(modules, device='cuda:0', seed=42)

    if args.synthetic:
        root = tempfile.mkdtemp(prefix='sga_c4_')
        pc_res = 64 if 'SGA_PC_RES' not in os.environ else pc_res
        S.write_dataset(root, n_pairs=args.synthetic, seed=3, resolutions=(pc_res,))
        ckpt = osp.join(root, 'epoch-0.pth.tar')
        torch.save({'epoch': 0, 'model': OrderedDict(('module.' + k, v.cpu()) for k, v in steps.model.state_dict().items())}, ckpt)
        source = f'synthetic on-disk dataset ({args.synthetic} pairs, reference layout), random-init snapshot'
    else:
        root, ckpt = os.environ.get('SGA_3RSCAN_ROOT'), os.environ.get('SGA_CHECKPOINT')
        if not root or not ckpt:
            sys.exit('run_c4.py: set SGA_3RSCAN_ROOT and SGA_CHECKPOINT (or pass --synthetic); see the header of this file')
        source = f'{root} + {ckpt}'
    cfg = S.make_cfg(root, pc_res=pc_res, data_mode=os.environ.get('SGA_DATA_MODE', 'orig'), modules=modules, batch_size=batch,
                     overlap_low=float(os.environ.get('SGA_OVERLAP_LOW', '0')), overlap_high=float(os.environ.get('SGA_OVERLAP_HIGH', '0')))
    cfg.preprocess.anchor_type_name = os.environ.get('SGA_ANCHOR_TYPE', '')
    n_keys = load_checkpoint(steps.model, ckpt)
    steps.model.eval()

    ds = Scan3RDataset(cfg, 'val')
    loader = torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=False, collate_fn=ds.collate_fn, num_workers=0)
    meter = new_meter(all_k, recall_modes)
    meter_o = new_meter(all_k, recall_modes) if args.check_oracle else None
    params = {k: v.detach().cpu().clone() for k, v in steps.model.state_dict().items() if 'num_batches' not in k}
    key = 'joint' if len(modules) > 1 else modules[0]
    n_pairs = n_obj = 0
    t_gpu = 0.0
    for it, dd in enumerate(loader):
        if args.max_batches and it >= args.max_batches:
            break
        ddd = DeviceBatch(dd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = steps.test_step(it, ddd)
        res = steps.eval_step(it, ddd, out, all_k=tuple(all_k))
        torch.cuda.synchronize()
        t_gpu += time.perf_counter() - t0
        accumulate(meter, res, all_k, recall_modes)
        n_pairs += int(dd['batch_size'])
        n_obj += int(np.sum(dd['tot_obj_count']))
        if meter_o is not None:
            from oracle import sga_oracle as O                    # checker only (CI mode)
            with torch.no_grad():
                m_o = O.evaluate_batch(O.encoder_forward(params, dd, modules)[key], dd, ks=tuple(all_k))
            meter_o['mrr'] += list(m_o['mrr'])
            for k in all_k:
                meter_o[k]['correct'] += m_o['hits'][k][0]
                meter_o[k]['total'] += m_o['hits'][k][1]
            for r in recall_modes:
                meter_o['sgar'][r] += list(m_o['sgar'][r])
    line = {'config': 'BASELINE.json configs[3]: 3RScan/3DSSG val sub-scan pairs, alignment metrics of inference_align_reg.py',
            'source': source, 'modules': modules, 'points_per_object': pc_res, 'checkpoint_keys': n_keys, 'pairs': n_pairs,
            'objects': n_obj, 'anchors': meter[all_k[0]]['total'], 'metrics': compute_metrics(meter),
            'pairs_per_s_encoder_plus_metrics': round(n_pairs / max(t_gpu, 1e-9), 1)}
    if meter_o is not None:
        line['oracle_metrics'] = compute_metrics(meter_o)
        line['hits_equal_oracle'] = all(meter[k] == meter_o[k] for k in all_k)
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
