"""End-to-end sanity run: synthetic on-disk Scan3R-format dataset -> Scan3RDataset -> engine.Trainer (Adam) -> Hits@K on the
validation split before and after training.  Objects shared by the two scans of a pair have the same shape, so alignment
is learnable; the point is that the whole HIP pipeline (not just single-step gradients) optimises the objective."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sgaligner_amd.datasets import DeviceBatch, Scan3RDataset, synthetic_scan3r as S
from sgaligner_amd.epoch_trainer import Trainer

mods = sys.argv[1].split(',') if len(sys.argv) > 1 else ['point', 'gat', 'rel', 'attr']
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
root = tempfile.mkdtemp(prefix='sga_demo_')
S.write_dataset(root, n_pairs=96, seed=0, resolutions=(128,))
cfg = S.make_cfg(root, pc_res=128, modules=mods, batch_size=16, max_epoch=epochs, lr=2e-3, output_dir=os.path.join(root, 'run'))
np.random.seed(0)
tr = Trainer(cfg, log_steps=1000)

def hits():
    tr.set_eval_mode()
    tot = {k: [0, 0] for k in (1, 3, 5)}
    mrr = []
    for dd in tr.val_loader:
        ddd = DeviceBatch(dd)
        out = tr.steps.test_step(0, ddd)
        m = tr.steps.eval_step(0, ddd, out)
        for k in tot:
            tot[k][0] += m[k]['correct']; tot[k][1] += m[k]['total']
        mrr += m['mrr']
    tr.set_train_mode()
    return {k: v[0] / max(1, v[1]) for k, v in tot.items()}, float(np.mean(mrr))

h0, m0 = hits()
t0 = time.time()
hist = tr.run()
dt = time.time() - t0
h1, m1 = hits()
print(f'modules {mods}: {epochs} epochs x {len(tr.train_loader)} iterations of 16 pairs in {dt:.1f} s')
print(f'  train loss {hist[0]["train"]["loss"]:.3f} -> {hist[-1]["train"]["loss"]:.3f};  val loss {hist[0]["val"]["loss"]:.3f} -> {hist[-1]["val"]["loss"]:.3f}')
print(f'  Hits@1 {h0[1]:.3f} -> {h1[1]:.3f}   Hits@3 {h0[3]:.3f} -> {h1[3]:.3f}   Hits@5 {h0[5]:.3f} -> {h1[5]:.3f}   MRR {m0:.3f} -> {m1:.3f}')
