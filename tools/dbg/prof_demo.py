import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgaligner_amd.datasets import Scan3RDataset, synthetic_scan3r as S
from sgaligner_amd.epoch_trainer import Trainer
root = tempfile.mkdtemp(prefix='sga_demo_')
S.write_dataset(root, n_pairs=96, seed=0, resolutions=(128,))
cfg = S.make_cfg(root, pc_res=128, modules=['point', 'gat', 'rel', 'attr'], batch_size=16, max_epoch=6, lr=2e-3, output_dir=os.path.join(root, 'run'))
np.random.seed(0)
tr = Trainer(cfg, log_steps=1000)
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); tr.run(); dt = time.time() - t0
pr.disable()
print('6 epochs x', len(tr.train_loader), 'train iterations (+ val) in %.2f s' % dt)
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
