import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import sga_oracle as O
from sgaligner_amd.synthetic import make_batch
mods = ['point', 'gat', 'rel']
dd = make_batch(2, 64, 512, seed=43)
params = O.init_params(mods, seed=42)
print('cpu_count', os.cpu_count())
for th in (4, 8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    O.train_step(params, dd, mods)
    ts = []
    for _ in range(3):
        t0 = time.time(); O.train_step(params, dd, mods); ts.append(time.time() - t0)
    print(th, 'threads:', round(min(ts) * 1e3, 1), 'ms/iter ->', round(2 / min(ts), 2), 'pairs/s', flush=True)
