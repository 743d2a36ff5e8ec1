import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from oracle import sga_oracle as O
from sgaligner_amd.synthetic import make_batch
mods = ['point', 'gat', 'rel']
dd = make_batch(16, 128, 512, seed=43)
params = O.init_params(mods, seed=42)
print('cpu_count', os.cpu_count(), flush=True)
for th in (32, 64, 96, 128):
    torch.set_num_threads(th)
    t0 = time.time(); O.train_step(params, dd, mods); w = time.time() - t0
    ts = []
    for _ in range(2):
        t0 = time.time(); O.train_step(params, dd, mods); ts.append(time.time() - t0)
    print(th, 'threads: warm', round(w, 2), 's; best', round(min(ts), 2), 's/iter ->', round(16 / min(ts), 3), 'pairs/s', flush=True)
