"""|column mean|^2 of each modality's L2-normalised embedding rows on the bench's batch and weights (1 = identical rows, 0 = centred)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd.synthetic import make_batch_fast
from sgaligner_amd.trainer import AlignerSteps
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mods = (sys.argv[2] if len(sys.argv) > 2 else 'point,gat,rel').split(',')
dd = make_batch_fast(pairs, 64, 512, seed=43, device='cuda')
steps = AlignerSteps(mods, device='cuda', seed=42)
with torch.no_grad():
    out = steps.model(dd)
for k in mods:
    z = torch.nn.functional.normalize(out[k].float(), dim=1)
    zb = z.mean(0)
    print(k, '|zbar|^2 = %.5f' % float((zb * zb).sum()), ' rms spread per row = %.4f' % float((z - zb).norm(dim=1).mean()))
