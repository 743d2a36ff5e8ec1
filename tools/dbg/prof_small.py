import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgaligner_amd.synthetic import make_batch, to_device
from sgaligner_amd.trainer import AlignerSteps
steps = AlignerSteps(['point', 'gat', 'rel', 'attr'], device='cuda', seed=42)
dds = [to_device(make_batch(4, 40, 512, seed=7 + i, ragged=True), 'cuda') for i in range(4)]
for i in range(8):
    steps.forward_backward(dds[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    steps.forward_backward(dds[i % 4])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(40)
