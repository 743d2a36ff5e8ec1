"""Error of every mode's OverallLoss gradients against the fp64 oracle, side by side (same inputs): python tools/dbg/modes_vs_fp64.py [pairs nobj seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from test_bf16x6_gpu import _overall_vs_fp64
cases = [(64, 64, 23), (16, 40, 5), (3, 30, 9)] if len(sys.argv) < 4 else [tuple(int(a) for a in sys.argv[1:4])]
modes = tuple(os.environ.get('MODES', 'f32,f16x2,bf16x6').split(','))
for c in cases:
    errs = _overall_vs_fp64(*c, modes=modes)
    print('case', c)
    for k in errs[modes[0]]:
        print(f'  {k:14s} ' + '  '.join(f'{m} {errs[m][k]:.3e}' for m in modes))
