"""GPU: tools/run_c4.py -- the configs[3] runner (alignment evaluation of inference_align_reg.py on a val split + a snapshot) --
in its CI mode: synthetic on-disk dataset in the reference's layout, reference-style snapshot (`module.`-prefixed keys, strict
load), the reference's meter dict and `compute_metrics` keys; the same meters come out of the oracle on the CPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('modules', ['point,gat,rel,attr', 'point'])
def test_run_c4_synthetic_matches_oracle(modules):
    env = dict(os.environ, SGA_MODULES=modules, SGA_BATCH='5')
    for k in ('SGA_3RSCAN_ROOT', 'SGA_CHECKPOINT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_c4.py'), '--synthetic', '12', '--check-oracle'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['pairs'] == 12 and line['anchors'] > 0
    want = {'hits@_1', 'hits@_2', 'hits@_3', 'hits@_4', 'hits@_5', 'mrr', 'sgar_2', 'sgar_50', 'sgar_100'}
    assert set(line['metrics']) == want == set(line['oracle_metrics'])
    assert line['hits_equal_oracle'] is True
    for k in want:
        assert abs(line['metrics'][k] - line['oracle_metrics'][k]) <= 1e-5, (k, line['metrics'][k], line['oracle_metrics'][k])


def test_run_c4_without_data_says_what_it_needs():
    env = {k: v for k, v in os.environ.items() if k not in ('SGA_3RSCAN_ROOT', 'SGA_CHECKPOINT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_c4.py')], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and 'SGA_3RSCAN_ROOT' in (r.stderr + r.stdout)
