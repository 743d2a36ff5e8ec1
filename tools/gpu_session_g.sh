#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_onepass_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py tests/test_fullsize_gpu.py tests/test_modules_gpu.py tests/test_dist_gpu.py tests/test_drift_gpu.py -x -q 2>&1 | tail -3
python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-bf16x3 > gpurun_out/c3_quick.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/c3_quick.json').read().strip().splitlines()[-1])
print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['loss'], d['config']['peak_hbm_gib'], [ (r['kernel'][:30], r['step_ms']) for r in d.get('roofline_other',[])])
PY
python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-hits --no-pct --no-bf16x3 --no-attr > gpurun_out/c2_quick.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/c2_quick.json').read().strip().splitlines()[-1])
print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['loss'])
PY
timeout 600 python tools/fuzz_r03.py 150 11 2>&1 | tail -2
