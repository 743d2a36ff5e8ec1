#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_k_bench_driver_flags.json 2> gpurun_out/r03_k_bench_driver_flags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_k_bench_driver_flags.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('speedup_vs_cpu_baseline'), d.get('hits_at_1',{}).get('hip'), d.get('extra_bf16x3',{}).get('value'))
PY
