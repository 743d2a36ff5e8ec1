#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_bf16x3_gpu.py tests/test_loss_gpu.py -x -q 2>&1 | tail -2
echo "== bf16x3 c2: $(SGA_MFMA_MODE=bf16x3 python tools/bench_sweep.py 512 64 8 2>&1 | tail -1)"
python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-hits --no-pct --no-attr > gpurun_out/c2_quick.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/c2_quick.json').read().strip().splitlines()[-1])
print('c2', d['value'], d['ms_per_step'], 'bf16x3:', {k:v for k,v in d.get('extra_bf16x3',{}).items() if k in ('value','ms_per_step','max_grad_err_rel_to_own_max')})
PY
python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct > gpurun_out/c3_quick.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/c3_quick.json').read().strip().splitlines()[-1])
print('c3', d['value'], d['ms_per_step'], 'bf16x3:', {k:v for k,v in d.get('extra_bf16x3',{}).items() if k in ('value','ms_per_step','max_grad_err_rel_to_own_max','f32_rerun_err_rel_to_own_max')})
PY
