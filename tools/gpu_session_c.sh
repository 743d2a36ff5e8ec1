#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_onepass_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py tests/test_fullsize_gpu.py tests/test_modules_gpu.py tests/test_bf16x3_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -3
for t in head hip; do
  lib=variants/libsga_$t.so; [ $t == hip ] && lib=sgaligner_amd/csrc/libsga_hip.so
  echo "== bf16x3 $t c2: $(SGA_MFMA_MODE=bf16x3 SGA_LIB_PATH=$lib python tools/bench_sweep.py 512 64 8 2>&1 | tail -1)"
  echo "== bf16x3 $t c3/8: $(SGA_MFMA_MODE=bf16x3 SGA_LIB_PATH=$lib python tools/bench_sweep.py 512 128 3 2>&1 | tail -1)"
done
for lib in sgaligner_amd/csrc/libsga_hip.so variants/libsga_noslp.so; do
SGA_LIB_PATH=$lib python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-bf16x3 > gpurun_out/c3_quick.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/c3_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], [ (r['kernel'][:30], r['step_ms']) for r in d.get('roofline_other',[])])
PY
done
