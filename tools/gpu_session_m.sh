#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_bf16x3_gpu.py -x -q 2>&1 | tail -2
for t in "$@"; do
  lib=variants/libsga_$t.so; [ $t == hip ] && lib=sgaligner_amd/csrc/libsga_hip.so
  echo "== bf16x3 $t c2: $(SGA_MFMA_MODE=bf16x3 SGA_BENCH_SWEEP_COMPARE=1 SGA_LIB_PATH=$lib python tools/bench_sweep.py 512 64 8 2>&1 | tail -6 | tr '\n' ' ')"
  echo "== bf16x3 $t c3/8: $(SGA_MFMA_MODE=bf16x3 SGA_LIB_PATH=$lib python tools/bench_sweep.py 512 128 3 2>&1 | tail -1)"
done
