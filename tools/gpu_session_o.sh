#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
echo "== bf16x3 c2: $(SGA_MFMA_MODE=bf16x3 python tools/bench_sweep.py 512 64 8 2>&1 | tail -1)"
echo "== bf16x3 c3/8: $(SGA_MFMA_MODE=bf16x3 python tools/bench_sweep.py 512 128 3 2>&1 | tail -1)"
echo "== bf16x3 c3: $(SGA_MFMA_MODE=bf16x3 python tools/bench_sweep.py 4096 128 1 2>&1 | tail -1)"
