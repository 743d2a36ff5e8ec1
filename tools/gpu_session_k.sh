#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_onepass_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -2
python tools/bench_aa.py 155648 2048 3 3 2>&1 | tail -2
python tools/bench_aa.py 19456 2048 4 3 2>&1 | tail -2
