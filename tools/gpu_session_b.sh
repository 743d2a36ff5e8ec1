#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_onepass_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py tests/test_fullsize_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3
tools/exp_sweep.sh head new noslp
