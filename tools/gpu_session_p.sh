#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
tools/profile_round3.sh r03_m > gpurun_out/r03_m_profile.log 2>&1
tail -c 1500 gpurun_out/r03_m_profile.log
