#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python tools/dbg/aa_sym_check.py 2100 512 3
python tools/dbg/aa_sym_check.py 1000 96 2
python tools/dbg/aa_sym_check.py 4099 1024 3
python tools/bench_aa.py 155648 2048 3 3 2>&1 | tail -2
python tools/bench_aa.py 19456 2048 3 3 2>&1 | tail -2
