#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 700 python tools/fuzz_parity.py 600 41 2>&1 | tail -1
timeout 700 python tools/fuzz_r03.py 600 42 2>&1 | tail -1
timeout 400 python tools/fuzz_groups.py 240 43 2>&1 | tail -1
