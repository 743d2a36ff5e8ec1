#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/fuzz_r03.py 100 11 2>&1 | tail -1
SGA_FUZZ_NOSYM=1 timeout 600 python tools/fuzz_r03.py 100 11 2>&1 | tail -1
