#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for t in head pin0 pin1 pin2; do
  echo "== $t: $(SGA_LIB_PATH=variants/libsga_$t.so python tools/bench_aa.py 19456 4096 3 5 2>&1 | tail -1)"
  echo "== $t: $(SGA_LIB_PATH=variants/libsga_$t.so python tools/bench_aa.py 155648 2048 3 3 2>&1 | tail -1)"
done
