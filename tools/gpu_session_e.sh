#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  lib=variants/libsga_$t.so; [ $t == hip ] && lib=sgaligner_amd/csrc/libsga_hip.so
  echo "== $t: $(SGA_LIB_PATH=$lib python tools/bench_aa.py 155648 2048 3 3 2>&1 | tail -1)"
done
