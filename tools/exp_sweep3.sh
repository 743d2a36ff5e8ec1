#!/bin/bash
# usage: tools/exp_sweep3.sh <variant tags...>  -- time the default (three-plane) loss sweeps of each tools/dbg/libsga_<tag>.so
# (built by tools/dbg/build_variant_lib.sh <tag> -D...) at a configs[2]-sized shard (1024 x 128) and at configs[1]; prints the gradient checksum.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  lib=tools/dbg/libsga_$t.so
  [ "$t" = default ] && lib=sgaligner_amd/csrc/libsga_hip.so
  for rep in 1 2; do
    echo "== $t 1024x128: $(SGA_LIB_PATH=$lib timeout 300 python tools/bench_sweep.py 1024 128 4 2>&1 | tail -1)"
  done
  echo "== $t c2: $(SGA_LIB_PATH=$lib timeout 300 python tools/bench_sweep.py 512 64 8 2>&1 | tail -1)"
done
