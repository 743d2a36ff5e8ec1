#!/bin/bash
# usage: tools/exp_sweeph.sh <variant tags...>  -- time the split-fp16 loss sweeps of each variants/libsga_<tag>.so (c2 and a c3-sized shard)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  echo "== $t c2: $(SGA_MFMA_MODE=f16x2 SGA_LIB_PATH=variants/libsga_$t.so python tools/bench_sweep.py 512 64 6 2>&1 | tail -1)"
done
