#!/bin/bash
# usage: tools/exp_sweeph.sh <variant tags...>  -- time the split-fp16 loss sweeps of each variants/libsga_<tag>.so (c2 and a c3-sized shard;
# with and without the coefficients' lo terms)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  for lo in 1 0; do
  echo "== $t coef_lo=$lo c2: $(SGA_F16X2_COEF_LO=$lo SGA_MFMA_MODE=f16x2 SGA_LIB_PATH=variants/libsga_$t.so python tools/bench_sweep.py 512 64 6 2>&1 | tail -1)"
  echo "== $t coef_lo=$lo c3/8: $(SGA_F16X2_COEF_LO=$lo SGA_MFMA_MODE=f16x2 SGA_LIB_PATH=variants/libsga_$t.so python tools/bench_sweep.py 512 128 3 2>&1 | tail -1)"
  done
done
