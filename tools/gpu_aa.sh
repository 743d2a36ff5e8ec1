#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_aa.sh <variant tag | hip> ...   -- tools/bench_aa.py (anchors x anchors kernel + stash GEMMs, one
# 2048 x 155 648 block, ordered and symmetric) with variants/libsga_<tag>.so (tools/build_variant.sh) or the in-tree library ("hip")
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  lib=variants/libsga_$t.so; [ $t == hip ] && lib=sgaligner_amd/csrc/libsga_hip.so
  echo "== $t: $(SGA_LIB_PATH=$lib python tools/bench_aa.py 155648 2048 3 3 2>&1 | tail -2 | tr '\n' ' ')"
done
