#!/bin/bash
# A x A kernel check + fresh PCT-step profile + quick headline
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_onepass_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_i_pct_stats -- python tools/bench_small.py 4 40 pct,gat,rel,attr < /dev/null > gpurun_out/r03_i_pct_small.txt 2>&1
python tools/prof_summary.py gpurun_out/r03_i_pct_stats gpurun_out/r03_i_pct_small_kernel_stats.csv > /dev/null
cat gpurun_out/r03_i_pct_small.txt | tail -2
python tools/bench_small.py 4 40 pct,gat,rel,attr
python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-bf16x3 | cut -c1-1500
SGA_STASH_BYTES=$((40*1024*1024*1024)) python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-bf16x3 | cut -c1-400
