#!/bin/bash
# round-2 closing measurements: bench lines at c2 / c3 (with the bf16x3 extra key), kernel stats of the opt-in mode, inference bench
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --config c2 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err
python bench.py --config c3 --steps 5 --warmup 2 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err
SGA_MFMA_MODE=bf16x3 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bf16x3_c2_stats -- python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-hits --no-bf16x3 --no-attr < /dev/null > gpurun_out/${tag}_bf16x3_c2_under_rocprof.json 2> /dev/null
python tools/prof_summary.py gpurun_out/${tag}_bf16x3_c2_stats gpurun_out/${tag}_bf16x3_c2_kernel_stats.csv > /dev/null
python tools/bench_eval.py > gpurun_out/${tag}_eval_c2.json 2> /dev/null
python tools/bench_eval.py 4096 128 > gpurun_out/${tag}_eval_c3.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_eval_stats -- python tools/bench_eval.py < /dev/null > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/${tag}_eval_stats gpurun_out/${tag}_eval_c2_kernel_stats.csv > /dev/null
cat gpurun_out/${tag}_bench_c2.json gpurun_out/${tag}_bench_c3.json gpurun_out/${tag}_eval_c2.json gpurun_out/${tag}_eval_c3.json | cut -c1-1200
