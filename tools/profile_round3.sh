#!/bin/bash
# Round-3 measurements in one GPU-box call:  tools/profile_round3.sh <tag>
#   default bench line (configs[2] headline), kernel-trace stats at c3 and c2, HBM-traffic PMC passes, SQ (MFMA-busy) passes.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2> gpurun_out/${tag}_bench_driver_flags.err
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c3_stats -- python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-bf16x3 < /dev/null > gpurun_out/${tag}_c3_bench_under_rocprof.json 2> gpurun_out/${tag}_c3_stats.err
python tools/prof_summary.py gpurun_out/${tag}_c3_stats gpurun_out/${tag}_c3_kernel_stats.csv > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c2_stats -- python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-hits --no-attr --no-bf16x3 --no-pct < /dev/null > gpurun_out/${tag}_c2_bench_under_rocprof.json 2> gpurun_out/${tag}_c2_stats.err
python tools/prof_summary.py gpurun_out/${tag}_c2_stats gpurun_out/${tag}_c2_kernel_stats.csv > /dev/null
rm -f gpurun_out/${tag}_pmc_traffic.csv
tools/pmc_traffic.sh c2 gpurun_out/${tag}_pmc_traffic.csv > /dev/null
tools/pmc_traffic.sh c3 gpurun_out/${tag}_pmc_traffic.csv > /dev/null
re='sweep.*_kernel<3, true|pointnet_fwd_kernel|pointnet_bwd_fused_kernel|sweep.*_kernel<3, false|anchor_multi'
tools/pmc_kernel.sh "$re" ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" > gpurun_out/${tag}_sq_counters.txt 2>&1
head -c 3000 gpurun_out/${tag}_bench_driver_flags.json; echo; cat gpurun_out/${tag}_pmc_traffic.csv; head -40 gpurun_out/${tag}_sq_counters.txt
