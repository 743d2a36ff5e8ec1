#!/bin/bash
# Round-5 measurements in one GPU-box call:  tools/profile_round5.sh <tag>
#   default bench line with the driver's flags (configs[2] headline in the default arithmetic + extras), kernel-trace stats at c3 and c2.
#   (HBM-traffic PMC passes: tools/pmc_traffic.sh -> profiles/r05_pmc_traffic.csv, run separately: counters only.)
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2> gpurun_out/${tag}_bench_driver_flags.err
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c3_stats -- python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct < /dev/null > gpurun_out/${tag}_c3_bench_under_rocprof.json 2> gpurun_out/${tag}_c3_stats.err
python tools/prof_summary.py gpurun_out/${tag}_c3_stats gpurun_out/${tag}_c3_kernel_stats.csv > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c2_stats -- python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-hits --no-pct < /dev/null > gpurun_out/${tag}_c2_bench_under_rocprof.json 2> gpurun_out/${tag}_c2_stats.err
python tools/prof_summary.py gpurun_out/${tag}_c2_stats gpurun_out/${tag}_c2_kernel_stats.csv > /dev/null
rm -rf gpurun_out/${tag}_c3_stats gpurun_out/${tag}_c2_stats
head -c 1500 gpurun_out/${tag}_bench_driver_flags.json; echo; head -8 gpurun_out/${tag}_c3_kernel_stats.csv | cut -c1-200
