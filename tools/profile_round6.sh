#!/bin/bash
# Round-6 measurements in one GPU-box call:  tools/profile_round6.sh <tag>
#   1. HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, counters only) at configs[2], configs[1] and configs[4] (the fp16 tile core) -> profiles/r06_pmc_traffic.csv ON THE BOX, so that
#      the bench line of step 2 carries `traffic`; the file comes back as gpurun_out/<tag>_pmc_traffic.csv
#   2. the default bench line with the driver's flags (the compact line on stdout + bench_extras.json)
#   3. rocprofv3 kernel-trace stats at configs[2] (default arithmetic only) and configs[1]
#   4. SQ counters (MFMA busy, instruction mix, LDS) of the two sweeps and the PointNet forward at configs[1]
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/${tag}_pmc_traffic.csv
bash tools/pmc_traffic.sh c3 gpurun_out/${tag}_pmc_traffic.csv --no-exact > gpurun_out/${tag}_pmc_c3.log 2>&1
bash tools/pmc_traffic.sh c2 gpurun_out/${tag}_pmc_traffic.csv --no-exact > gpurun_out/${tag}_pmc_c2.log 2>&1
bash tools/pmc_traffic_c5.sh gpurun_out/${tag}_pmc_traffic.csv > gpurun_out/${tag}_pmc_c5.log 2>&1
cp gpurun_out/${tag}_pmc_traffic.csv profiles/r06_pmc_traffic.csv
rm -rf gpurun_out/pmc_t_*
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2> gpurun_out/${tag}_bench_driver_flags.err
cp bench_extras.json gpurun_out/${tag}_bench_extras.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c3_stats -- python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact < /dev/null > gpurun_out/${tag}_c3_bench_under_rocprof.json 2> gpurun_out/${tag}_c3_stats.err
python tools/prof_summary.py gpurun_out/${tag}_c3_stats gpurun_out/${tag}_c3_default_only_kernel_stats.csv > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_c2_stats -- python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-hits --no-pct --no-exact < /dev/null > gpurun_out/${tag}_c2_bench_under_rocprof.json 2> gpurun_out/${tag}_c2_stats.err
python tools/prof_summary.py gpurun_out/${tag}_c2_stats gpurun_out/${tag}_c2_kernel_stats.csv > /dev/null
rm -rf gpurun_out/${tag}_c3_stats gpurun_out/${tag}_c2_stats
SGA_PMC_CONFIG=c2 bash tools/pmc_kernel.sh 'sweep3_kernel<3|pointnet_fwd_p3_kernel' ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" > gpurun_out/${tag}_sq_counters.txt 2>&1
rm -rf gpurun_out/pmc_${tag}_sq_*
head -c 1200 gpurun_out/${tag}_bench_driver_flags.json; echo; cat gpurun_out/${tag}_pmc_traffic.csv | tail -12; head -12 gpurun_out/${tag}_c3_default_only_kernel_stats.csv | cut -c1-160
