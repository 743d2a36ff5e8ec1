#!/bin/bash
# Everything the round's profiles/ entries come from, in one GPU-box call:  tools/profile_round.sh <tag>   (e.g. r02_b)
#   kernel-trace stats of bench.py at c2 and c3, HBM-traffic PMC passes, SQ (MFMA-utilisation) PMC passes.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in c2 c3; do
  st=10; wu=3; if [ $cfg = c3 ]; then st=3; wu=1; fi
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${cfg}_stats -- python bench.py --config $cfg --steps $st --warmup $wu --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct < /dev/null > gpurun_out/${tag}_${cfg}_bench_under_rocprof.json 2> gpurun_out/${tag}_${cfg}_stats.err
  python tools/prof_summary.py gpurun_out/${tag}_${cfg}_stats gpurun_out/${tag}_${cfg}_kernel_stats.csv > /dev/null
done
rm -f gpurun_out/${tag}_pmc_traffic.csv
tools/pmc_traffic.sh c2 gpurun_out/${tag}_pmc_traffic.csv > /dev/null
tools/pmc_traffic.sh c3 gpurun_out/${tag}_pmc_traffic.csv > /dev/null
# MFMA utilisation: busy cycles of the MFMA pipe vs the SQ's busy / wave cycles, instruction mix
re='sweep.*_kernel<3, true>|pointnet_fwd_kernel|pointnet_bwd_fused_kernel|sweep.*_kernel<3, false>'
tools/pmc_kernel.sh "$re" ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" > gpurun_out/${tag}_sq_counters.txt 2>&1
cat gpurun_out/${tag}_sq_counters.txt | head -60
