#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=r03_j
rm -f gpurun_out/${tag}_pmc_traffic.csv
tools/pmc_traffic.sh c2 gpurun_out/${tag}_pmc_traffic.csv > /dev/null
tools/pmc_traffic.sh c3 gpurun_out/${tag}_pmc_traffic.csv > /dev/null
re='sweep.*_kernel<3, true|pointnet_fwd_kernel|pointnet_bwd_fused_kernel|sweep.*_kernel<3, false|anchor_multi'
tools/pmc_kernel.sh "$re" ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" > gpurun_out/${tag}_sq_counters.txt 2>&1
cat gpurun_out/${tag}_pmc_traffic.csv; grep "sweep16" gpurun_out/${tag}_sq_counters.txt
