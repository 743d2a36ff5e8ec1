#!/bin/bash
# SQ counters of the PCT step's top kernels (tools/bench_small.py, reference-default module list)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
re='gemm_nt_kernel|gemm_tn_kernel|attn_bwd_dq_kernel|attn_apply_kernel|attn_stats_kernel|head_dw_kernel|head_scatter_kernel|segment_max_affine|bn_apply_kernel|colsum_kernel'
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "$re" --output-format csv -d gpurun_out/pmc_pct_$i -- python tools/bench_small.py 4 40 pct,gat,rel,attr < /dev/null > gpurun_out/pmc_pct_$i.log 2>&1
  python - "$i" <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(f'gpurun_out/pmc_pct_{i}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    for c, v in acc[k].items():
        print(f'{k} | {c} | total {v:.6g} | per-launch {v / max(1, n[(k, c)]):.6g} | launches {n[(k, c)]}')
PY
done
