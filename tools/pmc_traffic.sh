#!/bin/bash
# HBM traffic of the two dominant kernels: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over
# `python bench.py --steps 2 --warmup 1` (counters only, no trace domains).  usage: tools/pmc_traffic.sh <out.csv>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=$1
echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-include-regex 'sweep16_kernel<3, true>|pointnet_fwd_kernel'), bench.py --steps 2 --warmup 1" > $out
echo "# unit: KiB per dispatch (average over the profiled launches); gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) -> bytes = (2*FETCH + WRITE) * 1024" >> $out
echo "kernel,counter,avg_kib_per_launch,launches" >> $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex 'sweep16_kernel<3, true>|pointnet_fwd_kernel' --output-format csv -d gpurun_out/pmc_t_$c -- python bench.py --steps 2 --warmup 1 < /dev/null > gpurun_out/pmc_t_$c.log 2>&1
  python - $c >> $out <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f'gpurun_out/pmc_t_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c:
            continue
        k = 'pointnet_fwd_kernel' if 'pointnet_fwd' in r['Kernel_Name'] else 'sweep16_kernel<3,true>'
        acc[k] += float(r['Counter_Value']); n[k] += 1
for k in sorted(acc):
    print(f'{k},{c},{acc[k] / n[k]:.4f},{n[k]}')
PY
done
cat $out
