#!/bin/bash
# HBM traffic of the two dominant kernels: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over
# `python bench.py --config <c2|c3> --steps 2 --warmup 1` (counters only, no trace domains).
# usage: tools/pmc_traffic.sh <c2|c3> <out.csv>   (rows are APPENDED; bench.py reads profiles/r06_pmc_traffic.csv)
# row: kernel;workload_key;sha16(kernel source);counter;avg KiB per dispatch;launches  -- bench.py uses a row only when the
# workload key matches what it runs and the source file is unchanged since the pass.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
cfg=$1
out=$2
EXTRA=${3:-}      # e.g. --no-exact to leave the fp32-MFMA extra out of the pass
if [ ! -f "$out" ]; then
  echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-include-regex 'sweep.*_kernel<3, (true|false)|pointnet_fwd_kernel'), bench.py --config <cfg> --steps 2 --warmup 1" > $out
  echo "# unit: KiB per dispatch (average over the profiled launches); gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) -> bytes = (2*FETCH + WRITE) * 1024" >> $out
  echo "kernel;workload_key;source_sha16;counter;avg_kib_per_launch;launches" >> $out
fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-include-regex 'sweep.*_kernel<3, (true|false)|pointnet_fwd(_p3)?_kernel' --output-format csv -d gpurun_out/pmc_t_${cfg}_$c -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct $EXTRA < /dev/null > gpurun_out/pmc_t_${cfg}_$c.log 2>&1
  python - $c $cfg >> $out <<'PY'
import csv, glob, sys, collections, hashlib
c, cfg = sys.argv[1], sys.argv[2]
sha = lambda f: hashlib.sha256(open('sgaligner_amd/csrc/' + f, 'rb').read()).hexdigest()[:16]
keys = {'c2': ('T=65536,P=512', 'ns=9728,A=9728,J=46080'), 'c3': ('T=1048576,P=512', 'ns=155648,A=155648,J=737280')}[cfg]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f'gpurun_out/pmc_t_{cfg}_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c:
            continue
        name = r['Kernel_Name']
        if 'pointnet_fwd' in name:
            k = ('pointnet_fwd_p3_kernel' if 'pointnet_fwd_p3' in name else 'pointnet_fwd_kernel', keys[0], sha('pointnet.hip'))
        else:
            import re
            mm = re.search(r'(\w+_kernel)<3, (true|false)', name)
            src = 'sweep3.hip' if mm.group(1) == 'sweep3_kernel' else 'contrastive.hip'
            k = (mm.group(1) + '<3,' + mm.group(2) + '>', keys[1], sha(src))
        acc[k] += float(r['Counter_Value']); n[k] += 1
for k in sorted(acc):
    print(f'{k[0]};{k[1]};{k[2]};{c};{acc[k] / n[k]:.4f};{n[k]}')
PY
done
cat $out
