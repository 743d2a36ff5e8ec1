#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-regex> <tag> "<counter set 1>" "<counter set 2>" ...   (env SGA_PMC_CONFIG=c2|c3, default c2)
# one rocprofv3 --pmc pass per counter set (counters only, no trace domains), summarised per kernel on stdout as
#   kernel | counter | per-launch average | launches
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
re="$1"; tag="$2"; shift 2
cfg=${SGA_PMC_CONFIG:-c2}
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-include-regex "$re" --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-pct < /dev/null > gpurun_out/pmc_${tag}_$i.log 2>&1
  python - "$tag" "$i" <<'PY'
import csv, glob, sys, collections
tag, i = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(f'gpurun_out/pmc_{tag}_{i}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k in acc:
    for c, v in acc[k].items():
        print(f'{k} | {c} | per-launch {v / max(1, n[(k, c)]):.6g} | launches {n[(k, c)]}')
PY
done
