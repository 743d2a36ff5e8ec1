#!/bin/bash
# usage: tools/pmc_sweep_variant.sh <variant tag | ""> <pairs> <objects>  -- FETCH_SIZE / WRITE_SIZE per launch of sweep16_kernel<3,true|false> in tools/bench_sweep.py
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; B=$2; N=$3
lib=""; if [ -n "$tag" ]; then lib="variants/libsga_$tag.so"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcv_${tag}_$c
  SGA_LIB_PATH=$lib timeout 900 rocprofv3 --pmc $c --kernel-include-regex 'sweep16_kernel<3, (true|false)' --output-format csv -d gpurun_out/pmcv_${tag}_$c -- python tools/bench_sweep.py $B $N 2 > /dev/null 2>&1
  python - "$tag" "$c" <<'PY'
import csv, glob, sys, collections
tag, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f'gpurun_out/pmcv_{tag}_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c:
            k = 'grad' if 'true' in r['Kernel_Name'] else 'sums'
            acc[k] += float(r['Counter_Value']); n[k] += 1
for k in acc: print(f'{tag or "current"} {k} {c} avg KiB/launch {acc[k]/n[k]:.0f} ({n[k]} launches)')
PY
done
