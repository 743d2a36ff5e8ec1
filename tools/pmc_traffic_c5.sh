#!/bin/bash
# HBM traffic of the fp16 tile core at BASELINE configs[4]: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over
# `python bench.py --config c5 --steps 2 --warmup 1` (counters only, no trace domains), summed per STEP over every launch of a template.
# usage: tools/pmc_traffic_c5.sh <out.csv>   (rows are APPENDED in the format of tools/pmc_traffic.sh; bench.py reads profiles/r06_pmc_traffic.csv)
# rows: wide16_batch_kernel<0> = the forward sums, <1> = the coefficient pass of the negatives backward, <2> = every gradient GEMM (the
# negatives backward's eight launches AND the four stash-product launches of the anchors x anchors backward), <3> = the A x A similarity blocks.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=$1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-include-regex 'wide16_batch_kernel' --output-format csv -d gpurun_out/pmc_t_c5_$c -- python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact < /dev/null > gpurun_out/pmc_t_c5_$c.log 2>&1
  python - $c >> $out <<'PY'
import csv, glob, sys, collections, hashlib, re
c = sys.argv[1]
sha = hashlib.sha256(open('sgaligner_amd/csrc/wide16.hip', 'rb').read()).hexdigest()[:16]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f'gpurun_out/pmc_t_c5_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c:
            continue
        mm = re.search(r'wide16_batch_kernel<(\d)>', r['Kernel_Name'])
        if mm:
            k = f'wide16_batch_kernel<{mm.group(1)}>'
            acc[k] += float(r['Counter_Value']); n[k] += 1
steps = 3        # 2 timed + 1 warm-up
for k in sorted(acc):
    print(f'{k};c5:64x256x2048 per step;{sha};{c};{acc[k] / steps:.4f};{n[k]}')
PY
done
rm -rf gpurun_out/pmc_t_c5_*
tail -8 $out
