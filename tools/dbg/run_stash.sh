cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for sb in "" 51539607552 103079215104; do
  SGA_STASH_BYTES=$sb timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact > gpurun_out/stash_$sb.json 2>/dev/null
  echo "stash [$sb]: $(python -c "
import json,sys
for l in open('gpurun_out/stash_$sb.json'):
    l=l.strip()
    if l.startswith('{'): d=json.loads(l)
print(d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gib'))
")"
done
