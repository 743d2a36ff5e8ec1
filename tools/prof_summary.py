"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text summary for profiles/."""
import csv, glob, os, sys
d = sys.argv[1]
out = sys.argv[2]
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: -float(r.get('TotalDurationNs', r.get('TotalDuration(ns)', 0)) or 0))
with open(out, 'w') as fo:
    fo.write('# rocprofv3 --kernel-trace --stats summary (top kernels by total time)\n')
    if rows:
        keys = list(rows[0].keys())
        fo.write(','.join(keys) + '\n')
        for r in rows[:40]:
            fo.write(','.join(str(r[k]) for k in keys) + '\n')
print(open(out).read()[:6000])
