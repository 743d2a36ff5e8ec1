"""print the headline numbers and roofline objects of a bench.py JSON line:  python tools/dbg/show_bench.py <file>"""
import json, sys
d = None
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d.get('hits_at_1', {}).get('gpu'), d.get('hits_at_1', {}).get('oracle'))
for r in [d['roofline']] + d.get('roofline_other', []):
    print('  ', r['kernel'][:70], r.get('avg_launch_ms') or r.get('step_ms'), r['frac'], r.get('traffic'))
