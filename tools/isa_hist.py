"""Instruction histogram of the large basic blocks of one kernel in a hipcc -S listing (what the loop body is made of).
  python tools/isa_hist.py <file.s> <kernel-name-substring> [min block size]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
blocks, cur = [], ('entry', [])
for l in lines[start + 1:end]:
    l = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = (l, [])
    elif l and not l.startswith(';') and not l.startswith('.'):
        cur[1].append(l)
blocks.append(cur)
for name, ins in blocks:
    if len(ins) < minsz:
        continue
    c = collections.Counter(i.split()[0] for i in ins)
    groups = collections.Counter()
    for k, v in c.items():
        if 'mfma' in k: g = 'mfma'
        elif k.startswith('v_accvgpr'): g = 'accvgpr'
        elif k.startswith('v_exp'): g = 'v_exp'
        elif k.startswith('v_'): g = 'valu_other'
        elif k.startswith('s_waitcnt'): g = 's_waitcnt'
        elif k.startswith('s_nop'): g = 's_nop'
        elif k.startswith('s_'): g = 'salu'
        else: g = k
        groups[g] += v
    print(name, len(ins), 'instructions', dict(groups))
    vo = collections.Counter({k: v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k})
    print('   VALU:', vo.most_common(30))
