"""Prefix a tools/pmc_kernel.sh counter dump (profiles/<tag>_sq_counters.txt) with the derived per-kernel utilisation table.
  python tools/sq_summary.py <tag>      # reads profiles/<tag>_sq_counters.txt and profiles/<tag>_c2_kernel_stats.csv"""
import collections, re, sys
tag = sys.argv[1]
dur = {}
for l in open(f'profiles/{tag}_c2_kernel_stats.csv'):
    l = l.rstrip('\n')
    if l.startswith('#') or l.startswith('Name'):
        continue
    parts = l.rsplit(',', 7)
    if len(parts) == 8:
        dur[parts[0].strip('"')] = float(parts[3]) / 1e6
cnt = collections.defaultdict(dict)
path = f'profiles/{tag}_sq_counters.txt'
lines = open(path).read().split('\n')
for l in lines:
    m = re.match(r'(.*?) \| (\w+) \| per-launch ([\d.e+]+)', l)
    if m:
        cnt[m.group(1).strip()][m.group(2)] = float(m.group(3))
hdr = [f'# rocprofv3 --pmc passes (one counter set per pass, counters only) over `python bench.py --config c2 --steps 2 --warmup 1` with the',
       f'# opt-in bf16x3 extra measurement on (tools/profile_round.sh {tag} -> tools/pmc_kernel.sh), summed over the chip, averaged per launch.',
       f'# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock); duration from profiles/{tag}_c2_kernel_stats.csv (same command',
       '# under --kernel-trace --stats).  SQ_INSTS_VALU counts MFMA instructions too: "other VALU" = SQ_INSTS_VALU - SQ_INSTS_MFMA.',
       '# kernel | duration ms | MFMA busy @2.4GHz | @2.2GHz | MFMA insts | other VALU insts | LDS insts | LDS bank-conflict cycles | conflict cycles / (256 CUs x duration @2.2GHz)']
for k, c in cnt.items():
    d = None
    for n, v in dur.items():
        if n.startswith(k[:60]):
            d = v
    if d is None:
        continue
    busy = c['SQ_VALU_MFMA_BUSY_CYCLES']
    hdr.append(f"# {k[:58]} | {d:.2f} | {busy / (1024 * d * 1e-3 * 2.4e9):.3f} | {busy / (1024 * d * 1e-3 * 2.2e9):.3f} | {c['SQ_INSTS_MFMA']:.3e} | "
               f"{c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']:.3e} | {c['SQ_INSTS_LDS']:.3e} | {c['SQ_LDS_BANK_CONFLICT']:.3e} | "
               f"{c['SQ_LDS_BANK_CONFLICT'] / (256 * d * 1e-3 * 2.2e9):.3f}")
open(path, 'w').write('\n'.join(hdr + [l for l in lines if l and not l.startswith('#')]) + '\n')
print('\n'.join(hdr))
