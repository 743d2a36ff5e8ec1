"""Per-kernel resource report of the HIP sources (registers, LDS, scratch, spills) and what sits INSIDE loops that should not: scalar
loads (a dynamically indexed kernel argument), scratch traffic (spills), v_readlane (spilled SGPRs).  Cross-compiles, needs no GPU.
  python tools/kernel_resources.py [file.hip ...]        # default: every csrc/*.hip; prints only kernels with something to report
  python tools/kernel_resources.py --all contrastive.hip # every kernel of the file
This is the audit of DESIGN.md 3a: a kernel on the hot path must report scratch 0 and no scalar loads in its tile loop."""
import concurrent.futures as cf
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgaligner_amd import _build

REMARK = re.compile(r'Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?'
                    r'Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)', re.S)


def demangle(names):
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
            return {n: re.sub(r'\(anonymous namespace\)::', '', o).split('(')[0].replace('void ', '') for n, o in zip(names, out)}
        except OSError:
            continue
    return {n: n for n in names}


def analyse(src):
    base = os.path.basename(src)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'k.s')
        cmd = [_build.HIPCC] + _build.FLAGS + _build.FILE_FLAGS.get(base, []) + ['-S', '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', src, '-o', asm]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        res = {m.group(1): dict(zip(('sgpr', 'vgpr', 'agpr', 'scratch', 'occ', 'sspill', 'vspill', 'lds'), map(int, m.groups()[1:]))) for m in REMARK.finditer(r.stderr)}
        kern, depth = None, 0
        for line in open(asm):
            m = re.match(r'^(_Z\S+):', line)
            if m:
                kern, depth = m.group(1), 0
                continue
            if line.startswith('.Lfunc_end'):
                kern = None
            if kern not in res:
                continue
            if re.match(r'^(\.LBB\S+:|; %bb\.\d+:)', line):
                md = re.search(r'Depth[= ](\d+)', line)
                depth = int(md.group(1)) if md else 0
                continue
            op = line.split()[0] if line.strip() and not line.strip().startswith(';') else ''
            if depth > 0:
                for key, pre in (('loop_s_load', 's_load'), ('loop_scratch', 'scratch_'), ('loop_readlane', 'v_readlane')):
                    if op.startswith(pre):
                        res[kern][key] = res[kern].get(key, 0) + 1
    return base, res


def main(argv):
    show_all = '--all' in argv
    files = [a for a in argv if not a.startswith('--')] or sorted(glob.glob(os.path.join(_build.CSRC, '*.hip')))
    files = [f if os.path.exists(f) else os.path.join(_build.CSRC, f) for f in files]
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(analyse, files))
    for base, res in results:
        names = demangle(list(res))
        for k, v in res.items():
            flagged = v['scratch'] or v['sspill'] or v['vspill'] or v.get('loop_scratch') or v.get('loop_readlane')
            if show_all or flagged:
                extra = ' '.join(f'{kk}={v[kk]}' for kk in ('loop_s_load', 'loop_scratch', 'loop_readlane') if v.get(kk))
                print(f'{base:16s} {names[k][:72]:72s} VGPR {v["vgpr"]:3d} AGPR {v["agpr"]:3d} SGPR {v["sgpr"]:3d} occ {v["occ"]} LDS {v["lds"]:6d} '
                      f'scratch {v["scratch"]:4d} spills v{v["vspill"]}/s{v["sspill"]} {extra}')
    return results


if __name__ == '__main__':
    main(sys.argv[1:])
