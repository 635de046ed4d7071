"""Register / LDS / occupancy table of the layer-kernel instantiations (pw_fwd_kernel, pw_wgrad_kernel):
compiles each translation unit to gfx950 assembly and reads the kernel descriptors' metadata.
    .vgpr_count -> allocation (granule 8) -> waves per SIMD (MI355X_MICROARCH.md, register files)
    workgroups per CU = min(waves-per-SIMD * 4 / waves-per-workgroup, 160 KB / LDS per workgroup)
usage: python tools/isa_regs.py [-j N] [-D MACRO ...] [file.hip ...]   (default: the pwconv units)"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nesie_amd', 'csrc')


def waves_per_simd(vgprs):
    alloc = -(-max(vgprs, 1) // 8) * 8
    return min(8, 512 // alloc), alloc


def scan(path, defines=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off',
                        '-std=c++17', '-S', '--cuda-device-only', *['-D' + m for m in defines], path,
                        '-o', out], check=True, capture_output=True)
        txt = open(out).read()
    rows = []
    for blk in txt.split('  - .agpr_count:')[1:]:
        def f(key):
            m = re.search(r'\.%s:\s+(\S+)' % key, blk)
            return m.group(1) if m else '0'
        name = f('name')
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        rows.append(dict(name=dem, vgpr=int(f('vgpr_count')), sgpr=int(f('sgpr_count')),
                         spill=int(f('vgpr_spill_count')), lds=int(f('group_segment_fixed_size')),
                         wg=int(f('max_flat_workgroup_size'))))
    return rows


def main():
    args = sys.argv[1:]
    jobs, defines, files = 8, [], []
    while args:
        a = args.pop(0)
        if a == '-j':
            jobs = int(args.pop(0))
        elif a == '-D':
            defines.append(args.pop(0))
        else:
            files.append(a)
    if not files:
        files = sorted(glob.glob(os.path.join(CSRC, 'pwconv_g*.hip'))) + [os.path.join(CSRC, 'pwconv_wgrad.hip')]
    with ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(lambda p: scan(p, defines), files))
    print(f"{'kernel':78s} vgpr alloc spill sgpr waves/SIMD  wg/CU(by registers, %d-thread wg)" % 512)
    for path, rows in zip(files, results):
        print('==', os.path.relpath(path, ROOT))
        for r in sorted(rows, key=lambda r: r['name']):
            if 'pw_fwd_kernel' not in r['name'] and 'pw_wgrad_kernel' not in r['name']:
                continue
            w, alloc = waves_per_simd(r['vgpr'])
            wpw = max(r['wg'] // 64, 1)
            short = re.sub(r'^void nesie::', '', r['name']).split('(')[0]
            print(f"{short:78s} {r['vgpr']:4d} {alloc:5d} {r['spill']:5d} {r['sgpr']:4d} {w:6d}      {w * 4 // wpw}")


if __name__ == '__main__':
    main()
