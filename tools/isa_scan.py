"""Per-kernel count of global load / store widths and full waits in the gfx950 ISA of a .hip
file: scalar `global_load_dword` with a `s_waitcnt vmcnt(0)` after each is what a float4 path
guarded by run-time conditions can silently compile into (conv_wgrad before its interior-tile
fast path: 64 scalar loads, 65 waits).
usage: python tools/isa_scan.py nesie_amd/csrc/mlp.hip [more .hip files]"""
import os
import re
import subprocess
import sys
import tempfile


def scan(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC',
                        '-ffp-contract=off', '-std=c++17', '-S', '--cuda-device-only', path,
                        '-o', out], check=True, capture_output=True)
        txt = open(out).read()
    cur, stats = None, {}
    for line in txt.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            stats[cur] = dict(x4=0, x2=0, x1=0, s4=0, s1=0, w0=0, end=0)
            continue
        if cur is None:
            continue
        s = stats[cur]
        if 'global_load_dwordx4' in line: s['x4'] += 1
        elif 'global_load_dwordx2' in line: s['x2'] += 1
        elif re.search(r'global_load_dword\s', line): s['x1'] += 1
        if 'global_store_dwordx4' in line: s['s4'] += 1
        elif re.search(r'global_store_dword\s', line): s['s1'] += 1
        if 's_waitcnt vmcnt(0)' in line: s['w0'] += 1
        if 's_endpgm' in line: s['end'] += 1
    for k, s in stats.items():
        if s['end']:
            print(f"{k[:72]:72s} ld x4 {s['x4']:3d} x2 {s['x2']:3d} x1 {s['x1']:3d} | "
                  f"st x4 {s['s4']:3d} x1 {s['s1']:3d} | vmcnt(0) {s['w0']:3d}")


if __name__ == '__main__':
    for f in sys.argv[1:]:
        print('==', f)
        scan(f)
