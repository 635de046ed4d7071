"""Instruction mix of the tile loop of pw_fwd_kernel instantiations (from the gfx950 assembly): on
gfx950 an fp32 MFMA holds its SIMD's issue, so the loop's matrix-pipe utilisation is bounded by
MFMA cycles / (MFMA cycles + issue cycles of every other instruction in the loop).
usage: python tools/isa_loop.py nesie_amd/csrc/pwconv_g3.hip [EPI ...]"""
import re
import subprocess
import sys
import tempfile
from collections import Counter


def main():
    path, epis = sys.argv[1], sys.argv[2:] or ['1', '65', '67', '129']
    with tempfile.TemporaryDirectory() as d:
        out = d + '/k.s'
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off',
                        '-std=c++17', '-S', '--cuda-device-only', path, '-o', out], check=True, capture_output=True)
        txt = open(out).read()
    for name in re.findall(r'^(_ZN5nesie13pw_fwd_kernel\w+):', txt, re.M):
        m = re.match(r'_ZN5nesie13pw_fwd_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E', name)
        if m.group(6) not in epis or m.group(8) != '0':
            continue
        body = re.search(r'^%s:[^\n]*\n(.*?)s_endpgm' % re.escape(name), txt, re.S | re.M).group(1).splitlines()
        bar = [i for i, l in enumerate(body) if 's_barrier' in l][-1]
        start = max(i for i, l in enumerate(body[:bar]) if re.match(r'^\.LBB\d+_\d+:', l))
        lab = body[start].split(':')[0]
        end = max(i for i, l in enumerate(body) if re.search(r's_cbranch_\w+ ' + re.escape(lab) + r'\b', l))
        loop = [l.strip() for l in body[start:end + 1]
                if l.strip() and not l.strip().startswith((';', '.'))]
        c = Counter()
        for l in loop:
            op = l.split()[0]
            kind = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else
                    'waitcnt' if op.startswith('s_waitcnt') else 'salu' if op.startswith('s_') else
                    'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_')) else op)
            c[kind] += 1
        other = sum(v for k, v in c.items() if k != 'mfma')
        print('<%s> loop: %d instructions: %s; %.2f other instructions per MFMA' % (
            ','.join(m.groups()), len(loop), dict(c), other / max(c['mfma'], 1)))
        if '-v' in sys.argv:
            print('   VALU:', Counter(l.split()[0] for l in loop if l.startswith('v_') and not l.startswith('v_mfma')).most_common(10))
            print('   SALU:', Counter(l.split()[0] for l in loop if l.startswith('s_')).most_common(10))


if __name__ == '__main__':
    main()
