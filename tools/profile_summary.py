"""Per-step summary of a `rocprofv3 --kernel-trace --stats` kernel_stats.csv of bench.py.
usage: python tools/profile_summary.py <kernel_stats.csv> <steps in the process> [header line ...]"""
import csv
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r'nesie::(\w+)', name)
    if m:
        return 'nesie::' + m.group(1)
    if name.startswith('Cijk_') or 'rocblas' in name.lower() or 'Tensile' in name:
        return 'rocBLAS/Tensile fp32-MFMA GEMM (torch.bmm/matmul)'
    if 'multi_tensor' in name or 'fused_adam' in name.lower() or 'FusedAdam' in name or 'lpnorm' in name.lower():
        return 'ATen multi_tensor / fused AdamW / clip'
    if 'reduce_kernel' in name or 'Reduce' in name:
        return 'ATen reduce'
    if 'CatArrayBatchedCopy' in name:
        return 'ATen cat'
    if 'elementwise' in name or 'vectorized' in name:
        return 'ATen elementwise'
    if name.startswith('__amd_rocclr'):
        return 'runtime copy / fill'
    return 'ATen other'


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    fam = defaultdict(lambda: [0, 0.0])
    for r in rows:
        f = fam[family(r['Name'])]
        f[0] += int(r['Calls'])
        f[1] += float(r['TotalDurationNs'])
    total = sum(v[1] for v in fam.values())
    for line in sys.argv[3:]:
        print(line)
    print(f"{'kernel family':58s} {'calls/step':>10s} {'ms/step':>11s} {'share':>9s}")
    for k, (c, ns) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:58s} {c / steps:10.1f} {ns / steps / 1e6:11.3f} {100 * ns / total:8.1f}%')
    print(f"{'TOTAL (sum of kernel durations)':58s} {sum(v[0] for v in fam.values()) / steps:10.1f} {total / steps / 1e6:11.3f}")
    print('\ntop 25 kernels by total time: name | calls | avg us | total ms/step')
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:25]:
        print(f"{r['Name'][:110]:110s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} {float(r['TotalDurationNs']) / steps / 1e6:8.3f}")


if __name__ == '__main__':
    main()
