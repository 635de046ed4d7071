"""MFMA utilisation per kernel from one rocprofv3 PMC pass of bench.py (eager mode):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv ...
GRBM_GUI_ACTIVE is summed over the 8 XCDs (a 135 us kernel reads 2.85 M cycles = 8 x 0.148 ms at
2.4 GHz), so utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs):
the share of the chip's matrix-pipe cycles the kernel kept busy while it ran (cross-check: the
rocBLAS 256x128 tile reads 66 % here and 100 of 157 TFLOP/s = 64 % by the clock).
usage: python tools/mfma_util.py <counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        name = re.sub(r'\(.*', '', r['Kernel_Name'])[:70]
        key = (name, int(r['Grid_Size']))
        rows[key][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            calls[key] += 1
    out = []
    for key, c in rows.items():
        busy, act = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('GRBM_GUI_ACTIVE', 0.0)
        if busy <= 0 or act <= 0:
            continue
        out.append((busy, key, calls[key], busy / (act / 8 * 256 * 4), act / 8 / max(calls[key], 1)))
    out.sort(reverse=True)
    print(f"{'kernel':72s} {'grid':>9s} {'n':>4s} {'MFMA util':>10s} {'cycles/launch':>14s}")
    for busy, (name, grid), n, util, cyc in out[:40]:
        print(f'{name:72s} {grid:9d} {n:4d} {100 * util:9.1f}% {cyc:14.0f}')


if __name__ == '__main__':
    main()
