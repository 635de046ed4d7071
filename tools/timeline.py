"""Main-stream timeline of ONE replayed training step from a rocprofv3 kernel_trace.csv of bench.py:
every kernel in start order with its duration and the idle gap in front of it, the per-queue busy /
idle totals and the small-kernel tail.  Steps are delimited by the once-per-step SA1 FPS launch
(side stream); the queue that carries the layer kernels is the main stream.
usage: python tools/timeline.py <kernel_trace.csv> [step from the end = 2] [--list]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'at::native::(\(anonymous namespace\)::)?', 'at::', name)
    m = re.match(r'(nesie::\w+)(<[^>]*>)?', name)
    if m:
        return m.group(1) + (m.group(2) or '')
    return name[:90]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 2
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
    marks = [i for i, r in enumerate(rows) if 'fps_pruned' in r['Kernel_Name']]
    # a step = from one FPS launch to the next; take the window `back` from the end -- or, with
    # back = 0, the LAST window that looks like a graph-replayed step: two queues in it (the index
    # chain on the side stream) and a full step's worth of layer-kernel launches on one of them
    if back == 0:
        best = None
        for j in range(len(marks) - 1, 0, -1):
            a, b = int(rows[marks[j - 1]]['Start_Timestamp']), int(rows[marks[j]]['Start_Timestamp'])
            qs = collections.Counter(r[qkey] for r in rows[marks[j - 1]:marks[j]] if 'pw_fwd_kernel' in r['Kernel_Name'])
            nq = len({r[qkey] for r in rows[marks[j - 1]:marks[j]]})
            if nq >= 2 and qs and max(qs.values()) >= 60 and (b - a) < 40e6:
                best = j
                break
        if best is None:
            raise SystemExit('no window that looks like a replayed step')
        lo, hi = marks[best - 1], marks[best]
    else:
        lo, hi = marks[-back - 1], marks[-back]
    t_lo, t_hi = int(rows[lo]['Start_Timestamp']), int(rows[hi]['Start_Timestamp'])
    per_q = collections.defaultdict(list)
    for r in rows:
        s = int(r['Start_Timestamp'])
        if t_lo <= s < t_hi:
            per_q[r[qkey]].append(r)
    main_q = max(per_q, key=lambda q: sum('pw_fwd_kernel' in r['Kernel_Name'] for r in per_q[q]))
    print(f'step window {1e-6 * (t_hi - t_lo):.3f} ms; queues: ' +
          ', '.join(f'{q}: {len(v)} launches' for q, v in per_q.items()) + f'; main = {main_q}')
    for q, ks in per_q.items():
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks)
        print(f'  queue {q}: busy {busy / 1e6:.3f} ms in {len(ks)} launches')
    ks = per_q[main_q]
    gaps, prev_end = [], None
    small_n = small_t = 0
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    lines = []
    for r in ks:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = 0 if prev_end is None else max(0, s - prev_end)
        gaps.append(gap)
        prev_end = max(e, prev_end or 0)
        d = e - s
        if d < 12000:
            small_n += 1
            small_t += d
        f = fam[short(r['Kernel_Name'])]
        f[0] += 1; f[1] += d; f[2] += gap
        lines.append((s - t_lo, d, gap, short(r['Kernel_Name']), r.get('Grid_Size', ''), r.get('Workgroup_Size', '')))
    busy = sum(l[1] for l in lines)
    span = prev_end - int(ks[0]['Start_Timestamp'])
    print(f'main stream: {len(ks)} launches, busy {busy / 1e6:.3f} ms, idle gaps {sum(gaps) / 1e6:.3f} ms '
          f'(median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, > 5 us: {sum(g > 5000 for g in gaps)}), '
          f'span {span / 1e6:.3f} ms')
    print(f'kernels under 12 us: {small_n} launches, {small_t / 1e6:.3f} ms')
    cls = collections.defaultdict(lambda: [0, 0.0])
    for n, (c, d, g) in fam.items():
        k = 'nesie::' if n.startswith('nesie::') else 'rocBLAS' if n.startswith('Cijk') else 'ATen / runtime copy / fill'
        cls[k][0] += c; cls[k][1] += d
    print('by class (main stream): ' + '; '.join(f'{k}: {c} launches, {d / 1e6:.3f} ms' for k, (c, d) in cls.items()))
    print('\nby kernel (main stream): launches | busy ms | gap-before ms | name')
    for n, (c, d, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:70]:
        print(f'{c:5d} {d / 1e6:8.3f} {g / 1e6:8.3f}  {n}')
    if '--list' in sys.argv:
        print('\nstart us | dur us | gap us | kernel | grid | wg')
        for s, d, g, n, grid, wg in lines:
            print(f'{s / 1e3:9.1f} {d / 1e3:8.1f} {g / 1e3:6.1f}  {n}  {grid} {wg}')


if __name__ == '__main__':
    main()
