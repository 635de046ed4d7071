"""Aggregate a rocprofv3 kernel_trace.csv over the LAST k training steps only
(steps are delimited by the once-per-step SA1 FPS launch), so MIOpen's one-time
find/tuning kernels in the warm-up do not pollute the per-step breakdown."""
import collections, csv, sys

path, k = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
marker = sys.argv[3] if len(sys.argv) > 3 else 'fps_pruned'
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
sel = rows[marks[-k]:]
t0, t1 = int(sel[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in sel)
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r['Kernel_Name']]
    a[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); a[1] += 1
busy = sum(a[0] for a in agg.values())
print(f"steps={k} wall/step={(t1 - t0) / k / 1e6:.3f} ms  kernel-busy/step={busy / k / 1e6:.3f} ms  "
      f"launches/step={len(sel) / k:.0f}")
for name, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print("%6.2f%% %8.3f ms/step  n/step=%6.1f  avg=%9.1f us  %s" % (100 * d / busy, d / k / 1e6, n / k, d / n / 1e3, name[:110]))
