"""Review item 3(c): `roofline.frac_binding` recomputed from the two PMC profile files alone --
profiles/<tag>_mfma_util.txt (per (kernel, grid): launches, matrix-pipe utilisation, active cycles per
launch) x profiles/<tag>_pmc_hbm_traffic.txt (per (kernel, grid): corrected FETCH + WRITE KB per launch).
Per launch: floor = max(utilisation x cycles  [the cycles the matrix pipe was busy: what the launch would
take at 100 %], counted bytes / 8 TB/s x clock); frac_binding = sum(floor) / sum(cycles) over every
pw_fwd_kernel / pw_wgrad_kernel launch.  COUNTED bytes and MEASURED pipe cycles instead of the bench
line's algorithmic bytes and flops: an independent reading of the same quantity (the un-captured
counter passes run every launch alone with grids sized for 256 CUs, like the bench line's eager pass).
usage: python tools/frac_binding_check.py [tag = r05] [clock GHz = 2.4]"""
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
R = __file__.rsplit('/tools/', 1)[0] + '/profiles/'
def norm(name):
    return re.sub(r'^void ', '', name).replace(' ', '')


util = {}
for line in open(R + f'{tag}_mfma_util.txt'):
    f = line.split()
    if len(f) >= 5 and 'nesie::pw_' in line and f[-2].endswith('%') and ('pw_fwd_kernel' in line or 'pw_wgrad_kernel' in line):
        util[(norm(' '.join(f[:-4])), int(f[-4]))] = (int(f[-3]), float(f[-2][:-1]) / 100, int(f[-1]))
traffic = {}
for line in open(R + f'{tag}_pmc_hbm_traffic.txt'):
    f = line.split()
    if len(f) >= 6 and ('pw_fwd_kernel' in line or 'pw_wgrad_kernel' in line):
        try:
            grid, corr, wr = int(f[-5]), float(f[-2]), float(f[-1])
        except ValueError:
            continue
        traffic.setdefault(norm(' '.join(f[:-5])), {})[grid] = (corr + wr) * 1024
tot = flo = hb = 0.0
rows = []
for (name, grid), (n, u, cyc) in util.items():
    # (the traffic file truncates long kernel names: match by prefix)
    by = next((v[grid] for k, v in traffic.items() if name.startswith(k) and grid in v), 0.0)
    t_mfma = u * cyc
    t_hbm = by / 8e12 * ghz * 1e9
    f = max(t_mfma, t_hbm)
    tot += n * cyc; flo += n * f; hb += n * cyc if t_hbm > t_mfma else 0
    rows.append((n * (cyc - f), name, grid, n, u, by / 1e6, 'hbm' if t_hbm > t_mfma else 'mfma', f / cyc))
print(f'{tag}: frac_binding from the PMC files, over the launches listed there = {flo / tot:.3f}  (matrix-pipe utilisation alone: '
      f'{sum(n * u * c for (_, _), (n, u, c) in util.items()) / tot:.3f}; {100 * hb / tot:.0f} % of the family\'s cycles in HBM-bound launches; clock {ghz} GHz)')
print('most cycles lost:  kernel | grid | launches | pipe util | MB per launch | bound | own fraction')
for lost, name, grid, n, u, mb, b, fr in sorted(rows, reverse=True)[:10]:
    print(f'  {name[:60]:60s} {grid:8d} {n:3d} {100 * u:5.1f}% {mb:8.0f} {b:5s} {fr:.2f}')
