"""profiles/<tag>_* from the output of tools/prof_round.sh (gpurun_out/prof_<tag>/).
usage: python tools/make_profiles.py r04"""
import csv
import json
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
STEPS = 20          # 2 eager warm-up + 3 warm-up + 10 timed + 5 un-captured steps of the traced process
PMC_STEPS = 5       # 1 warm-up + 2 timed + 2 un-captured steps of the counter passes (no parity gate)
BIG_BYTES = 100e6   # family = launches of the layer kernels that move >= 100 MB (>= 32768 positions)


def run(*cmd):
    return subprocess.run(['python', *cmd], capture_output=True, text=True, cwd=R).stdout


def pmc_rows(path, counter):
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') == counter:
            yield re.sub(r'\(.*', '', r['Kernel_Name']), int(r['Grid_Size']), float(r['Counter_Value'])


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
    P = R + f'gpurun_out/prof_{tag}/'
    out = R + f'profiles/{tag}_'
    sha = open(P + 'lib.sha256').read().strip()
    bench_args = open(P + 'bench_args.txt').read().strip() if os.path.exists(P + 'bench_args.txt') else ''
    extra = (' ' + bench_args) if bench_args else ''
    open(out + 'bench_kernel_stats.csv', 'w').write(open(P + 'kernel_stats.csv').read())
    b, clean = last_json(P + 'trace_bench.json'), last_json(P + 'clean_bench.json')
    json.dump(clean, open(out + 'bench_line.json', 'w'), indent=1)
    rows = list(csv.DictReader(open(P + 'kernel_stats.csv')))

    def tot(pred):
        return sum(float(r['TotalDurationNs']) for r in rows if pred(r['Name'])) / STEPS / 1e6
    native = tot(lambda n: any(k in n for k in ('pw_fwd_kernel', 'pw_wgrad_kernel', 'conv_wgrad_kernel', 'mlp_stream_kernel')))
    rb = tot(lambda n: n.startswith('Cijk'))
    calls = sum(int(r['Calls']) for r in rows) / STEPS
    aten = tot(lambda n: 'nesie::' not in n and not n.startswith('Cijk'))
    allt = tot(lambda n: True)
    side = tot(lambda n: any(k in n for k in ('fps_pruned', 'fps_reg', 'ball_query', 'inverted_index', 'three_nn_kernel')))
    small = sum(int(r['Calls']) for r in rows if float(r['AverageNs']) < 12000) / STEPS
    small_ms = sum(float(r['TotalDurationNs']) for r in rows if float(r['AverageNs']) < 12000) / STEPS / 1e6
    hdr = [
        f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --cpu-baseline 0 --other-workloads 0{extra}   (1x MI355X, {tag}, tools/prof_round.sh; libnesie_hip.so sha256 {sha[:16]}..)",
        f"bench line under the profiler: value={b['value']:.1f} scenes/s  ms_per_step={b['ms_per_step']:.2f}; without it (profiles/{tag}_bench_line.json): {clean['value']:.1f} scenes/s, {clean['ms_per_step']:.2f} ms",
        "The process runs 20 training steps in all (2 eager warm-up steps before capture, 3 warm-up, 10 timed, 5 un-captured for the HIP-event",
        "timings of the roofline entries) plus the parity gate's one B=2 step on each leg; per-step = total / 20.  FPS, ball query, inverted indices, FP taps and",
        f"the vote targets run on a side stream under the previous step (fps_pruned + fps_reg + ball_query + inverted_index + three_nn = {side:.1f} ms of kernel time, mostly one CU per scene: overlapped, at a measured cost of 0.30 ms of interference + 0.17 ms for the forward's reduced CU budget, DESIGN.md section 0 item 6).",
        f"GEMM time: nesie::pw_fwd_kernel + pw_wgrad_kernel (+ conv_wgrad, mlp_stream) = {native:.2f} ms vs rocBLAS {rb:.2f} ms -> {100 * native / (native + rb):.0f} % of the GEMM time is in nesie:: kernels.",
        f"Launches per step (both streams): {calls:.0f}; kernel families averaging under 12 us: {small:.0f} launches, {small_ms:.2f} ms; everything that is neither nesie:: nor rocBLAS "
        f"(ATen elementwise / reduce / cat / copies / fills): {aten:.2f} ms = {100 * aten / allt:.1f} % of the summed kernel time.",
        ""]
    open(out + 'bench_per_step_summary.txt', 'w').write(
        run('tools/profile_summary.py', f'profiles/{tag}_bench_kernel_stats.csv', str(STEPS), *hdr))
    open(out + 'timeline.txt', 'w').write(
        "tools/timeline.py on the kernel trace of the same run: the last window between two SA1 sampling launches that holds a graph-replayed step (two queues: main stream and the pipelined index chain)\n"
        + open(P + 'timeline.txt').read())
    # ---- matrix-pipe utilisation
    mf = P + 'mfma_counters.csv'
    busy = act = 0.0
    for name, grid, val in pmc_rows(mf, 'SQ_VALU_MFMA_BUSY_CYCLES'):
        if 'pw_fwd_kernel' in name or 'pw_wgrad_kernel' in name:
            busy += val
    for name, grid, val in pmc_rows(mf, 'GRBM_GUI_ACTIVE'):
        if 'pw_fwd_kernel' in name or 'pw_wgrad_kernel' in name:
            act += val
    util = busy / (act / 8 * 256 * 4) if act else 0.0
    head = (f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --other-workloads 0 --graph 0 --parity-gate 0{extra}   ({tag}; tools/prof_round.sh)\n"
            "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) (tools/mfma_util.py)\n"
            f"time-weighted over every pw_fwd_kernel / pw_wgrad_kernel launch (1-D chains included): {100 * util:.1f} % of the matrix-pipe cycles\n\n")
    open(out + 'mfma_util.txt', 'w').write(head + run('tools/mfma_util.py', mf))
    # ---- HBM traffic
    fe, wr = P + 'fetch_counters.csv', P + 'write_counters.csv'
    fam_f = fam_w = 0.0
    per = {}
    for name, grid, kb in pmc_rows(fe, 'FETCH_SIZE'):
        if 'pw_fwd_kernel' in name or 'pw_wgrad_kernel' in name:
            per.setdefault((name, grid), [0, 0.0, 0.0])
            per[(name, grid)][0] += 1
            per[(name, grid)][1] += kb
    for name, grid, kb in pmc_rows(wr, 'WRITE_SIZE'):
        if (name, grid) in per:
            per[(name, grid)][2] += kb
    for (name, grid), (n, f, w) in per.items():
        if (2 * f + w) / n * 1024 >= BIG_BYTES:
            fam_f += 2 * f
            fam_w += w
    family_bytes = (fam_f + fam_w) * 1024 / PMC_STEPS
    big = max((kv for kv in per.items() if 'pw_fwd_kernel' in kv[0][0]), key=lambda kv: (2 * kv[1][1] + kv[1][2]) / kv[1][0])
    bf, bw = 2 * big[1][1] / big[1][0] * 1024, big[1][2] / big[1][0] * 1024
    # the SA1 64 -> 128 layer with the pooled tail and a store (round-3 review: writes above the algorithmic size)
    sa1 = [kv for kv in per.items() if re.search(r'pw_fwd_kernel<4, 1, 4, 1, 2, 79, 32', kv[0][0])]
    worst = None
    if sa1:
        (name, grid), (n, f, w) = max(sa1, key=lambda kv: kv[1][2] / kv[1][0])
        alg_w = 8 * 128 * 131072 * 4 + 2 * 8 * 128 * (131072 // 32) * 5       # Y + (max, min) x (value, position)
        worst = {'kernel': name, 'grid': grid, 'write_bytes': w / n * 1024, 'fetch_corrected_bytes': 2 * f / n * 1024,
                 'algorithmic_write_bytes': alg_w, 'write_over_algorithmic': w / n * 1024 / alg_w}
    json.dump({'lib_sha256': sha, 'bench_args': bench_args, 'family_bytes_per_step': family_bytes,
               'fetch_corrected_bytes_per_step': fam_f * 1024 / PMC_STEPS,
               'write_bytes_per_step': fam_w * 1024 / PMC_STEPS,
               'largest_launch': {'kernel': big[0][0], 'grid': big[0][1], 'fetch_corrected_bytes': bf, 'write_bytes': bw},
               'worst_launch': worst,
               'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of bench.py --steps 2 --warmup 1 --graph 0 --parity-gate 0; '
                         'FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md)'},
              open(out + 'pmc_hbm_traffic.json', 'w'), indent=1)
    head = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --other-workloads 0 --graph 0 --parity-gate 0{extra}   ({tag}; tools/prof_round.sh)\n"
            f"libnesie_hip.so sha256 {sha}\n"
            "Counter_Value is in KB per dispatch, averaged over the dispatches of a (kernel, grid) (tools/pmc_summary.py).  gfx950 correction\n"
            "(MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request, so dense streaming reads are DOUBLED in the corr. column;\n"
            "WRITE_SIZE is exact.  Gathers / atomics (blend, group) are left uncorrected.\n"
            f"Layer-kernel family (launches that move >= 100 MB): {family_bytes / 1e9:.2f} GB per step = corrected FETCH {fam_f * 1024 / PMC_STEPS / 1e9:.2f} + WRITE {fam_w * 1024 / PMC_STEPS / 1e9:.2f}\n"
            f"(bench.py divides this by the algorithmic bytes of the same launches: roofline.traffic.over_algorithmic).\n"
            f"Largest launch: {big[0][0]} grid {big[0][1]}: {bf / 1e6:.0f} MB read + {bw / 1e6:.0f} MB written.\n"
            + (f"SA1 64 -> 128 pooled-store layer: {worst['write_bytes'] / 1e6:.0f} MB written for {worst['algorithmic_write_bytes'] / 1e6:.0f} MB of output + pool partials "
               f"= {worst['write_over_algorithmic']:.2f} x.\n" if worst else "") + "\n")
    open(out + 'pmc_hbm_traffic.txt', 'w').write(head + run('tools/pmc_summary.py', fe, wr))
    print(open(out + 'bench_per_step_summary.txt').read()[:3000])


if __name__ == '__main__':
    main()
