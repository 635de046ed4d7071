"""profiles/r03_* from the raw rocprofv3 output of tools/prof_r03.sh (gpurun_out/prof_r03/) and a
clean bench line (gpurun_out/b_r3_final.json).  usage: python tools/make_r03_profiles.py"""
import csv
import glob
import json
import os
import re
import subprocess

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
P = R + 'gpurun_out/prof_r03/'
STEPS = 20          # 2 eager warm-up + 3 warm-up + 10 timed + 5 un-captured steps of the traced process
PMC_STEPS = 5       # 1 warm-up + 2 timed + 2 un-captured steps of the counter passes (no parity gate)
BIG = 32768 * 4     # family = launches over >= 32768 positions; told apart here by >= 100 MB moved


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def run(*cmd):
    return subprocess.run(['python', *cmd], capture_output=True, text=True, cwd=R).stdout


def pmc_rows(path, counter):
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') == counter:
            yield re.sub(r'\(.*', '', r['Kernel_Name']), int(r['Grid_Size']), float(r['Counter_Value'])


def main():
    stats = newest(P + 'trace/*/*_kernel_stats.csv')
    open(R + 'profiles/r03_bench_kernel_stats.csv', 'w').write(open(stats).read())
    b = json.loads(open(P + 'trace_bench.json').read().strip().splitlines()[-1])
    clean = json.loads(open(R + 'gpurun_out/b_r3_final.json').read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(stats)))

    def tot(pred):
        return sum(float(r['TotalDurationNs']) for r in rows if pred(r['Name'])) / STEPS / 1e6
    native = tot(lambda n: any(k in n for k in ('pw_fwd_kernel', 'pw_wgrad_kernel', 'conv_wgrad_kernel', 'mlp_stream_kernel')))
    rb = tot(lambda n: n.startswith('Cijk'))
    calls = sum(int(r['Calls']) for r in rows) / STEPS
    aten = tot(lambda n: 'nesie::' not in n and not n.startswith('Cijk'))
    allt = tot(lambda n: True)
    side = tot(lambda n: any(k in n for k in ('fps_pruned', 'fps_reg', 'ball_query', 'inverted_index', 'three_nn_kernel')))
    hdr = [
        "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --cpu-baseline 0   (1x MI355X, round 3 final code, tools/prof_r03.sh)",
        f"bench line under the profiler: value={b['value']:.1f} scenes/s  ms_per_step={b['ms_per_step']:.2f}; without it (gpurun_out/b_r3_final.json, another box of the pool): {clean['value']:.1f} scenes/s, {clean['ms_per_step']:.2f} ms",
        "The process runs 20 training steps in all (2 eager warm-up steps before capture, 3 warm-up, 10 timed, 5 un-captured for the HIP-event",
        "timings of the roofline entries) plus the parity gate's one B=2 step; per-step = total / 20.  FPS, ball query, inverted indices, FP taps and",
        f"the vote targets run on a side stream under the previous step (fps_pruned + fps_reg + ball_query + inverted_index + three_nn = {side:.1f} ms, on 8 CUs: hidden).",
        f"GEMM time: nesie::pw_fwd_kernel + pw_wgrad_kernel (+ conv_wgrad, mlp_stream) = {native:.2f} ms vs rocBLAS {rb:.2f} ms -> {100 * native / (native + rb):.0f} % of the GEMM time is in nesie:: kernels.",
        f"Launches per step (both streams): {calls:.0f} (round 2: 619); everything that is neither nesie:: nor rocBLAS (ATen elementwise / reduce / cat / copies / fills): {aten:.2f} ms = {100 * aten / allt:.1f} % of the summed kernel time.",
        "The 1-D chains (vote module, prediction trunk + output convolutions, feature propagation, score heads) run on the layer kernel (fused_mlp.Stack1dFn): their",
        "bn_stats / bn_apply passes are gone; rocBLAS keeps the blend tables, the wide weight gradients (256 x 256, 256 x 512) and the skinny coordinate rows.",
        ""]
    open(R + 'profiles/r03_bench_per_step_summary.txt', 'w').write(
        run('tools/profile_summary.py', 'profiles/r03_bench_kernel_stats.csv', str(STEPS), *hdr))
    # ---- matrix-pipe utilisation
    mf = newest(P + 'mfma/*/*_counter_collection.csv')
    busy = act = 0.0
    for name, grid, val in pmc_rows(mf, 'SQ_VALU_MFMA_BUSY_CYCLES'):
        if 'pw_fwd_kernel' in name or 'pw_wgrad_kernel' in name:
            busy += val
    for name, grid, val in pmc_rows(mf, 'GRBM_GUI_ACTIVE'):
        if 'pw_fwd_kernel' in name or 'pw_wgrad_kernel' in name:
            act += val
    util = busy / (act / 8 * 256 * 4) if act else 0.0
    head = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0   (round 3; tools/prof_r03.sh)\n"
            "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) (tools/mfma_util.py)\n"
            f"time-weighted over every pw_fwd_kernel / pw_wgrad_kernel launch (1-D chains included): {100 * util:.1f} % of the matrix-pipe cycles\n\n")
    open(R + 'profiles/r03_mfma_util.txt', 'w').write(head + run('tools/mfma_util.py', mf))
    # ---- HBM traffic
    fe, wr = newest(P + 'fetch/*/*_counter_collection.csv'), newest(P + 'write/*/*_counter_collection.csv')
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
        if (2 * f + w) / n * 1024 >= BIG * 256:      # >= ~100 MB per launch: the grouped per-seed MLPs
            fam_f += 2 * f
            fam_w += w
    family_bytes = (fam_f + fam_w) * 1024 / PMC_STEPS
    # (the largest LAYER-kernel launch: bench.py compares it with the largest algorithmic byte count of that kernel)
    big = max((kv for kv in per.items() if 'pw_fwd_kernel' in kv[0][0]), key=lambda kv: (2 * kv[1][1] + kv[1][2]) / kv[1][0])
    bf, bw = 2 * big[1][1] / big[1][0] * 1024, big[1][2] / big[1][0] * 1024
    json.dump({'family_bytes_per_step': family_bytes, 'fetch_corrected_bytes_per_step': fam_f * 1024 / PMC_STEPS,
               'write_bytes_per_step': fam_w * 1024 / PMC_STEPS,
               'largest_launch': {'kernel': big[0][0], 'grid': big[0][1], 'fetch_corrected_bytes': bf, 'write_bytes': bw},
               'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of bench.py --steps 2 --warmup 1 --graph 0 --parity-gate 0; '
                         'FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md)'},
              open(R + 'profiles/r03_pmc_hbm_traffic.json', 'w'), indent=1)
    head = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0   (round 3; tools/prof_r03.sh)\n"
            "Counter_Value is in KB per dispatch, averaged over the dispatches of a (kernel, grid) (tools/pmc_summary.py).  gfx950 correction\n"
            "(MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request, so dense streaming reads are DOUBLED in the corr. column;\n"
            "WRITE_SIZE is exact.  Gathers / atomics (blend, group) are left uncorrected.\n"
            f"Layer-kernel family (launches that move >= 100 MB): {family_bytes / 1e9:.2f} GB per step = corrected FETCH {fam_f * 1024 / PMC_STEPS / 1e9:.2f} + WRITE {fam_w * 1024 / PMC_STEPS / 1e9:.2f}\n"
            f"(bench.py divides this by the algorithmic bytes of the same launches: roofline.traffic.over_algorithmic).\n"
            f"Largest launch: {big[0][0]} grid {big[0][1]}: {bf / 1e6:.0f} MB read + {bw / 1e6:.0f} MB written.\n\n")
    open(R + 'profiles/r03_pmc_hbm_traffic.txt', 'w').write(head + run('tools/pmc_summary.py', fe, wr))
    # ---- registers and residency
    occ = sorted(set(l.strip() for l in open(P + 'occupancy.err') if l.startswith('pw_fwd_kernel<')))
    regs = run('tools/isa_regs.py', '-j', '8')
    open(R + 'profiles/r03_isa_regs.txt', 'w').write(
        "python tools/isa_regs.py   (every instantiation of the layer kernels: .vgpr_count, allocation granule 8, waves per SIMD, workgroups per CU by registers)\n"
        "followed by the residency the runtime reports for the instantiations the step launches (NESIE_PW_OCCUPANCY=1: hipOccupancyMaxActiveBlocksPerMultiprocessor\n"
        "with the launch's dynamic LDS) -- the two must agree for the variants sized for two workgroups per CU.\n\n" + regs +
        "\n---- hipOccupancyMaxActiveBlocksPerMultiprocessor (bench.py --steps 1, B = 8) ----\n" + '\n'.join(occ) + '\n')
    print(open(R + 'profiles/r03_bench_per_step_summary.txt').read()[:2500])


if __name__ == '__main__':
    main()
