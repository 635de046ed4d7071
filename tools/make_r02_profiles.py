"""profiles/r02_* from the raw rocprofv3 output of tools/prof_r02.sh (gpurun_out/prof_r02/) and a
clean bench line (gpurun_out/b_r2_final.json).  usage: python tools/make_r02_profiles.py"""
import csv
import glob
import json
import os
import subprocess

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
P = R + 'gpurun_out/prof_r02/'


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def run(*cmd):
    return subprocess.run(['python', *cmd], capture_output=True, text=True, cwd=R).stdout


stats = newest(P + 'trace/runc/*_kernel_stats.csv')
open(R + 'profiles/r02_bench_kernel_stats.csv', 'w').write(open(stats).read())
b = json.loads(open(P + 'trace_bench.json').read())
clean = json.loads(open(R + 'gpurun_out/b_r2_final.json').read())
rows = list(csv.DictReader(open(stats)))
STEPS = 20


def tot(pred):
    return sum(float(r['TotalDurationNs']) for r in rows if pred(r['Name'])) / STEPS / 1e6


native = tot(lambda n: any(k in n for k in ('pw_fwd_kernel', 'pw_wgrad_kernel', 'conv_wgrad_kernel', 'mlp_stream_kernel')))
rb = tot(lambda n: n.startswith('Cijk'))
calls = sum(int(r['Calls']) for r in rows) / STEPS
aten = tot(lambda n: 'nesie::' not in n and not n.startswith('Cijk'))
allt = tot(lambda n: True)
side = tot(lambda n: any(k in n for k in ('fps_pruned', 'fps_reg', 'ball_query', 'inverted_index', 'three_nn_kernel')))
hdr = [
    "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --cpu-baseline 0   (1x MI355X, round 2 final code, tools/prof_r02.sh)",
    f"bench line under the profiler: value={b['value']:.1f} scenes/s  ms_per_step={b['ms_per_step']:.2f}; without it (gpurun_out/b_r2_final.json, another box of the pool): {clean['value']:.1f} scenes/s, {clean['ms_per_step']:.2f} ms",
    "The process runs 20 training steps in all (2 eager warm-up steps before capture, 3 warm-up, 10 timed, 5 un-captured for the HIP-event",
    "timings of the roofline entries) plus the parity gate's one B=2 step; per-step = total / 20.  FPS, ball query, inverted indices, FP taps and",
    f"the vote targets run on a side stream under the previous step (fps_pruned + fps_reg + ball_query + inverted_index + three_nn = {side:.1f} ms, on 8 CUs: hidden).",
    f"GEMM time: nesie::pw_fwd_kernel + pw_wgrad_kernel (+ conv_wgrad, mlp_stream) = {native:.2f} ms vs rocBLAS {rb:.2f} ms -> {100 * native / (native + rb):.0f} % of the GEMM time is in nesie:: kernels.",
    f"Launches per step (both streams): {calls:.0f} (round 1: ~1 100); everything that is neither nesie:: nor rocBLAS (ATen elementwise / reduce / cat / copies / fills): {aten:.2f} ms = {100 * aten / allt:.1f} % of the summed kernel time.",
    "bn_stats / bn_apply remain only for the layers outside the fused stacks (FP modules, vote module, prediction heads).",
    ""]
open(R + 'profiles/r02_bench_per_step_summary.txt', 'w').write(
    run('tools/profile_summary.py', 'profiles/r02_bench_kernel_stats.csv', str(STEPS), *hdr))
for name, args in (('r02_mfma_util.txt', ['tools/mfma_util.py', newest(P + 'mfma/runc/*_counter_collection.csv')]),
                   ('r02_pmc_hbm_traffic.txt', ['tools/pmc_summary.py', newest(P + 'fetch/runc/*_counter_collection.csv'),
                                                newest(P + 'write/runc/*_counter_collection.csv')])):
    old = open(R + 'profiles/' + name).read()
    head = old[:old.index('kernel  ')]
    open(R + 'profiles/' + name, 'w').write(head + run(*args))
print(open(R + 'profiles/r02_bench_per_step_summary.txt').read()[:1800])
