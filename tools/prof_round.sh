# Profile set of bench.py on one MI355X (run through gpurun from the repo root):
#   bash tools/prof_round.sh r05                          (the headline workload)
#   bash tools/prof_round.sh r05_semi --workload semi     (any further bench.py arguments)
#   kernel trace + stats (graph mode), MFMA-utilisation PMC pass, HBM FETCH / WRITE PMC passes
# (separate processes; counters are never combined with tracing), the sha256 of the library that ran,
# and a clean bench line without the profiler.  Raw output: gpurun_out/prof_<tag>/ (the raw traces
# are deleted; tools/make_profiles.py turns the rest into profiles/<tag>_*).
TAG=${1:-r05}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
echo "$@" > $O/bench_args.txt
sha256sum $R/nesie_amd/libnesie_hip.so | cut -d' ' -f1 > $O/lib.sha256
python3 $R/bench.py --steps 20 --warmup 5 "$@" > $O/clean_bench.json 2> $O/clean.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 --other-workloads 0 "$@" > $O/trace_bench.json 2> $O/trace.err || exit 1
T=$(find $O/trace -name "*_kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*_kernel_stats.csv" | head -1)
cp $S $O/kernel_stats.csv
python3 $R/tools/timeline.py $T 0 --list > $O/timeline.txt 2>&1
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --other-workloads 0 --graph 0 --parity-gate 0 "$@" > $O/mfma_bench.json 2> $O/mfma.err || exit 1
cp $(find $O/mfma -name "*_counter_collection.csv" | head -1) $O/mfma_counters.csv && rm -rf $O/mfma
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --other-workloads 0 --graph 0 --parity-gate 0 "$@" > $O/fetch_bench.json 2> $O/fetch.err || exit 1
cp $(find $O/fetch -name "*_counter_collection.csv" | head -1) $O/fetch_counters.csv && rm -rf $O/fetch
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --other-workloads 0 --graph 0 --parity-gate 0 "$@" > $O/write_bench.json 2> $O/write.err || exit 1
cp $(find $O/write -name "*_counter_collection.csv" | head -1) $O/write_counters.csv && rm -rf $O/write
ls -la $O
