# kernel trace of bench.py (graph mode) -> gpurun_out/<tag>/: stats csv, kernel trace csv, timeline
# usage (through gpurun, from the repo root): bash tools/prof_trace.sh <tag> [bench args]
TAG=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 --other-workloads 0 "$@" > $O/trace_bench.json 2> $O/trace.err || exit 1
T=$(find $O/trace -name "*_kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*_kernel_stats.csv" | head -1)
cp $S $O/kernel_stats.csv
python3 $R/tools/timeline.py $T ${BACK:-0} --list > $O/timeline.txt 2>&1
python3 $R/tools/profile_summary.py $S 20 > $O/per_step_summary.txt 2>&1
head -c 400 $O/trace_bench.json; echo; head -30 $O/timeline.txt
rm -rf $O/trace    # (the raw trace is hundreds of MB)
