# Round-3 profile set of bench.py on one MI355X (run through gpurun from the repo root):
#   kernel trace + stats (graph mode), MFMA-utilisation PMC pass, HBM FETCH / WRITE PMC passes
# (separate processes; counters are never combined with tracing).  Raw output: gpurun_out/prof_r03/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 > $O/trace_bench.json 2> $O/trace.err &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 > $O/mfma_bench.json 2> $O/mfma.err &&
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 > $O/fetch_bench.json 2> $O/fetch.err &&
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 > $O/write_bench.json 2> $O/write.err
find $O -name "*.csv" | head -20
