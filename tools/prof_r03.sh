# Round-3 profile set of bench.py on one MI355X (run through gpurun from the repo root):
#   kernel trace + stats (graph mode), MFMA-utilisation PMC pass, HBM FETCH / WRITE PMC passes
# (separate processes; counters are never combined with tracing), and the residency report of the
# layer-kernel instantiations (hipOccupancyMaxActiveBlocksPerMultiprocessor, NESIE_PW_OCCUPANCY).
# Raw output: gpurun_out/prof_r03/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 > $O/trace_bench.json 2> $O/trace.err &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0 > $O/mfma_bench.json 2> $O/mfma.err &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0 > $O/fetch_bench.json 2> $O/fetch.err &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0 > $O/write_bench.json 2> $O/write.err &&
NESIE_PW_OCCUPANCY=1 timeout -k 10 300 python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0 > $O/occ_bench.json 2> $O/occupancy.err
find $O -name "*.csv" | head -20
