cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline 0 > $R/gpurun_out/prof_r2b_bench.json 2> $R/gpurun_out/prof_r2b.err
ls $R/gpurun_out/prof_r2b/*/ | head
