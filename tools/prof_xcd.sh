# FETCH_SIZE of the layer kernel with and without the XCD-aware row-block mapping (NESIE_PW_XCD)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_xcd
mkdir -p $O
export NESIE_PW_XCD=0
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/off --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0 > $O/off.json 2> $O/off.err &&
export NESIE_PW_XCD=1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/on --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --graph 0 --parity-gate 0 > $O/on.json 2> $O/on.err
find $O -name "*counter_collection.csv"
