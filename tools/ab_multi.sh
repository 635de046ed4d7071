# same-box comparison of bench.py under several environments, alternating:
#   bash tools/ab_multi.sh <runs> "VAR=a VAR2=b" "VAR=c" ...     (use "X=0" for the plain run)
RUNS=$1; shift
R=$GRAFT_REPO_ROOT
for i in $(seq 1 $RUNS); do
  for v in "$@"; do
    ms=$(cd $R && env $v python bench.py --steps 30 --warmup 5 --cpu-baseline 0 --parity-gate 0 --other-workloads 0 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v $ms"
  done
done
