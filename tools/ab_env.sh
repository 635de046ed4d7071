# same-box A/B of bench.py under two environments: bash tools/ab_env.sh "VAR=a" "VAR=b" [runs] [bench args]
A=$1; B=$2; RUNS=${3:-3}; shift; shift; shift
R=$GRAFT_REPO_ROOT
for i in $(seq 1 $RUNS); do
  for v in "$A" "$B"; do
    ms=$(cd $R && env $v python bench.py --steps 30 --warmup 5 --cpu-baseline 0 --parity-gate 0 --other-workloads 0 "$@" 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v $ms"
  done
done
