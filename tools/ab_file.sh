# same-box A/B of bench.py with ONE python file swapped: bash tools/ab_file.sh <path in repo> <old copy> [runs]
F=$1; OLD=$2; RUNS=${3:-3}
R=$GRAFT_REPO_ROOT
cp $R/$F /tmp/ab_new_file
run() { (cd $R && python bench.py --steps 30 --warmup 5 --cpu-baseline 0 --parity-gate 0 --other-workloads 0 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); }
for i in $(seq 1 $RUNS); do
  cp $OLD $R/$F; echo "old $(run)"
  cp /tmp/ab_new_file $R/$F; echo "new $(run)"
done
