# same-box A/B of bench.py: this tree vs build_ab/<name> (a checkout of another commit with its own
# built library), alternating runs.  usage (through gpurun): bash tools/ab.sh <name> [runs] [bench args]
NAME=${1:-r4}; RUNS=${2:-3}; shift; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_$NAME
mkdir -p $O
for i in $(seq 1 $RUNS); do
  (cd $R/build_ab/$NAME && python bench.py --steps 30 --warmup 5 --cpu-baseline 0 --parity-gate 0 "$@" > $O/base_$i.json 2> $O/base_$i.err) || exit 1
  (cd $R && python bench.py --steps 30 --warmup 5 --cpu-baseline 0 --parity-gate 0 --other-workloads 0 "$@" > $O/new_$i.json 2> $O/new_$i.err) || exit 1
done
python - <<PY
import json, glob
for kind in ('base', 'new'):
    v = [json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'] for f in sorted(glob.glob('$O/%s_*.json' % kind))]
    print(kind, ' '.join('%.3f' % x for x in v), 'mean %.3f' % (sum(v) / len(v)))
PY
