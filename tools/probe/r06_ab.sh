# same-box A/B of an environment switch: bash tools/probe/r06_ab.sh VAR A B [runs] [extra bench args]
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; N=${4:-3}; shift 4
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$VAR=$v', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'], 'padded', d.get('ms_per_step_padded'))"
  done
done
