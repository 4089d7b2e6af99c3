# where the sampling graph's gate opens (bench.py --fork), same box, two rounds
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for f in start sa1 sa2 sa3 loss; do
    python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg --fork $f 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('fork $f', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"
  done
done
