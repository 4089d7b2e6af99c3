# what the gate's spinning wave costs by itself: fixed plan (no side stream) against an EMPTY gated side graph, with and without extra spinning
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$*', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"; }
for i in 1 2; do
  run --diag-fixed-plan
  run --diag-empty-side
  run --diag-empty-side --side-delay-us 300
  run --diag-empty-side --fork start
  run --batches 1
done
