# what the side stream's wait for the previous step's end-of-step event costs, by event flavour (EMPTY gated side graph; diagnostic)
cd "$GRAFT_REPO_ROOT"
run() { env $1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg --diag-empty-side 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"; }
for i in 1 2 3; do
  run PAPC_PREV_END_MODE=timing
  run PAPC_PREV_END_MODE=plain
  run PAPC_PREV_END_MODE=hipdev
  run PAPC_DIAG_NO_PREV_END=1
done
