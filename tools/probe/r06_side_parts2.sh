# the remaining pieces of the two-stream structure (end-of-step wait already off): EMPTY gated side graph, diagnostic
cd "$GRAFT_REPO_ROOT"
run() { env $1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg --diag-empty-side 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"; }
for i in 1 2 3; do
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('fixed plan', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"
  run X=0
  run PAPC_DIAG_NO_SLOT_COPY=1
  run PAPC_DIAG_NO_SIDE_EV=1
  run "PAPC_DIAG_NO_SLOT_COPY=1 PAPC_DIAG_NO_SIDE_EV=1"
done
