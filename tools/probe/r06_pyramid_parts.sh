# what each stage of the sampling pyramid costs the step (stages left out of the side graph one at a time: stale plans, timing only)
cd "$GRAFT_REPO_ROOT"
run() { env PAPC_DIAG_SKIP=$1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg --batches 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('skip [$1]', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"; }
for i in 1 2; do
  run none
  run fps512
  run bq512
  run xyzpre512
  run fps128
  run bq128,compact128,lists128
  run lists128
  run fps512,fps128
  run bq512,xyzpre512,bq128,compact128,lists128
  run fps512,bq512,xyzpre512,fps128,bq128,compact128,lists128
done
