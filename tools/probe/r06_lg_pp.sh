# gather-add backward over point lists: the two-stream kernel (PAPC_LG_PP=0: gathers dz and y) against the one that gathers dz alone (=1).
# Same box, interleaved: whole step (real and fixed plan) + the kernels' own durations under rocprofv3.
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-padded-leg "$@" 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('pp', os.environ.get('PAPC_LG_PP'), '$*', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"; }
for i in ${REPS:-1 2 3}; do
  for v in 0 1; do
    export PAPC_LG_PP=$v
    run
    run --diag-fixed-plan
  done
done
export TMPDIR=/tmp
for v in 0 1; do
  export PAPC_LG_PP=$v
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lgpp$v -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-padded-leg --no-graph --steps 10 --warmup 3 >/dev/null 2>&1)
  echo "== PAPC_LG_PP=$v"
  f=$(find /tmp/lgpp$v -name "run_kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('lingather','point_lists','stream_kernel<3, 2, 8','stream_kernel<(papc::AMode)3','wx','lg_')):
        print('%-110s calls %5s avg %9.1f ns' % (n[:110], r['Calls'], float(r['AverageNs'])))
PY
done
