# the padded SA2 stack (PAPC_COMPACT=0) with its gather-add backward on float atomics (PAPC_POINT_LISTS=1: lists for compacted stacks only) against
# the point lists (=2: the one-stream list kernel), same box, interleaved; + the kernels' durations
cd "$GRAFT_REPO_ROOT"
export PAPC_COMPACT=0
for i in 1 2; do
  for v in 1 2; do
    PAPC_POINT_LISTS=$v python bench.py --no-cpu-baseline --no-padded-leg --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('lists=$v padded step', d['ms_per_step'], d['ms_median'], d['ms_min'])"
  done
done
export TMPDIR=/tmp PAPC_POINT_LISTS=2
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/padl -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-padded-leg --steps 40 --warmup 10 >/dev/null 2>&1)
f=$(find /tmp/padl -name "run_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('lingather','point_lists','point_moments','fill_kernel')):
        print('%-80s calls %5s avg %8.1f us' % (n[:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
