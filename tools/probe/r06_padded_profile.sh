# per-kernel profile of the step with SA2 forced padded (PAPC_COMPACT=0): what `value_padded` is made of
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp PAPC_COMPACT=0
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/padp -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-padded-leg --steps 40 --warmup 10 >/dev/null 2>&1)
f=$(find /tmp/padp -name "run_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=None
for r in rows[:34]:
    print('%-112s calls %5s avg %8.1f us' % (r['Name'][:112], r['Calls'], float(r['AverageNs'])/1e3))
PY
python bench.py --no-cpu-baseline --no-padded-leg --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('padded step', d['ms_per_step'], d['ms_median'])"
