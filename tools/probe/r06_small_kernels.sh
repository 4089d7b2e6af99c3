# launch-sized kernels of the step after the round's last changes: calls per step and average duration (rocprofv3 kernel trace of the graph-replayed bench)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/smallk -o run -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-padded-leg --steps 40 --warmup 10 >/dev/null 2>&1)
f=$(find /tmp/smallk -name "run_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('copy2d','bn_finalize','bn_bwd_finalize','transpose_batch','lingather','point_moments','point_lists','copyBuffer')):
        print('%-70s calls %5s avg %9.1f ns' % (n[:70], r['Calls'], float(r['AverageNs'])))
PY
