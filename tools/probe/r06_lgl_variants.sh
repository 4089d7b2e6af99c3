# kernel time of the list-based gather-add backward per geometry variant (same box): rocprofv3 kernel stats of a short bench each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_lgl
mkdir -p $O
for v in 0 1 2 3; do
PAPC_LG_PP=0 PAPC_LGL_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -o run -- python bench.py --no-cpu-baseline --no-padded-leg --steps 30 --warmup 5 --diag-fixed-plan > $O/v$v.log 2>&1
f=$(find $O/v$v -name "*kernel_stats.csv" | head -1)
python - "$f" $v <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lingather_bwd_lists" in r["Name"]:
        print("variant", sys.argv[2], "calls", r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
rm -rf $O/v$v
done
