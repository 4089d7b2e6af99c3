cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_e
mkdir -p $O
for v in 1 0; do
PAPC_LG_LISTS=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$v -o run -- python bench.py --no-cpu-baseline --no-padded-leg --steps 50 --warmup 10 > $O/prof$v.log 2>&1
f=$(find $O/prof$v -name "*kernel_stats.csv" | head -1)
echo "== lists=$v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("lingather","point_lists","fill_kernel","compact_","fps_kernel","ball_query","seg_max","flag_")):
        print("%-60s calls %6s avg %9.1f us total %10.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
done
