#!/bin/bash
# per-launch durations of the planes kernels in one bench run, grouped by (kernel, grid): bash tools/probe/ktrace_pg.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > /tmp/kt.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "pg_" not in n: continue
    g[(n[:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items()):
    v.sort()
    print("%-62s grid %8s  n %4d  min %6.1f  med %6.1f  max %6.1f us" % (k[0], k[1], len(v), v[0], v[len(v) // 2], v[-1]))
PY
