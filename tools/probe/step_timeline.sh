#!/bin/bash
# kernel sequence of ONE eager training step (between two Adam launches), in launch order with durations:
#   bash tools/probe/step_timeline.sh --config msg_seg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python bench.py --no-cpu-baseline --no-graph --steps 4 --warmup 2 "$@" > /tmp/tl.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = ad[-2], ad[-1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void papc::", "").replace("papc::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n[:90]))
PY
