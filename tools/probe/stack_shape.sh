#!/bin/bash
# bash tools/probe/stack_shape.sh <tag> B N S K D radius c1,c2,c3 [compact]   -> kernel-time table (us per call; 12 fwd+bwd passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
rm -rf /tmp/ss_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ss_$tag -o run -- python tools/probe/stack_shape_time.py "$@" > /tmp/ss_$tag.log 2>&1
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ss_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
if not f:
    print(open("/tmp/ss_%s.log" % sys.argv[1]).read()[-2000:]); sys.exit(1)
tot = 0.0
for r in csv.DictReader(open(f[0])):
    if int(r["Calls"]) < 12: continue
    per = float(r["TotalDurationNs"]) / 12e3
    tot += per
    if per > 8: print("%-100s %3d x %7.1f = %7.1f us/pass" % (r["Name"][:100], int(r["Calls"]) // 12, float(r["AverageNs"]) / 1e3, per))
print("== %s: %.1f us per fwd+bwd pass" % (sys.argv[1], tot))
PY
