#!/bin/bash
# kernel-time table of one bench configuration: bash tools/probe/kstats.sh <tag> <bench.py args...>   (on the GPU box; table -> stdout)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
rm -rf /tmp/ks_$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o run -- python bench.py "$@" --no-cpu-baseline > /tmp/ks_$tag.log 2>&1
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ks_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
if not f:
    print(open("/tmp/ks_%s.log" % sys.argv[1]).read()[-2000:]); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:30]:
    print("%-110s %5s %10.0f %6s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
