"""Join rocprofv3 kernel-trace + one --pmc pass: per-kernel average counter value per dispatch (KB for FETCH/WRITE_SIZE).
    python tools/pmc_summary.py <dir with run_counter_collection.csv> [top]
"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n).replace("papc::", "")
    return re.sub(r"\(.*\)$", "", n)[:90]


d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = defaultdict(lambda: [0, 0.0])
cname = None
with open(d + "/run_counter_collection.csv") as f:
    for r in csv.DictReader(f):
        cname = r["Counter_Name"]
        k = (short(r["Kernel_Name"]), r.get("Grid_Size", ""))
        a = agg[k]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print("counter:", cname, "(rocprofv3 unit: KB for FETCH_SIZE / WRITE_SIZE)")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
for (k, g), (n, v) in rows:
    print("%-92s grid=%-9s calls=%4d avg=%12.1f" % (k, g, n, v / n))
