"""Pivot a rocprofv3 --pmc run (run_counter_collection.csv): one row per kernel, one column per counter (sum over all
dispatches).   python tools/pmc_table.py <dir> [top] [name-filter]"""
import csv, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n).replace("papc::", "")
    return re.sub(r"\(.*\)$", "", n)[:70]


d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
names = []
with open(d + "/run_counter_collection.csv") as f:
    for r in csv.DictReader(f):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        agg[k][c] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
key = names[0]
rows = sorted(agg.items(), key=lambda kv: -kv[1][key])
print("%-70s %5s " % ("kernel", "calls") + " ".join("%14s" % n[-14:] for n in names))
for k, v in rows[:top]:
    if flt in k:
        print("%-70s %5d " % (k, len(calls[k])) + " ".join("%14.4g" % v[n] for n in names))
