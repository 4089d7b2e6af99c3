"""per-step summary of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <kernel_stats.csv> <steps> [filter]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tot = 0.0
for r in rows:
    n, t = int(r["Calls"]), float(r["TotalDurationNs"])
    tot += t
    nm = r["Name"].replace("papc::", "").replace("void ", "")[:90]
    if flt in nm:
        print("%-90s %5.1f/step %8.1f us avg %8.1f us/step" % (nm, n / steps, float(r["AverageNs"]) / 1e3, t / steps / 1e3))
print("total us/step %.1f" % (tot / steps / 1e3))
