"""Per-dispatch timeline of ONE graph-replayed step out of a rocprofv3 rocpd database (kernel start / end, gap to the previous end).
    python tools/step_timeline.py <results.db> [adam index, default: the middle one]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("papc::", "")
    return re.sub(r"\(.*\)$", "", n)[:78]


db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0] or "adam_dev_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = prev_end = rows[a][2]
busy = 0.0
for name, s, e, gx, wx in rows[a + 1:b + 1]:
    print("%8.1f  gap %6.1f  dur %6.1f  wgs %6d x %4d  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, gx // max(wx, 1), wx, short(name)))
    busy += max(0, e - max(s, prev_end))
    prev_end = max(prev_end, e)
print("step span %.1f us, device busy %.1f us, %d launches" % ((rows[b][2] - rows[a][2]) / 1e3, busy / 1e3, b - a))
