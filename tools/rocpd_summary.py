"""Summarise a rocprofv3 rocpd sqlite database: per-kernel stats (like --stats) and a per-dispatch table of the last step.
    python tools/rocpd_summary.py gpurun_out/prof/run_results.db [--dispatches N]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("papc::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall() \
        if "grid_x" in cols else None
    if rows is None:
        print("columns:", cols)
        return
    agg = {}
    for name, s, e, gx, gy, gz, wx in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(short(name), [0, 0.0, 1e18, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("%-112s %6s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-112s %6d %10.1f %9.1f %9.1f %9.1f %6.2f" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    if "--dispatches" in sys.argv:
        n = int(sys.argv[sys.argv.index("--dispatches") + 1])
        print("\nlast %d dispatches:" % n)
        for name, s, e, gx, gy, gz, wx in rows[-n:]:
            print("%9.1f us  grid=(%d,%d,%d) wg=%d  %s" % ((e - s) / 1e3, gx, gy, gz, wx, short(name)))


if __name__ == "__main__":
    main()
