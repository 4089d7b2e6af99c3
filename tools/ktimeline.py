"""One steady-state step of a rocprofv3 --kernel-trace run as a timeline: start offset, duration, gap to the previous kernel's end, per queue.
    python tools/ktimeline.py <kernel_trace.csv> [anchor kernel substring = adam_kernel] [which occurrence from the end = 3]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]["End_Timestamp"])
prev_end = {}
busy = 0
for r in rows[i0 + 1:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    nm = r["Kernel_Name"].replace("papc::", "").replace("void ", "")[:70]
    print("q%-3s %8.1f us  dur %7.1f  gap %6.1f  grid %6s x %4s lds %6s  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Workgroup_Size_X"], r["LDS_Block_Size"], nm))
print("step span %.1f us" % ((int(rows[i1]["End_Timestamp"]) - t0) / 1e3))
