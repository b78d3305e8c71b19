"""Dev: a window of consecutive dispatches from a rocprofv3 --kernel-trace CSV, with queue ids and times relative to the first (us).
    python scripts/dev_trace_window.py <kernel_trace.csv> [start_fraction=0.6] [count=60] [name_filter_regex]"""
import csv, sys, re
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("mr::", "").replace("void ", "").split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 60
flt = re.compile(sys.argv[4]) if len(sys.argv) > 4 else None
if flt: rows = [r for r in rows if flt.search(r[2])]
i0 = int(len(rows) * frac)
t0 = rows[i0][0]
for s, e, n, q in rows[i0:i0 + cnt]:
    print("q%-3s %-44s start %9.1f  end %9.1f  dur %7.1f" % (q, n[:44], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
