#!/usr/bin/env python
"""Per-launch durations from a rocprofv3 kernel trace (csv): for every kernel whose name contains the
pattern, the launches in order with their grid and duration in microseconds, plus the gap to the
previous kernel of the trace.  usage: tools/trace_durations.py t_kernel_trace.csv forest_qr [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
n = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    if pat in name and n < mx:
        short = name[name.find(pat):][:60]
        print("%-62s grid %8s  %9.1f us  gap %7.1f us" % (short, r.get("Grid_Size_X", r.get("Grid_Size", "?")), (e - s) / 1e3,
                                                         (s - prev_end) / 1e3 if prev_end else 0.0))
        n += 1
    prev_end = e
