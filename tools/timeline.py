#!/usr/bin/env python3
"""Timeline of the last calls of a rocprofv3 --kernel-trace run: per dispatch start (relative), duration, and the gap
since the previous dispatch ended.   usage: timeline.py <dir with *kernel_trace.csv> [dispatches to show]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n - 60:-60] if len(rows) > n + 60 else rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:46]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-46s start %8.1f us  dur %7.1f us  gap %6.1f us  grid %s wg %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, gap,
          r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    prev_end = max(prev_end or e, e)
