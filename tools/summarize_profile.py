"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {"kernels": {}, "pmc": {}}


def short(name):
    for k in ("forward_kernel", "mac_kernel", "inverse_kernel", "filter_kernel"):
        if k in name:
            return k
    return name[:60]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = short(row["Name"])
        res["kernels"][n] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                             "total_ns": float(row["TotalDurationNs"]), "pct": float(row["Percentage"])}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            n = short(row["Kernel_Name"])
            acc[n][0] += float(row["Counter_Value"])
            acc[n][1] += 1
    res["pmc"][c] = {n: {"sum": v[0], "dispatches": v[1], "avg_per_dispatch": v[0] / max(1, v[1])} for n, v in acc.items()}
print(json.dumps(res, indent=1))
