"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a small JSON summary.

Dispatches are grouped by (kernel, grid size) so that the run-ahead launches
(T blocks per stream) and the streaming launches (1 block) of one bench run are
kept apart.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-B requests as 64 B for wide streaming reads (MI355X_MICROARCH.md §HBM), so
`hbm_read_bytes_corrected` doubles it.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {"kernel_trace": {}, "kernel_stats": {}, "pmc": {}}


def short(name):
    for k in ("forward_dual_kernel", "make_g_kernel", "forward_walker_kernel", "inverse_walker_kernel", "forward_chpair_kernel",
              "inverse_chpair_kernel", "forward_pair_kernel", "inverse_pair_kernel", "mac_slide_kernel", "mac_walk3_nt_kernel", "mac_walk3_kernel", "mac_walk_kernel",
              "mac_small_kernel", "forward_kernel", "mac_kernel", "inverse_kernel", "filter_kernel"):
        if k in name:
            t = name.split(k)[1].split(">")[0].strip("<")
            return "%s<%s>" % (k, t)
    return None


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = short(row["Name"])
        if n:
            res["kernel_stats"][n] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                      "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]),
                                      "pct": float(row["Percentage"])}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        n = short(row["Kernel_Name"])
        if n:
            if "Grid_Size" in row:
                grid = row["Grid_Size"]
            else:                                            # kernel-trace csv: per-dimension thread counts
                grid = str(int(row.get("Grid_Size_X", 1)) * int(row.get("Grid_Size_Y", 1)) * int(row.get("Grid_Size_Z", 1)))
            acc["%s grid=%s" % (n, grid)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items():
        res["kernel_trace"][k] = {"dispatches": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            n = short(row["Kernel_Name"])
            if not n:
                continue
            key = "%s grid=%s" % (n, row.get("Grid_Size", "?"))
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
    res["pmc"][c] = {n: {"dispatches": v[1], "avg_KiB_per_dispatch": v[0] / max(1, v[1])} for n, v in acc.items()}
hbm = {}
for key, w in res["pmc"].get("WRITE_SIZE", {}).items():
    fch = res["pmc"].get("FETCH_SIZE", {}).get(key)
    if fch:
        rd = fch["avg_KiB_per_dispatch"] * 1024 * 2
        wr = w["avg_KiB_per_dispatch"] * 1024
        hbm[key] = {"hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
res["hbm_per_dispatch"] = hbm
# (tools/profile.sh deletes csv files over 3 MB before the results travel back: a long --only-config run's kernel_trace.csv
# can be among them.  Its kernels ran on ONE grid each, so the stats csv's per-kernel averages are the same figures.)
for key in hbm:
    name = key.split(" grid=")[0]
    if key not in res["kernel_trace"] and name in res["kernel_stats"] and not any(k.startswith(name + " grid=") for k in res["kernel_trace"]):
        st = res["kernel_stats"][name]
        res["kernel_trace"][key] = {"dispatches": st["calls"], "avg_ns": st["avg_ns"], "min_ns": st["min_ns"], "max_ns": st["max_ns"],
                                    "source": "kernel_stats.csv of the same trace run (one grid per kernel)"}
print(json.dumps(res, indent=1))
