"""Kernel timeline of tools/dropin/dropin_threads under rocprofv3 --kernel-trace: per launch round (K1, K2, K3)
the kernel durations and the idle time of the GPU before the round (medians)."""
import csv, glob, statistics as st, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][:36],
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
rows = rows[len(rows) // 3:]
rounds = []
i = 0
while i + 2 < len(rows):
    a, b, c = rows[i], rows[i + 1], rows[i + 2]
    if "forward" in a[2] and "mac" in b[2] and "inverse" in c[2]:
        rounds.append((a, b, c)); i += 3
    else:
        i += 1
med = lambda v: st.median(v) / 1e3
print("rounds", len(rounds), "| kernels:", rounds[-1][0][2], "/", rounds[-1][1][2], "/", rounds[-1][2][2])
print("K1 %.1f us  K2 %.1f us  K3 %.1f us  | K1 start -> K3 end %.1f us | GPU idle before a round %.1f us | round period %.1f us"
      % (med([a[1] - a[0] for a, b, c in rounds]), med([b[1] - b[0] for a, b, c in rounds]), med([c[1] - c[0] for a, b, c in rounds]),
         med([c[1] - a[0] for a, b, c in rounds]), med([rounds[j + 1][0][0] - rounds[j][2][1] for j in range(len(rounds) - 1)]),
         med([rounds[j + 1][0][0] - rounds[j][0][0] for j in range(len(rounds) - 1)])))
