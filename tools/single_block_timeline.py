"""Timeline of the one-block call from a rocprofv3 --kernel-trace CSV of tools/single_block.py:
every kernel of a call with its duration and the gap to its predecessor (medians over the steady loop)."""
import csv
import glob
import statistics as st
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][:40]))
rows.sort()
rows = rows[len(rows) // 2:]                       # the steady loop
# a call starts at every kernel whose predecessor ended more than 8 us earlier
calls, cur = [], []
for k, r in enumerate(rows):
    if cur and r[0] - cur[-1][1] > 8000:
        calls.append(cur)
        cur = []
    cur.append(r)
calls = [c for c in calls[1:] if len(c) == len(calls[1])]
n = len(calls[0])
print("calls:", len(calls), "kernels per call:", n)
for i in range(n):
    dur = st.median(c[i][1] - c[i][0] for c in calls) / 1e3
    gap = st.median(c[i][0] - c[i - 1][1] for c in calls) / 1e3 if i else 0.0
    print("  %-40s gap %5.1f us  duration %5.1f us" % (calls[0][i][2], gap, dur))
print("  first start -> last end %.1f us;  last end -> next call's first start %.1f us;  period %.1f us"
      % (st.median(c[-1][1] - c[0][0] for c in calls) / 1e3,
         st.median(calls[j + 1][0][0] - calls[j][-1][1] for j in range(len(calls) - 1)) / 1e3,
         st.median(calls[j + 1][0][0] - calls[j][0][0] for j in range(len(calls) - 1)) / 1e3))
