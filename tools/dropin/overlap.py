"""Bus-direction overlap of tools/dropin/dropin_threads under rocprofv3 --kernel-trace: how long forward kernels (PCM
in over the bus), inverse kernels (PCM out) and both at once were running, over the second half of the trace.
   usage: overlap.py <rocprofv3 output dir> [blocks per second from the harness]"""
import csv, glob, sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Queue_Id", 0) or 0),
                     int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)))
rows.sort()
t_lo = rows[len(rows) // 2][0]
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def inter(x, y):
    i = j = 0
    t = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a:
            t += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return t


fwd = union([(a, b) for a, b, n, q, g in rows if "forward" in n])
inv = union([(a, b) for a, b, n, q, g in rows if "inverse" in n])
mac = union([(a, b) for a, b, n, q, g in rows if "mac" in n])
anyk = union([(a, b) for a, b, n, q, g in rows])
print("kernels %d over %.2f ms; queues used: %s" % (len(rows), span / 1e6, sorted({q for _, _, _, q, _ in rows})))
print("busy: any %.1f %%  forward %.1f %%  inverse %.1f %%  mac %.1f %%  forward&inverse at once %.1f %%"
      % (100 * total(anyk) / span, 100 * total(fwd) / span, 100 * total(inv) / span, 100 * total(mac) / span, 100 * inter(fwd, inv) / span))
names = {}
for a, b, n, q, g in rows:
    d = names.setdefault(n, [0, 0, 0])
    d[0] += 1; d[1] += b - a; d[2] += g
for n, (c, t, g) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s launches %5d  avg %8.1f us  avg grid %8.0f threads" % (n, c, t / c / 1e3, g / c))
if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    t0 = rows[0][0]
    for a, b, n, q, g in rows[:36]:
        print("   q%-2d %-24s %8.1f -> %8.1f us  (%5.1f us, %6d threads)" % (q, n, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, g))
