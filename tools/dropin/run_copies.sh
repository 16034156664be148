#!/bin/bash
# rocprofv3 --memory-copy-trace --kernel-trace of the drop-in thread harness: how busy each direction of the bus is while
# N file threads convert (the second half of the run): union of the host->device copies, of the device->host copies, both
# at once, and the gaps.   usage: run_copies.sh <threads> <blocks> <run_ahead>     -> gpurun_out/copies_<threads>x<run_ahead>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
NT=${1:-64}; NB=${2:-4096}; RA=${3:-64}
OUT=$R/gpurun_out/copies_${NT}x$RA.txt
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/dropin/make_conf.py /tmp/copies_cfg 262144 > /dev/null
rm -rf /tmp/tr_copies
timeout 600 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d /tmp/tr_copies -- $R/tools/dropin/dropin_threads /tmp/copies_cfg/filter-44100.conf $NT $NB 1 json run_ahead=$RA > /tmp/tr_copies.log 2>&1
python3 - <<PY > $OUT
import csv, glob, json
def union(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
def total(iv): return sum(b - a for a, b in iv)
def inter(x, y):
    i = j = t = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a: t += b - a
        if x[i][1] < y[j][1]: i += 1
        else: j += 1
    return t
rows = [r for f in glob.glob("/tmp/tr_copies/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
kern = [r for f in glob.glob("/tmp/tr_copies/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
line = [l for l in open("/tmp/tr_copies.log").read().splitlines() if l.startswith("{")]
print("dropin_threads $NT threads x $NB blocks, run_ahead=$RA under rocprofv3 --memory-copy-trace --kernel-trace")
if line: print("harness:", {k: v for k, v in json.loads(line[-1]).items() if k in ("blocks_per_s", "threads", "run_ahead", "largest_batch_blocks", "requests", "batches")})
szcol = next((c for c in (rows[0].keys() if rows else []) if "size" in c.lower() or "bytes" in c.lower()), None)
def nbytes(r): return int(r.get(szcol, 0) or 0) if szcol else 0
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nbytes(r)) for r in rows if "HOST_TO_DEVICE" in r.get("Direction", "").upper() or "H2D" in r.get("Direction", "").upper()]
d2h = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nbytes(r)) for r in rows if "DEVICE_TO_HOST" in r.get("Direction", "").upper() or "D2H" in r.get("Direction", "").upper()]
print("(copy-trace columns: %s)" % ", ".join(rows[0].keys()) if rows else "")
if not h2d or not d2h:
    print("columns:", list(rows[0].keys()) if rows else "no rows"); raise SystemExit
t0 = sorted(a for a, _, _ in h2d)[len(h2d) // 2]
t1 = max(b for _, b, _ in d2h)
H = union([(max(a, t0), b) for a, b, _ in h2d if b > t0]); D = union([(max(a, t0), b) for a, b, _ in d2h if b > t0])
K = union([(max(int(r["Start_Timestamp"]), t0), int(r["End_Timestamp"])) for r in kern if int(r["End_Timestamp"]) > t0])
span = t1 - t0
bh = sum(n for a, b, n in h2d if a >= t0); bd = sum(n for a, b, n in d2h if a >= t0)
print("second half of the run: %.1f ms" % (span / 1e6))
# (this rocprofv3's copy trace has no size column: the bytes per second each way follow from the harness's own block rate —
# 8192 stereo float frames = 64 KiB per block and direction)
rate = json.loads(line[-1])["blocks_per_s"] * 65536 / 1e9 if line else 0.0
print("host->device copies busy %.1f %% of it: %.1f GB/s over the run = %.1f GB/s while busy (%d copies)" % (100 * total(H) / span, rate, rate * span / max(total(H), 1), sum(1 for a, _, _ in h2d if a >= t0)))
print("device->host copies busy %.1f %% of it: %.1f GB/s over the run = %.1f GB/s while busy" % (100 * total(D) / span, rate, rate * span / max(total(D), 1)))
print("both directions at once %.1f %%, neither %.1f %%; kernels running %.1f %%" % (100 * inter(H, D) / span, 100 * (span - total(union(H + D))) / span, 100 * total(K) / span))
PY
cat $OUT
