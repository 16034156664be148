#!/bin/bash
# rocprofv3 --marker-trace + --kernel-trace of the drop-in thread harness with the library's roctx ranges on
# (FOLVE_AMD_ROCTX=1: one range per launch round, one per chunk of the duplex DMA pipeline; trace.h) and the host-layer
# event log beside it (FOLVE_AMD_TRACE).  Run on the GPU box:  bash tools/dropin/run_markers.sh [threads] [blocks] [run_ahead]
# Leaves gpurun_out/markers/{markers.txt,host_events.txt}; copy them to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
NT=${1:-16}; NB=${2:-512}; RA=${3:-32}
OUT=$R/gpurun_out/markers
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/dropin/make_conf.py /tmp/markers_cfg 262144 > /dev/null
FOLVE_AMD_ROCTX=1 FOLVE_AMD_TRACE=$OUT/host_events_all.txt timeout 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $OUT/trace -- \
  $R/tools/dropin/dropin_threads /tmp/markers_cfg/filter-44100.conf $NT $NB 1 json run_ahead=$RA > $OUT/harness.log 2>&1
python3 - <<PY
import csv, glob, collections
out = "$OUT"
mk = [r for f in glob.glob(out + "/trace/**/*marker_api_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
kn = [r for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
with open(out + "/markers.txt", "w") as w:
    w.write("rocprofv3 --marker-trace --kernel-trace -- tools/dropin/dropin_threads <cfg3's filter> $NT threads x $NB blocks, run_ahead=$RA, FOLVE_AMD_ROCTX=1\n")
    w.write("%d roctx ranges, %d kernel dispatches\n\n" % (len(mk), len(kn)))
    if mk:
        cols = list(mk[0].keys())
        name = next((c for c in cols if c.lower() in ("function", "message", "name")), cols[0])
        start = next((c for c in cols if "start" in c.lower()), None)
        end = next((c for c in cols if "end" in c.lower()), None)
        kinds = collections.Counter(r[name].split(":")[0] for r in mk)
        w.write("ranges by kind: %s\n\n" % dict(kinds))
        t0 = min(int(r[start]) for r in mk) if start else 0
        w.write("first 40 ranges (us since the first, duration us, tid, message):\n")
        for r in sorted(mk, key=lambda r: int(r[start]) if start else 0)[:40]:
            w.write("%10.1f %8.1f %s %s\n" % ((int(r[start]) - t0) / 1e3, (int(r[end]) - int(r[start])) / 1e3, r.get("Thread_Id", r.get("Tid", "")), r[name]))
    by = collections.defaultdict(list)
    for r in kn:
        import re
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        by[m.group(1) if m else r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    w.write("\nkernels (dispatches, average us):\n")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w.write("%-60s %6d %9.1f\n" % (k[:60], len(v), sum(v) / len(v) / 1e3))
ev = open(out + "/host_events_all.txt").read().splitlines() if glob.glob(out + "/host_events_all.txt") else []
with open(out + "/host_events.txt", "w") as w:
    w.write("FOLVE_AMD_TRACE of the same run: %d events (<us since start> <thread id> <event> ...); the first 60 and the last 10\n" % len(ev))
    w.write("\n".join(ev[:60] + ["..."] + ev[-10:]) + "\n")
PY
rm -rf $OUT/trace $OUT/host_events_all.txt
tail -3 $OUT/harness.log | cut -c1-400; head -30 $OUT/markers.txt
