"""profiles/<tag>_summary.json (tools/summarize_profile.py) -> the entries of profiles/traffic.json
bench.py reads: per shape, HBM bytes per launch of K1 / K2 / K3 (FETCH_SIZE x 2 + WRITE_SIZE, see the
note written into the file) and the kernel-trace average durations of the same launches.

usage: python tools/make_traffic.py <tag> <streams> <blocks> <K> <channels> [--run-ahead-only] [--full]
(--full: the profiled filter was a full matrix — bench.py --only-config matrix; the key gets the suffix `_full`)
The run-ahead launches (T blocks per call) and the one-block-per-call launches of the same bench run
are told apart by kernel: walkers / mac_walk / mac_slide vs forward_kernel / mac_kernel<1> / inverse_kernel.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, S, T, K, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
summ = json.load(open(os.path.join(ROOT, "profiles", tag + "_summary.json")))
ROLES = {
    # (a lone stereo stream's call of fewer than 512 blocks takes the per-channel general kernels: the last names)
    "T": {"forward": ("forward_walker_kernel", "forward_chpair_kernel", "forward_dual_kernel", "forward_kernel"),
          "mac": ("mac_walk3_nt_kernel", "mac_walk3_kernel", "mac_walk_kernel", "mac_slide_kernel"),
          "inverse": ("inverse_walker_kernel", "inverse_chpair_kernel", "inverse_kernel")},
    "1": {"forward": ("forward_kernel",), "mac": ("mac_kernel<1>",), "inverse": ("inverse_kernel",)},
}


def pick_key(names):
    """The (kernel, grid) launch that stands for a role: of the first kernel in `names` that was launched at
    all, the grid on which it spent the most time — the S-stream launches of the timed and profiled loops
    (the same kernels also run on other grids in the single-block and end-to-end legs)."""
    trace = summ.get("kernel_trace", {})
    for n in names:
        cands = [(v["dispatches"] * v["avg_ns"], k) for k, v in trace.items() if k.startswith(n)]
        if cands:
            return max(cands)[1]
    return None


tpath = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
tj = {k: v for k, v in tj.items() if isinstance(v, dict) and "bytes" in v}       # drop entries of older formats
tj["_comment"] = ("HBM bytes per launch from rocprofv3 PMC passes over `python bench.py --no-cpu-baseline` (tools/profile.sh): "
                  "FETCH_SIZE x 2 + WRITE_SIZE.  The doubling is what MI355X_MICROARCH.md prescribes for coalesced streaming "
                  "reads on gfx950; it is checked in these very profiles on kernels whose bytes are known exactly: "
                  "mac_kernel<1> (16-byte loads: 277 MB of X rows + the 4.3 MB filter, counter 282 MB), the K3 walker "
                  "(8-byte loads: every Y row once = 537 MB, counter 514 MB) and K1's spectra stores (8-byte: 537 MB, WRITE_SIZE "
                  "513 MB): exact to 1 % for 16-byte accesses, 4-5 % low for 8-byte ones.  avg_ns: kernel-trace averages of the "
                  "same launches; bench.py uses an entry only for the same kernels (by name) while its own dispatch times agree with them (15 - 20 %).")
for mode, blocks in ((("T", T), ("1", 1)) if "--run-ahead-only" not in sys.argv else (("T", T),)):
    entry = {"profile": tag, "bytes": {}, "avg_ns": {}, "read": {}, "write": {}, "kernels": {}}
    for role, names in ROLES[mode].items():
        key = pick_key(names)
        if key is None:
            continue
        h = summ.get("hbm_per_dispatch", {}).get(key)
        t = summ.get("kernel_trace", {}).get(key)
        if h:
            entry["bytes"][role] = int(h["hbm_bytes"])
            entry["read"][role] = int(h["hbm_read_bytes_corrected"])
            entry["write"][role] = int(h["hbm_write_bytes"])
            entry["kernels"][role] = key
        if t:
            entry["avg_ns"][role] = round(t["avg_ns"], 1)
    if entry["bytes"]:
        tj["S%d_T%d_K%d_C%d" % (S, blocks, K, C) + ("_full" if "--full" in sys.argv else "")] = entry
json.dump(tj, open(tpath, "w"), indent=1)
print(json.dumps({k: v for k, v in tj.items() if k != "_comment"}, indent=1))
