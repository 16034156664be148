"""profiles/traffic.json: HBM bytes per launch counted by rocprofv3 PMC passes of the bench commands (tools/profile_all.sh),
and the rule by which bench.py takes them — only for launches of the SAME kernels (by instantiation name) whose in-run
times agree with the profiled run's."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_traffic():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def norm_kernel(name):
    """'mac_walk_kernel<33, 7, true, 4, 1, 1> grid=1048576' (profiles/traffic.json) or the engine's own spelling
    (fe_engine_last_kernels) -> 'mac_walk_kernel<33,7,true,4,1,1>'."""
    return (name or "").split(" grid=")[0].replace(" ", "")


def profile_applies(entry, launched, kms, rel, floor_us, roles=None):
    """Is profiles/traffic.json's `entry` a profile of THIS run's launches?  Its kernels must be the ones the engine
    launched — by NAME (the instantiation, as rocprofv3 and fe_engine_last_kernels both spell it) — and this run's kernel
    times must agree with the profile's kernel-trace averages within `rel` (or `floor_us`: HIP events around a short launch
    read a few microseconds long).  Returns (ok, note)."""
    by = entry.get("bytes") or {}
    if not by:
        return False, None
    for k in (roles or by):
        prof, ran = norm_kernel((entry.get("kernels") or {}).get(k)), norm_kernel((launched or {}).get(k))
        if not prof or not ran or prof != ran:
            return False, ("profile %s is of other kernels (%s: profiled %s, launched %s): its traffic is not used"
                           % (entry.get("profile"), k, prof or "?", ran or "?"))
    for k in (roles or by):
        ns = (entry.get("avg_ns") or {}).get(k, 0)
        if abs(kms[k] * 1e6 - ns) > max(rel * ns, floor_us * 1e3):
            return False, ("in-run %s time %.1f us differs from profile %s's %.1f us by more than %.0f %%: its traffic is not used"
                           % (k, kms[k] * 1e3, entry.get("profile"), ns / 1e3, rel * 100))
    return True, None

def traffic_key(S, T, K, C, full=False):
    return "S%d_T%d_K%d_C%d" % (S, T, K, C) + ("_full" if full else "")

