"""Dev aid: cfg3's batch with K2's walk in its plain and in its streaming (non-temporal rows) form, alternating on one box:
FE_TUNE_WALK_NT = 1 / 2.  Per-kernel dispatch times, the call, socket power and shader clock.  usage: python tools/ab_walk_nt.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
acc = {1: [], 2: []}
for r in range(rounds):
    for nt in (1, 2):
        m = measure_config(S=64, C=2, size=262144, T=256, steps=200, warmup=30, check=False, tune={"walk_nt": nt})
        k, p = m["kernels_ms"], m["power"] or {}
        acc[nt].append((m["ms_per_call"], k["mac"]))
        print("%s: %.4f ms/call  K1 %.4f K2 %.4f K3 %.4f  %6.1f Gsamples/s  %s MHz %s W  (%s)" % (
            "plain    " if nt == 1 else "streaming", m["ms_per_call"], k["forward"], k["mac"], k["inverse"], m["msamples_per_s"] / 1e3,
            p.get("sclk_mhz"), p.get("socket_w"), m["kernels_launched"]["mac"]), flush=True)
for nt in (1, 2):
    n = len(acc[nt])
    print("%s mean: call %.4f ms, K2 %.4f ms" % ("plain    " if nt == 1 else "streaming", sum(a for a, _ in acc[nt]) / n, sum(b for _, b in acc[nt]) / n))
