"""Throughput and per-kernel times of the other BASELINE.json configurations (bench.py's `configs` leg uses the same
function).  Dev aid:  python tools/config_rates.py [blocks per call] [tune, e.g. mac_form=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.configs import measure_config

if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tune = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[2].split(","))} if len(sys.argv) > 2 else None
    for name, kw in (("cfg2", dict(S=1, C=2, size=204800, populated=178193)),
                     ("cfg4", dict(S=1, C=8, size=524288)),
                     ("maxsize mono", dict(S=1, C=1, size=1048576)),
                     ("cfg3", dict(S=64, C=2, size=262144))):
        for t in ((T, 32) if name != "cfg3" else (T,)):
            r = measure_config(T=t, tune=tune, steps=50, check=False, **kw)
            P, K = r["block"], r["partitions"]
            minb = (4 * P + 8 * P + 8 * P * (t + K) / t + 8 * P + 8 * P + 4 * P) * r["channels"] * t * r["streams"]
            print("%-13s T=%3d K=%3d(%3d): %8.3f ms/call  K1 %.3f K2 %.3f K3 %.3f ms  %9.1f Msamples/s  min-bytes/t = %.2f TB/s"
                  % (name, t, K, r["populated_partitions"], r["ms_per_call"], r["kernels_ms"]["forward"], r["kernels_ms"]["mac"],
                     r["kernels_ms"]["inverse"], r["msamples_per_s"], minb / (r["ms_per_call"] * 1e-3) / 1e12))
