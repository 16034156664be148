"""Runs the randomized call-pattern parity test (tests/test_fuzz_gpu.py) over many more seeds than the
suite does (dev aid; GPU).  usage: python tests/soak/soak_fuzz.py [last_seed]   — 5000 seeds take ~25 s."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import folve_amd as fa
from oracle import oracle as O
import test_fuzz_gpu as T
eng = fa.Engine(0)
bad = 0
import time
t0 = time.time()
for seed in range(7, int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    try:
        T.test_random_filters_and_call_patterns.__wrapped__(eng, O, seed) if hasattr(T.test_random_filters_and_call_patterns, "__wrapped__") else T.test_random_filters_and_call_patterns(eng, O, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:200])
print("soak done, failures:", bad, "seconds: %.1f" % (time.time() - t0))
