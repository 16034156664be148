"""Runs the large-batch parity test (tests/test_walkers_gpu.py: fast kernel forms against the oracle, the
float64 convolution and the general kernels) over random shapes (dev aid; GPU).
usage: python tests/soak/soak_batches.py [cases]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import folve_amd as fa
from oracle import oracle as O
import test_walkers_gpu as T

eng = fa.Engine(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad, t0 = 0, time.time()
for c in range(cases):
    channels = int(rng.integers(1, 3))
    size = int(rng.choice([1500, 3000, 4000, 9000, 20000, 70000, 150000, 262144]))
    nblocks = int(rng.integers(4, 71))
    nstreams = max(int(rng.integers(6, 24)), -(-256 // nblocks))
    try:
        T.test_large_batches_match_oracle_and_general_kernels(eng, O, channels, size, nstreams, nblocks)
    except Exception as e:
        bad += 1
        print("case", c, (channels, size, nstreams, nblocks), "FAILED", repr(e)[:300])
print("soak done: %d cases, failures: %d, %.1f s" % (cases, bad, time.time() - t0))
