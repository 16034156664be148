"""Builds tools/dropin/dropin_threads.cpp against the in-tree library and runs it over thread counts, with and without
the combiner (dev aid; GPU box).  The filter: cfg3's shape through the real loader — a jconvolver .conf reading a
262 144-frame stereo WAV.  This script never touches the GPU itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fixtures import write_wav

d = "/tmp/dropin_cfg3"
os.makedirs(d, exist_ok=True)
size = 262144
rng = np.random.default_rng(3)
ir = rng.standard_normal((size, 2))
ir /= np.abs(ir).max()
write_wav(os.path.join(d, "ir.wav"), np.round(ir * 32767 * 0.9).astype(np.int16), 44100)
with open(os.path.join(d, "filter-44100.conf"), "w") as f:
    f.write("/convolver/new 2 2 256 %d\n" % size)
    f.write("/impulse/read 1 1 2e-3 0 0 0 1 ir.wav\n/impulse/read 2 2 2e-3 0 0 0 2 ir.wav\n")
exe = "/tmp/dropin_threads"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tools", "dropin", "dropin_threads.cpp"), "-o", exe,
                       "-L" + os.path.join(ROOT, "folve_amd"), "-lfolve_amd", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "folve_amd")])
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for nt in (1, 2, 4, 8, 16, 32, 64, 128):
    for batching in (1, 0):
        r = subprocess.run([exe, os.path.join(d, "filter-44100.conf"), str(nt), str(blocks), str(batching)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "(no output) rc=%d" % r.returncode, flush=True)
