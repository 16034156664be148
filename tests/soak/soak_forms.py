"""Random shapes through every K2 form and walker run length, each against the general kernels on the same
input (dev aid; GPU).  usage: python tests/soak/soak_forms.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import folve_amd as fa

eng = fa.Engine(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad, t0 = 0, time.time()
for c in range(cases):
    size = int(rng.choice([300, 700, 1500, 3000, 4096, 9000, 20000, 60000, 65536, 100000, 131072, 200000, 262144, 270000, 524288, 600000, 1048576]))
    ch = int(rng.choice([1, 2, 2, 4, 6, 8]))                 # (4, 6, 8: the channel-pair K1 / K3)
    S = int(rng.integers(1, 7))
    T = int(rng.integers(1, 90)) if size <= 270000 else int(rng.integers(1, 200))
    flt = fa.Filter(eng, ch, ch, size)
    matrix = ch > 1 and rng.random() < 0.4                  # several paths per output: the walk's lane sets (up to four)
    for k in range(ch):
        ins = [k]
        if matrix:
            ins = sorted(set(int(i) for i in rng.choice(ch, size=int(rng.integers(1, min(ch, 5) + 1)), replace=False)))
        for i in ins:
            n = int(rng.integers(1, size + 1))
            flt.add(i, k, (rng.standard_normal(n) / np.sqrt(n * len(ins))).astype(np.float32), int(rng.integers(0, size - n + 1)))
    flt.commit()
    P = flt.block_size
    calls = []
    for _ in range(3):
        calls.append([rng.uniform(-1, 1, (int(rng.integers(1, T * P + 1)), ch)).astype(np.float32) for _ in range(S)])
    maxb = max(-(-x.shape[0] // P) for call in calls for x in call)
    outs = {}
    runlen = int(rng.choice([1, 2, 3, 4, 8, 16, 32]))
    lpb, tiles = int(rng.choice([0, 1, 2, 4])), int(rng.choice([0, 1, 2, 3, 7]))       # the walk's lanes per bin and time tiles
    base = dict(walk_lpb=0, walk_tiles=0, walk_fma=0, walk_nt=0)
    for name, knobs in (("general", dict(mac_form=1, fft_form=1)), ("walk", dict(mac_form=100, fft_form=2, fwd_run=runlen, inv_run=runlen, walk_lpb=lpb, walk_tiles=tiles, walk_fma=3)),
                        ("walk4", dict(mac_form=100, fft_form=2, fwd_run=runlen, inv_run=runlen, walk_lpb=lpb, walk_tiles=tiles, walk_fma=4)),   # (the four-FMA form of the walk)
                        ("walknt", dict(mac_form=100, fft_form=2, fwd_run=runlen, inv_run=runlen, walk_lpb=1, walk_tiles=tiles, walk_fma=3, walk_nt=2)),   # (the streaming form: every one-lane rung has one)
                        ("slide16", dict(mac_form=16, fft_form=2, fwd_run=0, inv_run=0)), ("auto", dict(mac_form=0, fft_form=0, fwd_run=0, inv_run=0))):
        eng.set_tuning(**dict(base, **knobs))
        st = [flt.open_stream(maxb) for _ in range(S)]
        outs[name] = [fa.batch_process(st, call) for call in calls]
    eng.set_tuning(mac_form=0, fft_form=0, fwd_run=0, inv_run=0, walk_lpb=0, walk_tiles=0, walk_fma=0, walk_nt=0)
    worst = 0.0
    for name in ("walk", "walk4", "walknt", "slide16", "auto"):
        for a, b in zip(outs["general"], outs[name]):
            for x, y in zip(a, b):
                if x.size:
                    worst = max(worst, float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2))))
    if not worst <= 2e-6:
        bad += 1
        print("case", c, (size, ch, S, T, runlen, lpb, tiles, matrix), "FAILED rms", worst)
print("soak_forms done: %d cases, failures: %d, %.1f s" % (cases, bad, time.time() - t0))
