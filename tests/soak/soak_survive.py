"""An open file survives its GPU, under random load (dev aid; GPU): FOLVE_AMD_DEVICES=0,0,0 gives the router three slots on
device 0; every round opens file threads (each its own SoundProcessor at a random run-ahead depth), lets them convert files
of random length through the SantaLucia-shaped filter (K = 25: 25 blocks of state to replay), and — at a random moment while
they run — makes one slot's engine fail every launch round (FE_TUNE_FAIL_NEXT = -1).  Every output must equal the float64
convolution with its peak, every processor must still be ok(); then the slot is healed and probed
back in for the next round.   usage: FOLVE_AMD_DEVICES=0,0,0 python tests/soak/soak_survive.py [rounds] [threads] [seed]"""
import os, sys, threading, time
assert os.environ.get("FOLVE_AMD_DEVICES") == "0,0,0", "run with FOLVE_AMD_DEVICES=0,0,0"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import tempfile
import numpy as np
import folve_amd.capi as capi
from folve_amd import host as H
from fixtures import make_santalucia_shaped_dir
from oracle import oracle as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
d, hs = make_santalucia_shaped_dir(tempfile.mkdtemp(prefix="folve_survive_"))
L = H._L()
L.fh_router_health_policy(3, 0.2)
conf = os.path.join(d, "filter-44100.conf")
bad = moved_total = files = 0
t0 = time.time()
for r in range(rounds):
    H.set_run_ahead(int(rng.choice([1, 2, 4, 16, 64])))
    procs = [H.SoundProcessor.create(conf, 44100, 2) for _ in range(nthreads)]
    assert all(p is not None for p in procs)
    engines = [int(L.fh_router_slot_engine(s) or 0) for s in range(3)]
    victim = int(rng.integers(0, 3))
    xs = [rng.uniform(-1, 1, (int(rng.integers(30, 110)) * 8192 + int(rng.integers(1, 8192)), 2)).astype(np.float32) for _ in range(nthreads)]
    outs = [None] * nthreads
    errs = []

    def work(i):
        try:
            outs[i] = procs[i].run(xs[i])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in th]
    time.sleep(float(rng.uniform(0.0, 0.06)))                # somewhere in the middle of the files (a file takes tens of milliseconds)
    assert L.fe_engine_set_tuning(engines[victim], capi.FE_TUNE_FAIL_NEXT, -1) == 0
    [t.join() for t in th]
    assert not errs, errs[:2]
    for i, p in enumerate(procs):
        ref = O.linear_convolution_f64(xs[i], hs, 2)
        e = O.rms(outs[i] - ref)
        mx = max(0.0, float(outs[i].max()))
        # (a file that had finished before the slot died stays where it was, unmoved: that is not a failure)
        if not (e <= 1e-5) or not L.fh_processor_ok(p.h) or abs(p.max_output_value() - mx) > 1e-6:
            bad += 1
            print("round", r, "file", i, "rms", e, "ok", L.fh_processor_ok(p.h), "moves", L.fh_processor_moves(p.h))
        moved_total += L.fh_processor_moves(p.h)
        files += 1
    for p in procs:
        p.close()
    assert L.fe_engine_set_tuning(engines[victim], capi.FE_TUNE_FAIL_NEXT, 0) == 0
    time.sleep(0.3)                                           # past the re-probe interval: the next opens look at the slot again
print("soak_survive done: %d rounds, %d files, %d moved, failures: %d, %.1f s" % (rounds, files, moved_total, bad, time.time() - t0))
sys.exit(1 if bad else 0)
