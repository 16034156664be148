"""The drop-in host layer under random load (dev aid; GPU): rounds of file threads, each its own pooled SoundProcessor with
a random run-ahead depth, files of random length drained in uneven pieces (half of the processors then reset and reused for
a longer file), every output against the float64 convolution.
usage: python tests/soak/soak_host.py [rounds] [threads] [seed]"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from folve_amd import host as H
from fixtures import make_santalucia_shaped_dir, make_echo_filter_dir
from oracle import oracle as O
import tempfile

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 7)
tmp = tempfile.mkdtemp(prefix="folve_soak_")
d, hs = make_santalucia_shaped_dir(tmp)
pool = H.ProcessorPool(3)
bad, t0, nfiles = 0, time.time(), 0
for r in range(rounds):
    depth = int(rng.choice([1, 2, 3, 8, 32, 64, 128]))
    H.set_run_ahead(depth)
    jobs = []
    for t in range(nthreads):
        na = int(rng.integers(1, 120 * 8192))
        nb = int(rng.integers(1, 40 * 8192)) if rng.random() < 0.5 else 0         # a second file after a Reset()
        jobs.append((rng.uniform(-1, 1, (na, 2)).astype(np.float32), rng.uniform(-1, 1, (nb, 2)).astype(np.float32) if nb else None))
    outs = [None] * nthreads
    errs = []

    def work(i):
        try:
            a, b = jobs[i]
            p, err = pool.get_or_create(d, 44100, 2, 16)
            assert p is not None, err
            ya = []
            done = 0
            while done < len(a):                                   # AddMoreSoundData, draining in uneven pieces
                n = p.fill_buffer(a[done:])
                assert n > 0
                cut = int(rng.integers(0, n + 1))
                if cut: ya.append(p.write_processed(cut))
                if n - cut: ya.append(p.write_processed(n - cut))
                done += n
            if b is not None:
                # a second file on the same (pooled, reset) processor: the concatenation through run()
                p.reset()
                both = np.concatenate([a, b])
                y = p.run(both)
                outs[i] = (both, y)
            else:
                outs[i] = (a, np.concatenate(ya))
            pool.give_back(p)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        bad += len(errs)
        print("round", r, "depth", depth, "ERRORS", errs[:3])
        continue
    for i, (x, y) in enumerate(outs):
        nfiles += 1
        e = O.rms(y - O.linear_convolution_f64(x, hs, 2))
        if not (len(y) == len(x) and e <= 1e-5):
            bad += 1
            print("round", r, "depth", depth, "file", i, "frames", len(x), "FAILED rms", e)
H.set_run_ahead(H.AUTO_RUN_AHEAD)
print("soak_host done: %d rounds x %d threads, %d files, failures: %d, %.1f s; combiner %s" % (rounds, nthreads, nfiles, bad, time.time() - t0, H.batching_stats()))
