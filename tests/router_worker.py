"""Runs in a FRESH process (tests/test_multi_gpu.py): two router slots on one GPU
(FOLVE_AMD_DEVICES=0,0 must be in the environment before anything touches the GPU), a
ProcessorPool driven from 8 threads.  Prints one JSON line."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    tmp = sys.argv[1]
    assert os.environ.get("FOLVE_AMD_DEVICES") == "0,0"
    import folve_amd.host as H
    from fixtures import make_santalucia_shaped_dir, seeded_input
    from oracle import oracle as O
    d, hs = make_santalucia_shaped_dir(tmp)
    conf = os.path.join(d, "filter-44100.conf")
    L = H._L()
    pool = H.ProcessorPool(8)
    n = 8
    procs = [None] * n
    placed = []
    lock = threading.Lock()
    barrier = threading.Barrier(n)
    rms = [None] * n

    def work(i):
        p, err = pool.get_or_create(d, 44100, 2, 16)
        assert p is not None, err
        with lock:
            placed.append((L.fh_router_live_streams(0), L.fh_router_live_streams(1)))
        procs[i] = p
        barrier.wait()                          # all 8 processors are held at once
        x = seeded_input(200 + i, 3 * 8192 + 17 * i, 2)
        y = p.run(x)
        rms[i] = O.rms(y - O.linear_convolution_f64(x, hs, 2))

    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    live_held = [L.fh_router_live_streams(0), L.fh_router_live_streams(1)]
    engines = sorted({int(L.fh_processor_engine(p.h)) for p in procs})
    per_engine = [sum(1 for p in procs if int(L.fh_processor_engine(p.h)) == e) for e in engines]
    filters = L.fh_router_cached_filters()
    for p in procs:
        pool.give_back(p)
    out = {"slots": L.fh_router_device_count(), "live_while_held": live_held, "engines": len(engines),
           "per_engine": per_engine, "cached_filters": filters, "max_rms": max(rms),
           "placement_steps": placed, "pooled": pool.pooled_count(conf),
           "batching": H.batching_stats()}
    print("ROUTER_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
