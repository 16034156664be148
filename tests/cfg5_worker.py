"""Runs in a FRESH process (tests/test_multi_gpu.py::test_cfg5_shape_...): BASELINE.json's cfg5 — 512 concurrent
44.1 kHz stereo streams through a 256 k-tap filter, 64 per GPU over 8 GPUs via ProcessorPool — on a ONE-GPU box:
FOLVE_AMD_DEVICES=0,0,0,0,0,0,0,0 gives the router eight slots (eight engines, eight combiners, eight copies of the
filter) on device 0, which exercises everything of the 8-GPU path except seven more physical devices.  Then one slot's
engine starts failing (FE_TUNE_FAIL_NEXT = -1) and the router must fence it: new files go to the other seven, no open
returns NULL, the pool discards that slot's processors only (the reference's discard-and-recreate loop,
/root/reference/processor-pool.cc:71-77; its pass-through fallback when no processor can be had,
folve-filesystem.cc:78-88, must not be triggered by one bad GPU of eight).  Prints one JSON line."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "dropin"))

import numpy as np  # noqa: E402


def f64_convolution(x, taps):
    """Exact linear convolution in float64 (FFT), truncated to the input length; x [frames, 2], taps [2][n]."""
    n = x.shape[0]
    m = 1
    while m < n + len(taps[0]):
        m *= 2
    out = np.empty((n, 2), np.float64)
    for c in range(2):
        out[:, c] = np.fft.irfft(np.fft.rfft(x[:, c].astype(np.float64), m) * np.fft.rfft(taps[c].astype(np.float64), m), m)[:n]
    return out


def main():
    tmp = sys.argv[1]
    NSLOT, PER, THREADS = 8, 64, 64
    assert os.environ.get("FOLVE_AMD_DEVICES") == ",".join(["0"] * NSLOT)
    import folve_amd.capi as capi
    import folve_amd.host as H
    import make_conf
    size = 262144
    conf = make_conf.write(tmp, size)                   # cfg3 / cfg5's filter through the real loader
    d = os.path.dirname(conf)
    L = H._L()
    L.fh_router_health_policy(3, 0.3)
    pool = H.ProcessorPool(NSLOT * PER + 64)
    total = NSLOT * PER
    blocks = 24
    xs = [np.random.default_rng(100 + s).uniform(-1, 1, (blocks * 8192 + 1000 + 7 * s, 2)).astype(np.float32) for s in range(8)]
    procs = [None] * total
    fails = []

    def open_range(i0, i1, into, base=0):
        for i in range(i0, i1):
            p, err = pool.get_or_create(d, 44100, 2, 16)
            if p is None:
                fails.append(err)
            into[base + i] = p

    def threads(fn, n, *args):
        per = [threading.Thread(target=fn, args=(t,) + args) for t in range(n)]
        [t.start() for t in per]
        [t.join() for t in per]

    t0 = time.perf_counter()
    threads(lambda t: open_range(t * total // THREADS, (t + 1) * total // THREADS, procs), THREADS)
    t_open = time.perf_counter() - t0
    assert not fails, fails[:3]
    live = [L.fh_router_live_streams(s) for s in range(NSLOT)]
    engines = [int(L.fh_router_slot_engine(s) or 0) for s in range(NSLOT)]
    slot_of = {e: s for s, e in enumerate(engines)}
    per_slot = [0] * NSLOT
    for p in procs:
        per_slot[slot_of[int(L.fh_processor_engine(p.h))]] += 1
    depth = procs[0].run_ahead()

    # every stream converts a file at once (64 threads x 8 processors each, all 512 held): spot parity against float64
    outs = {}

    def run_range(t, which, keep):
        for i in range(t * len(which) // THREADS, (t + 1) * len(which) // THREADS):
            y = which[i].run(xs[i % 8])
            if i in keep:
                outs[i] = y

    keep = set(range(0, total, 37))
    t0 = time.perf_counter()
    threads(run_range, THREADS, procs, keep)
    t_run = time.perf_counter() - t0
    taps = []
    with open(os.path.join(d, "ir.wav"), "rb") as f:
        raw = np.frombuffer(f.read()[44:], "<i2").reshape(-1, 2)
    for c in range(2):
        taps.append((np.float32(2e-3) * (raw[:, c].astype(np.float32) / np.float32(32768.0))).astype(np.float32))
    refs = {}
    worst = 0.0
    for i, y in sorted(outs.items()):
        if i % 8 not in refs:
            refs[i % 8] = f64_convolution(xs[i % 8], taps)
        worst = max(worst, float(np.sqrt(np.mean((y.astype(np.float64) - refs[i % 8]) ** 2))))
    ok_before = sum(L.fh_processor_ok(p.h) for p in procs)
    states_before = [L.fh_router_slot_state(s) for s in range(NSLOT)]

    # ---- one GPU goes bad, in mid-conversion --------------------------------------------------------------------------
    BAD = 3
    for p in procs:
        p.reset()
    long_blocks = 96                                     # long enough that every file is still converting when the GPU dies
    xl = [np.random.default_rng(200 + s).uniform(-1, 1, (long_blocks * 8192 + 500 + 11 * s, 2)).astype(np.float32) for s in range(8)]
    on_bad = [i for i, p in enumerate(procs) if slot_of[int(L.fh_processor_engine(p.h))] == BAD]
    outs_bad = {}
    assert len(on_bad) == THREADS
    others = [i for i in range(total) if i not in set(on_bad)]
    gate = threading.Barrier(THREADS)
    KILL_AT = 20                                         # blocks of its file every bad-slot processor has returned when the GPU dies

    def run_long(t):
        # this thread's file on the bad slot first — block by block (the AddMoreSoundData loop, convolve-file-handler.cc:370-424),
        # so that all 64 of them are at block 20 of 96, chunks in flight, when the engine starts failing
        i = on_bad[t]
        p, x = procs[i], xl[i % 8]
        outs, done, blocks = [], 0, 0
        while done < x.shape[0]:
            r = p.fill_buffer(x[done:])
            assert r > 0
            outs.append(p.write_processed(r))
            done += r
            blocks += 1
            if blocks == KILL_AT and gate.wait(timeout=300) == 0:      # (a thread that died would otherwise leave the others here for good)
                assert L.fe_engine_set_tuning(engines[BAD], capi.FE_TUNE_FAIL_NEXT, -1) == 0
        outs_bad[i] = np.concatenate(outs, 0)
        # ... then its share of the other 448
        for k in range(t * len(others) // THREADS, (t + 1) * len(others) // THREADS):
            j = others[k]
            y = procs[j].run(xl[j % 8])
            if j % 41 == 0:
                outs_bad[j] = y

    t0 = time.perf_counter()
    threads(run_long, THREADS)
    t_run_bad = time.perf_counter() - t0
    refl = {}
    worst_bad, silent_blocks, peak_err = 0.0, 0, 0.0
    for i, y in sorted(outs_bad.items()):
        if i % 8 not in refl:
            refl[i % 8] = f64_convolution(xl[i % 8], taps)
        worst_bad = max(worst_bad, float(np.sqrt(np.mean((y.astype(np.float64) - refl[i % 8]) ** 2))))
        nblk = y.shape[0] // 8192
        silent_blocks += int(sum(not y[b * 8192:(b + 1) * 8192].any() for b in range(nblk)))
        peak_err = max(peak_err, abs(procs[i].max_output_value() - max(0.0, float(y.max()))))
    moves = [L.fh_processor_moves(procs[i].h) for i in on_bad]
    moved_mid = sum(1 for i in on_bad if L.fh_processor_moves(procs[i].h) >= 1)
    still_on_bad = sum(1 for p in procs if int(L.fh_processor_engine(p.h)) == engines[BAD])
    ok_after = sum(L.fh_processor_ok(p.h) for p in procs)
    states_bad = [L.fh_router_slot_state(s) for s in range(NSLOT)]
    failures_bad = L.fh_router_slot_failures(BAD)
    live_after_move = [L.fh_router_live_streams(s) for s in range(NSLOT)]
    # new files while it is fenced: never NULL, never there
    more = [None] * 56
    threads(lambda t: open_range(t * 7, (t + 1) * 7, more), 8)
    more_on_bad = sum(1 for p in more if p is not None and int(L.fh_processor_engine(p.h)) == engines[BAD])
    more_null = sum(1 for p in more if p is None)
    live_more = [L.fh_router_live_streams(s) for s in range(NSLOT)]
    y = more[0].run(xs[0])
    rms_more = float(np.sqrt(np.mean((y.astype(np.float64) - refs.setdefault(0, f64_convolution(xs[0], taps))) ** 2)))
    # everything back to the pool: nothing lives on the bad slot any more, nothing is discarded
    for p in procs + more:
        pool.give_back(p)
    pooled = pool.pooled_count(conf)
    live_pooled = [L.fh_router_live_streams(s) for s in range(NSLOT)]
    # the pool hands out nothing that lives on the fenced slot
    again = [None] * 128
    threads(lambda t: open_range(t * 16, (t + 1) * 16, again), 8)
    again_on_bad = sum(1 for p in again if p is None or int(L.fh_processor_engine(p.h)) == engines[BAD])

    # ---- and recovers ------------------------------------------------------------------------------------------------
    assert L.fe_engine_set_tuning(engines[BAD], capi.FE_TUNE_FAIL_NEXT, 0) == 0
    time.sleep(0.4)
    for p in again:
        pool.give_back(p)
    pooled_cfg = pool.pooled_count(conf)
    # drain the pool so that the next opens are Creates, which ask the router
    drained = []
    while pool.pooled_count(conf) > 0:
        p, _ = pool.get_or_create(d, 44100, 2, 16)
        drained.append(p)
    back = [None] * 8
    open_range(0, 8, back)
    state_back = L.fh_router_slot_state(BAD)
    back_on_bad = sum(1 for p in back if p is not None and int(L.fh_processor_engine(p.h)) == engines[BAD])
    y = back[0].run(xs[1])
    rms_back = float(np.sqrt(np.mean((y.astype(np.float64) - refs.setdefault(1, f64_convolution(xs[1], taps))) ** 2)))
    stats = H.batching_stats()
    out = {"slots": L.fh_router_device_count(), "live": live, "per_slot": per_slot, "distinct_engines": len(set(engines)),
           "run_ahead": depth, "open_s": round(t_open, 2), "run_s": round(t_run, 2), "run_bad_s": round(t_run_bad, 2),
           "msamples_per_s": round(total * (blocks * 8192) * 2 / t_run / 1e6, 1),
           "checked": len(outs), "max_rms": worst, "ok_before": ok_before, "states_before": states_before,
           "ok_after": ok_after, "states_bad": states_bad, "failures_bad": failures_bad, "files_on_bad": len(on_bad),
           "moved": moved_mid, "max_moves": max(moves) if moves else 0, "still_on_bad": still_on_bad, "checked_bad_phase": len(outs_bad),
           "max_rms_bad_phase": worst_bad, "silent_blocks": silent_blocks, "peak_err": peak_err, "live_after_move": live_after_move,
           "more_null": more_null, "more_on_bad": more_on_bad, "live_more": live_more, "rms_more": rms_more,
           "pooled": pooled, "live_pooled": live_pooled, "again_on_bad": again_on_bad, "pooled_cfg": pooled_cfg,
           "state_back": state_back, "back_on_bad": back_on_bad, "rms_back": rms_back,
           "cached_filters": L.fh_router_cached_filters(), "batching": stats, "bad_slot": BAD}
    print("CFG5_JSON " + json.dumps(out))
    sys.stdout.flush()
    os._exit(0)          # (hundreds of pinned rings: skip the interpreter's teardown order)


if __name__ == "__main__":
    main()
