"""GPU: run-ahead in folve::SoundProcessor (SURVEY §8(f)2 — the BufferThread role, buffer-thread.cc:34,73-105):
a processor reads whole blocks ahead of its reader, computes them in multi-block engine requests through the
per-GPU combiner (two launch lanes), and serves FillBuffer / WriteProcessed from its ring with the reference's
per-call results (convolve-file-handler.cc:370-424).  Checked against the oracle's restated SoundProcessor call
for call, against the float64 convolution, and against the same processor with run-ahead off."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import folve_amd as fa
from folve_amd import host as H
from fixtures import make_echo_filter_dir, make_pass_filter_dir, make_santalucia_shaped_dir, seeded_input
from test_host_gpu import _drive_gapless

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROUNDING = 2e-6          # run-ahead on vs off: same arithmetic, another summation order in K2


@pytest.fixture
def depth():
    """Sets the run-ahead depth for processors created inside a test and restores the default afterwards."""
    def set_depth(n):
        H.set_run_ahead(n)
    yield set_depth
    H.set_run_ahead(H.AUTO_RUN_AHEAD)


def _call_log(sp, x, split=False):
    """The AddMoreSoundData loop, keeping every call's result: (frames read, complete?, pending before/after)."""
    log, outs, done = [], [], 0
    while done < len(x):
        r = sp.fill_buffer(x[done:])
        assert r > 0
        entry = [r, sp.is_input_buffer_complete()]
        if split and r > 3:
            a = r // 3
            outs.append(sp.write_processed(a))
            entry.append(sp.pending_writes())
            outs.append(sp.write_processed(r - a))
        else:
            outs.append(sp.write_processed(r))
        entry.append(sp.pending_writes())
        log.append(tuple(entry))
        done += r
    return log, np.concatenate(outs)


@pytest.mark.parametrize("ra", [2, 8, 32])
def test_block_machine_call_for_call_with_run_ahead(oracle, tmp_path, depth, ra):
    """Every FillBuffer / WriteProcessed / pending_writes / is_input_buffer_complete result equals the restated
    reference's, for a file that ends in a short block, with the output drained in uneven pieces."""
    d, hs = make_santalucia_shaped_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    depth(ra)
    a = H.SoundProcessor.create(conf, 44100, 2)
    assert a.run_ahead() == ra
    b = oracle.SoundProcessor.create(conf, 44100, 2)
    x = seeded_input(3, 11 * 8192 + 3000, 2)
    la, ya = _call_log(a, x, split=True)
    lb, yb = _call_log(b, x, split=True)
    assert la == lb
    assert a.pending_writes() == b.pending_writes() == 8192 - 3000
    assert oracle.rms(ya - yb) <= TOL
    assert oracle.rms(ya - oracle.linear_convolution_f64(x, hs, 2)) <= TOL
    assert a.max_output_value() == pytest.approx(max(0.0, float(ya.max())), abs=1e-6)
    assert a.max_abs_output_value() == pytest.approx(float(np.abs(ya).max()), abs=1e-6)
    # the stream made few engine calls: the blocks really travelled in chunks
    assert fa.lib().fe_stream_max_blocks(C.c_void_p(H._L().fh_processor_stream(a.h))) == ra
    a.reset(); b.reset()
    assert a.pending_writes() == 0 and a.max_output_value() == 0.0
    assert oracle.rms(a.run(x) - b.run(x)) <= TOL


def test_run_ahead_on_equals_off_within_rounding(oracle, tmp_path, depth):
    d, hs = make_santalucia_shaped_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    x = seeded_input(17, 70 * 8192 + 4321, 2)                       # two full 32-block chunks past the ramp, and a tail
    depth(1)
    off = H.SoundProcessor.create(conf, 44100, 2)
    y_off = off.run(x)
    depth(32)
    on = H.SoundProcessor.create(conf, 44100, 2)
    before = H.batching_stats()
    y_on = on.run(x)
    after = H.batching_stats()
    assert y_on.shape == y_off.shape
    assert oracle.rms(y_on - y_off) <= ROUNDING
    assert oracle.rms(y_on - oracle.linear_convolution_f64(x, hs, 2)) <= TOL
    assert on.max_output_value() == pytest.approx(off.max_output_value(), abs=1e-5)
    # 70 whole blocks in chunks of 1, 2, 4, 8, 16, 32, 7 and the short block by itself
    assert after["blocks"] - before["blocks"] == 71
    assert after["requests"] - before["requests"] == 8
    # the same call pattern again gives the same bits (the forms depend on the shape of the calls only)
    on.reset()
    assert np.array_equal(on.run(x), y_on)


def test_gapless_handover_with_run_ahead(oracle, tmp_path, depth):
    """convolve-file-handler.cc:328-351,370-424: file A's short last block never runs ahead — it waits in the block
    buffer, the next file tops it up — so the hand-over finds the ring empty and the join is seamless."""
    d, hs = make_santalucia_shaped_dir(tmp_path)
    for ra in (4, 32):
        depth(ra)
        pool = H.ProcessorPool(3)
        a_sig = seeded_input(31, 9 * 8192 + 1234, 2)
        b_sig = seeded_input(32, 12 * 8192 + 99, 2)
        ya, yb = _drive_gapless(pool, d, a_sig, b_sig)
        assert len(ya) == len(a_sig) and len(yb) == len(b_sig)
        ref = oracle.linear_convolution_f64(np.concatenate([a_sig, b_sig]), hs, 2)
        assert oracle.rms(np.concatenate([ya, yb]) - ref) <= TOL
        pool.close()


def test_pool_return_with_a_half_consumed_ring(oracle, tmp_path, depth):
    """A file closed early: the processor goes back to the pool with blocks read ahead and computed but never
    handed out.  Return() resets it (processor-pool.cc:108): the next file starts from silence, from ITS data."""
    d = make_echo_filter_dir(tmp_path)
    depth(16)
    pool = H.ProcessorPool(3)
    p, _ = pool.get_or_create(d, 44100, 2, 16)
    handle = p.h
    x = seeded_input(5, 40 * 8192, 2)
    done = 0
    for _ in range(5):                                              # 5 of the 40 blocks, then the reader goes away
        r = p.fill_buffer(x[done:])
        p.write_processed(r)
        done += r
    assert done == 5 * 8192
    pool.give_back(p)
    q, _ = pool.get_or_create(d, 44100, 2, 16)
    assert q.h == handle and q.pending_writes() == 0 and q.max_output_value() == 0.0
    z = seeded_input(6, 30000, 2)
    y = q.run(z)
    exp = 0.7 * z.astype(np.float64)
    exp[22050:] += 0.3 * z[:-22050]
    assert oracle.rms(y - exp) <= 1e-6
    # ... also when the reader leaves in the middle of a block (pending_writes > 0)
    pool.give_back(q)
    q, _ = pool.get_or_create(d, 44100, 2, 16)
    for _ in range(3):
        r = q.fill_buffer(x)
        q.write_processed(r)
    r = q.fill_buffer(x)
    q.write_processed(r // 2)
    assert r == 8192 and q.pending_writes() == r - r // 2
    pool.give_back(q)
    q2, _ = pool.get_or_create(d, 44100, 2, 16)
    assert q2.h == handle and q2.pending_writes() == 0
    assert oracle.rms(q2.run(z) - exp) <= 1e-6
    pool.close()


def test_many_files_run_ahead_through_two_lanes(oracle, tmp_path, depth):
    """12 file threads with run-ahead: their chunks meet in the combiner, batches overlap on the engine's two
    launch lanes, every file still gets its own convolution."""
    d, hs = make_santalucia_shaped_dir(tmp_path)
    conf = os.path.join(d, "filter-44100.conf")
    depth(8)
    sigs = [seeded_input(70 + i, 37 * 8192 + 64 * i, 2) for i in range(12)]
    procs = [H.SoundProcessor.create(conf, 44100, 2) for _ in sigs]
    outs = [None] * len(sigs)
    before = H.batching_stats()
    th = [threading.Thread(target=lambda i=i: outs.__setitem__(i, procs[i].run(sigs[i]))) for i in range(len(sigs))]
    [t.start() for t in th]
    [t.join() for t in th]
    after = H.batching_stats()
    for x, y in zip(sigs, outs):
        assert oracle.rms(y - oracle.linear_convolution_f64(x, hs, 2)) <= TOL
    assert after["blocks"] - before["blocks"] == sum((len(x) + 8191) // 8192 for x in sigs)
    assert after["batches"] - before["batches"] < after["requests"] - before["requests"]   # requests shared launches
    assert all(H._L().fh_processor_ok(p.h) for p in procs)


def test_run_ahead_with_unequal_channel_counts_and_small_blocks(oracle, tmp_path, depth):
    """ninp != nout (the ring keeps input and output apart) and a filter short enough for P = 2048."""
    conf = os.path.join(str(tmp_path), "filter-44100.conf")
    with open(conf, "w") as f:
        f.write("/convolver/new 2 3 256 1500\n/impulse/dirac 1 1 0.5 0\n/impulse/dirac 2 2 0.25 700\n"
                "/impulse/dirac 1 3 1.0 1499\n/impulse/dirac 2 3 -0.5 3\n")
    depth(8)
    sp = H.SoundProcessor.create(conf, 44100, 2)
    assert (sp.ninp, sp.nout, sp.fragm) == (2, 3, 2048)
    x = seeded_input(12, 29 * 2048 + 17, 2)
    y = sp.run(x)
    exp = np.zeros((len(x), 3))
    exp[:, 0] = 0.5 * x[:, 0]
    exp[700:, 1] = 0.25 * x[:-700, 1]
    exp[1499:, 2] += x[:-1499, 0]
    exp[3:, 2] += -0.5 * x[:-3, 1]
    assert oracle.rms(y - exp) <= 1e-6


def test_many_channels_run_ahead_by_fewer_blocks(oracle, tmp_path, depth):
    """A processor pins at most 64 MB for its two chunks: a block of a 32-channel stream at P = 8192 is 1 MB of input and 1 MB
    of output (a chunk's output has its own half since round 5: the input stays, for a move to another GPU), so the depth asked
    for (64) becomes 16; the block machine's results do not know (dirac paths: closed form)."""
    conf = os.path.join(str(tmp_path), "filter-44100.conf")
    with open(conf, "w") as f:
        f.write("/convolver/new 32 32 256 20000\n")
        for c in range(32):
            f.write("/impulse/dirac %d %d %.3f %d\n" % (c + 1, c + 1, 0.5 + 0.01 * c, 100 * c))
    depth(64)
    sp = H.SoundProcessor.create(conf, 44100, 32)
    assert (sp.ninp, sp.nout, sp.fragm) == (32, 32, 8192)
    assert sp.run_ahead() == 16
    x = seeded_input(5, 37 * 8192 + 1234, 32)
    y = sp.run(x)
    exp = np.zeros_like(x, dtype=np.float64)
    for c in range(32):
        d = 100 * c
        exp[d:, c] = (0.5 + 0.01 * c) * x[:len(x) - d, c]
    assert oracle.rms(y - exp) <= 1e-6
    depth(64)
    stereo = H.SoundProcessor.create(os.path.join(make_echo_filter_dir(tmp_path), "filter-44100.conf"), 44100, 2)
    assert stereo.run_ahead() == 64


def test_automatic_depth_follows_the_block_size(oracle, tmp_path, depth):
    """Automatic run-ahead: 64 blocks of 8192 frames, the same number of frames for shorter blocks (at most 1024 blocks);
    an explicit depth is taken as given.  The results do not depend on it (dirac paths: closed form)."""
    depth(H.AUTO_RUN_AHEAD)
    conf = os.path.join(str(tmp_path), "filter-44100.conf")
    got = {}
    for size, fragm, want in ((1500, 2048, 256), (100, 128, 1024), (20000, 8192, 64)):
        with open(conf, "w") as f:
            f.write("/convolver/new 2 2 256 %d\n/impulse/dirac 1 1 0.5 0\n/impulse/dirac 2 2 0.25 %d\n" % (size, size - 1))
        os.utime(conf, (1000 + size, 1000 + size))                          # (another configuration for the filter cache)
        sp = H.SoundProcessor.create(conf, 44100, 2)
        assert sp.fragm == fragm and sp.run_ahead() == want, (size, sp.fragm, sp.run_ahead())
        x = seeded_input(size, 300 * fragm + 17, 2)
        y = sp.run(x)
        exp = np.zeros_like(x, dtype=np.float64)
        exp[:, 0] = 0.5 * x[:, 0]
        exp[size - 1:, 1] = 0.25 * x[:len(x) - (size - 1), 1]
        assert oracle.rms(y - exp) <= 1e-6, size
        got[size] = y
    depth(8)
    sp = H.SoundProcessor.create(conf, 44100, 2)
    assert sp.run_ahead() == 8


def test_lanes_on_and_off_give_the_same_bits(oracle, tmp_path, depth):
    """FE_TUNE_LANES = 1 puts every submitted batch on the engine's own stream: the arithmetic does not know."""
    d, hs = make_pass_filter_dir(tmp_path, "lowpass"), None
    conf = os.path.join(d, "filter-44100.conf")
    depth(4)
    sigs = [seeded_input(90 + i, 21 * 8192 + 5 * i, 2) for i in range(4)]

    def run_all(lanes):
        procs = [H.SoundProcessor.create(conf, 44100, 2) for _ in sigs]
        eng = H._L().fh_processor_engine(procs[0].h)
        assert fa.lib().fe_engine_set_tuning(C.c_void_p(eng), 5, lanes) == 0
        try:
            return [p.run(x) for p, x in zip(procs, sigs)]            # one after the other: the same call shapes both times
        finally:
            fa.lib().fe_engine_set_tuning(C.c_void_p(eng), 5, 0)

    one, two = run_all(1), run_all(2)
    for a, b in zip(one, two):
        assert np.array_equal(a, b)
