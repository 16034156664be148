"""GPU: every kernel form the engine can choose, reached with small batches by pinning it
(fe_engine_set_tuning), plus the benchmarked shape itself.

The automatic choice depends on the batch shape — walker run lengths grow with the batch, the MAC
form with the blocks per call — so without pinning a small test batch only ever sees run length 1
and the general kernels.  Reference semantics under test: one block through the convolver,
/root/reference/sound-processor.cc:98-127, evaluated many blocks and streams at a time.
"""
import numpy as np
import pytest

import folve_amd as fa
from helpers import dense_taps, make_pair

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


def _walk_name(got):
    """The launched K2 kernel's name as the pinned build spells it.  The NO_PIN fallback build (-DFOLVE_WALK_NO_PIN: loads and
    stores left to the compiler) has `false` for PIN and no streaming form — the non-temporal hint rides on pinned asm."""
    return got.replace("false", "true")


def _is_fallback_build(got):
    return ", false," in got


@pytest.fixture
def tuned(engine):
    """The session engine with every knob back on automatic afterwards."""
    yield engine
    engine.set_tuning(fwd_run=0, inv_run=0, mac_form=0, fft_form=0, fail_next=0, walk_lpb=0, walk_tiles=0, split=0, duplex_cap_mb=0, walk_fma=0, walk_nt=0)


def test_xlane_exchange_semantics(engine):
    """The cross-lane exchange the FFT rows rely on (fft_core.hpp xlane_*): v_permlane32_swap swaps
    lanes 32..63 of its first operand with lanes 0..31 of its second, v_permlane16_swap the odd
    16-lane rows of the first with the even rows of the second; together a 4 x 4 transpose between
    lane row and register index."""
    out = engine.xlane_selftest()
    lane = np.arange(64)
    a32, b32, a16, b16 = out[0:64], out[64:128], out[128:192], out[192:256]
    assert np.array_equal(a32, np.where(lane < 32, lane, 100 + lane - 32))
    assert np.array_equal(b32, np.where(lane < 32, lane + 32, 100 + lane))
    odd = (lane // 16) % 2 == 1
    assert np.array_equal(a16, np.where(odd, 100 + lane - 16, lane))
    assert np.array_equal(b16, np.where(odd, 100 + lane, lane + 16))
    for i in range(4):                       # lane row a, register i <- lane row i, register a
        assert np.array_equal(out[256 + 64 * i:320 + 64 * i], 10 * (lane // 16) + i)


@pytest.mark.parametrize("runlen", [2, 4, 8, 16, 32])
@pytest.mark.parametrize("channels", [2, 1])
def test_walkers_at_every_run_length(tuned, oracle, runlen, channels):
    """forward_walker<13> / inverse_walker<13, 1|2> with multi-block walks: the block-to-block carry of
    the prefetched PCM quads and Y rows, the peeled short last block after a walk, and a walk that ends
    inside a stream (blocks not a multiple of the run length)."""
    size = 20000                                                       # P = 8192, K = 3
    rng = np.random.default_rng(1000 * runlen + channels)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(channels)}
    if channels == 2:
        paths[(1, 0)] = [(11, (rng.standard_normal(5000) * 0.01).astype(np.float32))]
    sp, flt, _ = make_pair(tuned, oracle, channels, channels, size, paths)
    P = flt.block_size
    nblocks = 2 * runlen + 3
    lens = [nblocks * P, nblocks * P - 4321, (nblocks - 1) * P - 7, (runlen + 1) * P]
    xs = [rng.uniform(-1, 1, (n, channels)).astype(np.float32) for n in lens]
    hd = dense_taps(paths, size)

    tuned.set_tuning(fft_form=2, fwd_run=runlen, inv_run=runlen)
    walk = [flt.open_stream(nblocks) for _ in lens]
    ys = fa.batch_process(walk, xs)
    more = [rng.uniform(-1, 1, (2 * P + 9, channels)).astype(np.float32) for _ in lens]
    ys2 = fa.batch_process(walk, more)                                 # state carried across walker calls
    tuned.set_tuning(fft_form=1)
    gen = [flt.open_stream(nblocks) for _ in lens]
    yg = fa.batch_process(gen, xs)                                     # the general kernels on the same input
    for s in range(len(lens)):
        assert _rms(ys[s] - yg[s]) <= 2e-6, s
        y64 = oracle.linear_convolution_f64(xs[s], hd, channels)
        assert _rms(ys[s] - y64) <= TOL and _rms(ys[s] - y64) / _rms(y64) <= TOL, s
    for s in (0, 1):
        sp.reset()
        assert _rms(ys[s] - sp.run(xs[s])) <= TOL
        pad = (-lens[s]) % P
        full = oracle.linear_convolution_f64(np.concatenate([np.pad(xs[s], ((0, pad), (0, 0))), more[s]]), hd, channels)
        assert _rms(ys2[s] - full[lens[s] + pad:]) <= TOL
        both = np.concatenate([ys[s], ys2[s]])
        pk = walk[s].peaks()
        assert abs(pk[0] - max(0.0, float(both.max()))) <= 1e-6 and abs(pk[1] - float(np.abs(both).max())) <= 1e-6


@pytest.mark.parametrize("size,cross", [
    (262144, False),      # K = 32: 33 rows of G, mac_walk<33>
    (131072, False),      # K = 16: mac_walk<17>
    (65536, False),       # K = 8:  mac_walk<9>
    (50000, True),        # two paths into one output: the walk with a lane set per path (mac_walk<17, 2 lanes, 2 paths>)
    (204800, False),      # K = 25 with a sparse path (echo-like): rows of zeros
])
def test_mac_forms_agree(tuned, oracle, size, cross):
    """K2 in all its forms on the same call: general (1), sliding windows of 4 / 8 / 16 outputs, and
    the whole-call walk (100).  29 blocks: not a multiple of any tile, shorter than the walk's window."""
    rng = np.random.default_rng(size)
    if size == 204800:
        paths = {(0, 0): [(0, np.float32([0.7])), (22050, np.float32([0.3]))],
                 (1, 1): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))]}
    else:
        paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    if cross:
        paths[(0, 1)] = [(3, (rng.standard_normal(size // 2) * 0.01).astype(np.float32))]
    sp, flt, _ = make_pair(tuned, oracle, 2, 2, size, paths)
    P, T, S = flt.block_size, 29, 5
    lens = [T * P - 1000 * s for s in range(S)]
    xs = [rng.uniform(-1, 1, (n, 2)).astype(np.float32) for n in lens]
    more = [rng.uniform(-1, 1, (45 * P - 3, 2)).astype(np.float32) for _ in range(S)]     # longer than the window ring
    hd = dense_taps(paths, size)
    results = {}
    for form in (1, 4, 8, 16, 100):
        tuned.set_tuning(mac_form=form)
        st = [flt.open_stream(48) for _ in range(S)]
        y1 = fa.batch_process(st, xs)
        y2 = fa.batch_process(st, more)                               # history rows come from the first call
        results[form] = (y1, y2)
    for form in (4, 8, 16, 100):
        for s in range(S):
            assert _rms(results[form][0][s] - results[1][0][s]) <= 2e-6, (form, s)
            assert _rms(results[form][1][s] - results[1][1][s]) <= 2e-6, (form, s)
    for s in (0, S - 1):
        y64 = oracle.linear_convolution_f64(xs[s], hd, 2)
        pad = (-lens[s]) % P
        full = oracle.linear_convolution_f64(np.concatenate([np.pad(xs[s], ((0, pad), (0, 0))), more[s]]), hd, 2)
        for form in (1, 16, 100):
            y1, y2 = results[form]
            assert _rms(y1[s] - y64) <= TOL and _rms(y1[s] - y64) / _rms(y64) <= TOL, (form, s)
            assert _rms(y2[s] - full[lens[s] + pad:]) <= TOL, (form, s)


def test_one_block_call_forms_agree(tuned, oracle):
    """The drop-in call (one synchronous block on a page-locked buffer bound to the stream, SoundProcessor::Process,
    /root/reference/sound-processor.cc:98-127) through its three K1/K3 forms: the 1024-thread pair kernels it uses
    by itself, the walkers (fft_form = 2) and the general kernels (fft_form = 1) — whole blocks, then a short last
    block, then a reset and a replay."""
    import ctypes
    L = fa.lib()
    size, C = 100000, 2
    rng = np.random.default_rng(77)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
    sp, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
    P = flt.block_size
    assert P == 8192
    valid = [P, P, P, P, P - 1234]
    x = rng.uniform(-1, 1, (sum(valid), C)).astype(np.float32)
    buf = ctypes.c_void_p()
    assert L.fe_host_alloc(P * C * 4, ctypes.byref(buf)) == 0
    arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), shape=(P, C))
    outs, peaks = {}, {}
    try:
        for form in (0, 2, 1):
            tuned.set_tuning(fft_form=form)
            st = flt.open_stream(1)
            assert L.fe_stream_bind_host_buffer(st.h, buf, P * C * 4) == 0
            for rep in range(2):
                got, pos = [], 0
                for v in valid:
                    arr[:] = 0
                    arr[:v] = x[pos:pos + v]
                    ps, pa = ctypes.c_float(), ctypes.c_float()
                    assert L.fe_stream_process(st.h, buf, v, buf, ctypes.byref(ps), ctypes.byref(pa)) == 0
                    got.append(arr[:v].copy())
                    pos += v
                y = np.concatenate(got)
                if rep == 0:
                    outs[form], peaks[form] = y, (ps.value, pa.value)
                    st.reset()
                else:
                    assert np.array_equal(y, outs[form]), form      # replay after reset: the same bits
            st.close()
    finally:
        L.fe_host_free(buf)
    y64 = oracle.linear_convolution_f64(x, dense_taps(paths, size), C)
    for form in (0, 2, 1):
        assert _rms(outs[form] - y64) <= TOL and _rms(outs[form] - y64) / _rms(y64) <= TOL, form
        assert _rms(outs[form] - outs[1]) <= 2e-6, form
        assert abs(peaks[form][0] - max(0.0, float(outs[form].max()))) <= 1e-6, form
        assert abs(peaks[form][1] - float(np.abs(outs[form]).max())) <= 1e-6, form


def test_hbm_rate_hook(engine):
    """bench.py's measurement hook: three finite, ordered-of-magnitude rates (HBM3E: TB/s, not GB/s)."""
    r = engine.hbm_rates(256 << 20, 5)
    assert set(r) == {"read", "write", "copy"}
    assert all(500.0 < v < 20000.0 for v in r.values()), r
    r2 = engine.hbm_rates2(256 << 20, 5)
    assert set(r2) == {"read", "write", "copy", "write_regions", "copy_regions"}
    assert all(500.0 < v < 20000.0 for v in r2.values()), r2
    r3 = engine.hbm_rates3(256 << 20, 5)
    assert set(r3) == set(r2) | {"copy_best"} and 500.0 < r3["copy_best"] < 20000.0, r3
    with pytest.raises(Exception):
        engine.hbm_rates(1024, 1)                          # below the 1 MiB floor: FE_ERR_PARAM


def test_kernel_times_bound_to_the_dispatches():
    """Profiling mode 2 (what bench.py's per-kernel `ms` and `frac` come from): a start / stop event bound to each dispatch.
    The three kernel times of a call are positive, no longer than the event-to-event times of mode 1 (which hold a launch
    boundary each), their sum fits inside the call's wall time; results are the same bits with profiling on or off; the
    accumulators reset."""
    import time
    import torch
    from folve_amd.capi import BatchPlan, FE_ASYNC, FE_DEVICE_PTRS
    eng = fa.Engine(0)
    size, S, T, C = 65536, 8, 32, 2
    flt = fa.Filter(eng, C, C, size)
    rng = np.random.default_rng(5)
    for c in range(C):
        flt.add(c, c, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))
    flt.commit()
    P = flt.block_size
    streams = [flt.open_stream(T) for _ in range(S)]
    xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
    ys = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    plan = BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], [T * P] * S, FE_DEVICE_PTRS | FE_ASYNC)
    plan.run(); eng.synchronize()
    plain = [y.cpu().numpy().copy() for y in ys]
    n = 40
    for _ in range(5):
        plan.run()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan.run()
    eng.synchronize()
    wall_ms = (time.perf_counter() - t0) / n * 1e3
    eng.reset_profile()
    eng.set_profiling(2)
    for st in streams:
        st.reset()
    plan.run(); eng.synchronize()
    for y, ref in zip(ys, plain):
        assert np.array_equal(y.cpu().numpy(), ref)                 # the same kernels, the same bits
    for _ in range(n - 1):                                           # more rounds than the ring of 32 event sets holds
        plan.run()
    k = eng.get_kernel_profile()
    eng.set_profiling(1)
    for _ in range(n):
        plan.run()
    eng.synchronize()
    e = eng.get_profile()
    eng.set_profiling(0)
    kms = {r: v["ms"] / v["launches"] for r, v in k.items()}
    ems = {r: v["ms"] / v["launches"] for r, v in e.items()}
    assert all(v["launches"] == n for v in k.values()) and all(v["launches"] == n for v in e.values())
    for r in kms:
        assert 0.002 < kms[r] < 5.0, kms
        assert kms[r] <= ems[r] * 1.05 + 0.001, (kms, ems)           # an event behind the kernel also holds the boundary
    assert sum(kms.values()) <= wall_ms * 1.10 + 0.005, (kms, wall_ms)
    eng.reset_profile()
    assert all(v["launches"] == 0 and v["ms"] == 0.0 for v in eng.get_kernel_profile().values())
    for s_ in streams:
        s_.close()


def test_benchmarked_shape_parity(engine, oracle):
    """bench.py's workload, kernel for kernel: 64 streams x 2 channels x 256 blocks per call through a
    262 144-tap 2-path filter, streams opened for 256-block calls, device-resident PCM, automatic form
    choice (walker run length 32, the whole-call MAC walk).  Some ragged tails; a second call carries
    the state.  Checked streams: against the float64 convolution (absolute and relative) and against
    the oracle; all streams against an independent float64 FFT convolution."""
    torch = pytest.importorskip("torch")
    S, T, C, size = 64, 256, 2, 262144
    rng = np.random.default_rng(3)
    paths = {}
    for c in range(C):
        h = rng.standard_normal(size).astype(np.float32)
        paths[(c, c)] = [(0, h / np.linalg.norm(h))]
    sp, flt, _ = make_pair(engine, oracle, C, C, size, paths)
    P = flt.block_size
    lens = [T * P - (0 if s % 5 else 777 + 8 * s) for s in range(S)]
    check = (0, 5, 63)                                              # 0 and 5 are ragged, 63 whole
    xs = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
    ys = [torch.zeros(T * P, C, device="cuda") for _ in range(S)]
    streams = [flt.open_stream(T) for _ in range(S)]
    from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS
    torch.cuda.synchronize()                 # (the engine has its own HIP stream: torch's fills come first)
    BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys], lens, FE_DEVICE_PTRS).run()
    k = engine.last_kernels()
    # the benchmarked launch's own kernels: the stereo walkers and the walk's streaming form (2.1 GB of Y: non-temporal rows)
    assert k["forward"].startswith("forward_walker_kernel<13") and k["inverse"].startswith("inverse_walker_kernel<13"), k
    assert _walk_name(k["mac"]) == ("mac_walk3_kernel<33, 7, true, 1, 1>" if _is_fallback_build(k["mac"]) else "mac_walk3_nt_kernel<33, 7, true, 1, 1>"), k
    x2 = [torch.rand(T * P, C, device="cuda") * 2 - 1 for _ in range(S)]
    y2 = [torch.zeros(T * P, C, device="cuda") for _ in range(S)]
    torch.cuda.synchronize()
    BatchPlan(streams, [x.data_ptr() for x in x2], [y.data_ptr() for y in y2], [T * P] * S, FE_DEVICE_PTRS).run()
    hd = dense_taps(paths, size)
    for s in check:
        x = xs[s].cpu().numpy()[:lens[s]]
        y = ys[s].cpu().numpy()[:lens[s]]
        y64 = oracle.linear_convolution_f64(x, hd, C)
        assert _rms(y - y64) <= TOL and _rms(y - y64) / _rms(y64) <= TOL, s
        sp.reset()
        yo = sp.run(x)
        assert _rms(y - yo) <= TOL and _rms(y - yo) / _rms(yo) <= TOL, s
        pad = (-lens[s]) % P
        xx = np.concatenate([np.pad(x, ((0, pad), (0, 0))), x2[s].cpu().numpy()])
        full = oracle.linear_convolution_f64(xx, hd, C)
        second = y2[s].cpu().numpy()
        assert _rms(second - full[lens[s] + pad:]) <= TOL, s
    # blocks past a ragged tail were never written
    assert float(ys[0][lens[0]:].abs().max()) == 0.0
    # EVERY stream of the batch, both calls, against an independent float64 FFT convolution (torch.fft on
    # the GPU: test-only cross-check, never on the product path)
    n = 2 * T * P + size
    Hf = [torch.fft.rfft(torch.from_numpy(hd[(c, c)].astype(np.float64)).cuda(), n) for c in range(C)]
    worst = 0.0
    for s in range(S):
        x = torch.zeros(2 * T * P, C, dtype=torch.float64, device="cuda")
        x[:lens[s]] = xs[s][:lens[s]].double()
        x[T * P:] = x2[s].double()                       # time advanced by whole blocks: the second call starts at T*P
        ref = torch.stack([torch.fft.irfft(torch.fft.rfft(x[:, c], n) * Hf[c], n)[:2 * T * P] for c in range(C)], 1)
        got = torch.cat([ys[s][:lens[s]].double(), y2[s].double()])
        want = torch.cat([ref[:lens[s]], ref[T * P:]])
        err = float(torch.sqrt(torch.mean((got - want) ** 2)))
        worst = max(worst, err / float(torch.sqrt(torch.mean(want ** 2))))
    assert worst <= TOL, worst


def test_injected_failure_leaves_stream_state(tuned, oracle):
    """A launch round that fails before its kernels are enqueued (fault injection) reports
    FE_ERR_DEVICE and leaves the stream where it was: the same block then processes as if the failed
    call had never happened."""
    rng = np.random.default_rng(5)
    size = 20000
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    _, flt, st = make_pair(tuned, oracle, 2, 2, size, paths, max_blocks=4)
    P = flt.block_size
    x = rng.uniform(-1, 1, (3 * P, 2)).astype(np.float32)
    y_a = st.process_blocks(x[:P])
    done = st.blocks_done()
    tuned.set_tuning(fail_next=1)
    with pytest.raises(fa.FolveError) as err:
        st.process_blocks(x[P:2 * P])
    assert err.value.code == -4 and "injected" in str(err.value)
    assert st.blocks_done() == done
    y_b = st.process_blocks(x[P:])
    ref = flt.open_stream(4).process_blocks(x)
    assert np.array_equal(np.concatenate([y_a, y_b]), ref)


def test_failure_in_a_later_group_of_a_batch(tuned, oracle):
    """A batch over streams of two filters is launched group by group.  When the SECOND group's round is refused
    (fault injection: the 2nd launch round from now), the call fails, the first group's streams have consumed
    their block and the second's have not — which is what a caller (the per-GPU combiner) tells apart with
    fe_stream_blocks_done before it retries anything."""
    rng = np.random.default_rng(8)
    flts = []
    for size in (20000, 30000):
        paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
        flts.append(make_pair(tuned, oracle, 2, 2, size, paths)[1])
    P = flts[0].block_size
    sa, sb = flts[0].open_stream(2), flts[1].open_stream(2)
    x = rng.uniform(-1, 1, (2 * P, 2)).astype(np.float32)
    fa.batch_process([sa, sb], [x[:P], x[:P]])
    da, db = sa.blocks_done(), sb.blocks_done()
    tuned.set_tuning(fail_next=2)
    with pytest.raises(fa.FolveError) as err:
        fa.batch_process([sa, sb], [x[P:], x[P:]])
    assert err.value.code == -4
    moved = (sa.blocks_done() - da, sb.blocks_done() - db)
    assert sorted(moved) == [0, 1], moved                             # exactly one group ran
    # the group that did not run repeats its block as if nothing had happened
    late = sb if moved[1] == 0 else sa
    ref = (flts[1] if late is sb else flts[0]).open_stream(2).process_blocks(x)
    assert np.array_equal(late.process_blocks(x[P:]), ref[P:])


def test_submit_wait_split_matches_the_synchronous_call(tuned, oracle):
    """fe_batch_submit / fe_ticket_wait (what the per-GPU combiner keeps the GPU busy with): two batches of
    one-block calls on bound page-locked buffers submitted back to back, waited for afterwards — the same bits
    as the synchronous calls; buffers that are not bound are refused with FE_ERR_UNSUPPORTED and nothing runs."""
    import ctypes
    L = fa.lib()
    size, C, S = 60000, 2, 6
    rng = np.random.default_rng(99)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
    sp, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
    P = flt.block_size
    nblk = 4
    xs = [rng.uniform(-1, 1, (nblk * P, C)).astype(np.float32) for _ in range(S)]
    bufs, arrs = [], []
    for _ in range(S):
        b = ctypes.c_void_p()
        assert L.fe_host_alloc(P * C * 4, ctypes.byref(b)) == 0
        bufs.append(b)
        arrs.append(np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_float)), shape=(P, C)))

    def arrays(streams, idx):
        n = len(idx)
        ss = (ctypes.c_void_p * n)(*[streams[i].h for i in idx])
        pp = (ctypes.c_void_p * n)(*[bufs[i].value for i in idx])
        nn = (ctypes.c_longlong * n)(*([P] * n))
        return ss, pp, nn

    try:
        # synchronous reference: one fe_batch_process per block over all streams
        ref_streams = [flt.open_stream(1) for _ in range(S)]
        for s, b in zip(ref_streams, bufs):
            assert L.fe_stream_bind_host_buffer(s.h, b, P * C * 4) == 0
        ref = [[] for _ in range(S)]
        for k in range(nblk):
            for i in range(S):
                arrs[i][:] = xs[i][k * P:(k + 1) * P]
            ss, pp, nn = arrays(ref_streams, list(range(S)))
            assert L.fe_batch_process(ss, S, pp, nn, pp, 0) == 0
            for i in range(S):
                ref[i].append(arrs[i].copy())
        # two halves submitted back to back, then waited for
        streams = [flt.open_stream(1) for _ in range(S)]
        for s, b in zip(streams, bufs):
            assert L.fe_stream_bind_host_buffer(s.h, b, P * C * 4) == 0
        halves = [list(range(0, S // 2)), list(range(S // 2, S))]
        for k in range(nblk):
            for i in range(S):
                arrs[i][:] = xs[i][k * P:(k + 1) * P]
            tickets = []
            for idx in halves:
                ss, pp, nn = arrays(streams, idx)
                t = ctypes.c_void_p()
                assert L.fe_batch_submit(ss, len(idx), pp, nn, pp, ctypes.byref(t)) == 0 and t.value
                tickets.append(t)
            for t in tickets:
                assert L.fe_ticket_wait(t) == 0
            for i in range(S):
                assert np.array_equal(arrs[i], ref[i][k]), (k, i)
        y64 = oracle.linear_convolution_f64(xs[0], dense_taps(paths, size), C)
        assert _rms(np.concatenate(ref[0]) - y64) <= TOL
        # an ordinary numpy buffer is not bound: refused, nothing enqueued, the stream does not advance
        before = streams[0].blocks_done()
        plain = np.zeros((P, C), np.float32)
        ss = (ctypes.c_void_p * 1)(streams[0].h)
        pp = (ctypes.c_void_p * 1)(plain.ctypes.data)
        nn = (ctypes.c_longlong * 1)(P)
        t = ctypes.c_void_p()
        assert L.fe_batch_submit(ss, 1, pp, nn, pp, ctypes.byref(t)) == -6 and not t.value
        assert streams[0].blocks_done() == before
        for s in streams + ref_streams:
            s.close()
    finally:
        for b in bufs:
            L.fe_host_free(b)


@pytest.mark.parametrize("size,channels,lpbs", [
    (204800, 2, (1, 2, 4)),       # K + 1 = 26 rows: mac_walk<33, 1 lane>, <17, 2 lanes>, <9, 4 lanes>
    (524288, 2, (2, 4)),          # K + 1 = 65 rows (cfg4's filter): <33, 2>, <17, 4>
    (1048576, 1, (4,)),           # K + 1 = 129 rows (MAXSIZE, zita-config.h:61): <33, 4>
    (70000, 1, (1, 2, 4)),        # K + 1 = 10 rows: <17, 1>, <17, 2>, <9, 4>
])
def test_mac_walk_lanes_per_bin_and_time_tiles(tuned, oracle, size, channels, lpbs):
    """The whole-call walk with the filter's rows spread over 1 / 2 / 4 lanes of a bin (the window handed down from
    lane to lane by DPP shifts) and with the call cut into time tiles, against the general MAC kernel on the same
    calls; state carried into a second call; ragged lengths; and the float64 convolution."""
    rng = np.random.default_rng(size + channels)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(channels)}
    sp, flt, _ = make_pair(tuned, oracle, channels, channels, size, paths)
    P, K = flt.block_size, flt.partitions
    T = 2 * K + 7 if K <= 32 else K + 9                                  # longer than the filter, not a multiple of anything
    S = 3
    lens = [T * P - 777 * s for s in range(S)]
    xs = [rng.uniform(-1, 1, (n, channels)).astype(np.float32) for n in lens]
    more = [rng.uniform(-1, 1, (37 * P - 5, channels)).astype(np.float32) for _ in range(S)]
    tuned.set_tuning(mac_form=1)
    st = [flt.open_stream(T) for _ in range(S)]
    ref1 = fa.batch_process(st, xs)
    ref2 = fa.batch_process(st, more)
    hd = dense_taps(paths, size)
    y64 = oracle.linear_convolution_f64(xs[0], hd, channels)
    assert _rms(ref1[0] - y64) <= TOL
    for lpb in lpbs:
        for tiles in (1, 2, 5):
            # both arithmetic forms of the walk: four FMAs per complex multiply-add (kernels.hip) and three (mac_walk3.hip:
            # odd and even steps, the carried half, the carry out of the history at the start of every time tile)
            for fma in (4, 3):
                tuned.set_tuning(mac_form=100, walk_lpb=lpb, walk_tiles=tiles, walk_fma=fma)
                st = [flt.open_stream(T) for _ in range(S)]
                y1 = fa.batch_process(st, xs)
                assert tuned.last_kernels()["mac"].startswith("mac_walk3_kernel<" if fma == 3 else "mac_walk_kernel<"), tuned.last_kernels()
                y2 = fa.batch_process(st, more)
                for s in range(S):
                    assert _rms(y1[s] - ref1[s]) <= 2e-6, (lpb, tiles, fma, s)
                    assert _rms(y2[s] - ref2[s]) <= 2e-6, (lpb, tiles, fma, s)
                assert _rms(y1[0] - y64) <= TOL and _rms(y1[0] - y64) / _rms(y64) <= TOL, (lpb, tiles, fma)


@pytest.mark.parametrize("case", ["2x2 K32", "2x2 K64", "4to2 K8", "4to2 K16", "4to2 K32", "ragged 3x3"])
def test_mac_walk_with_several_paths_per_output(tuned, oracle, case):
    """Full filter matrices (a true-stereo reverb's four paths; jconvolver's /impulse/read lines for every (input, output),
    /root/reference/zita-config.cc:55-177): the whole-call walk with one lane set per path of an output — 2 or 4 sets,
    each 1 or 2 lanes wide, an output with fewer paths than sets, one with none — against the general MAC kernel on the
    same calls (state carried into a second call, time tiles) and the float64 convolution."""
    rng = np.random.default_rng(sum(map(ord, case)))
    def taps(n, scale=1.0):
        return (rng.standard_normal(n) * scale / np.sqrt(n)).astype(np.float32)
    if case.startswith("2x2"):
        size = 262144 if case.endswith("K32") else 524288
        cin = cout = 2
        paths = {(i, o): [(0, taps(size, 1.0 if i == o else 0.3))] for i in range(2) for o in range(2)}
        lpbs = (0, 4) if case.endswith("K32") else (0,)               # automatic; K32 also two lanes per path
    elif case.startswith("4to2"):
        size = {"K8": 65536, "K16": 131072, "K32": 262144}[case.split()[1]]
        cin, cout = 4, 2
        paths = {(i, 0): [(0, taps(size, 0.5))] for i in range(4)}
        paths.update({(i, 1): [(i * 100, taps(size - 1000, 0.5))] for i in (0, 2, 3)})      # three paths: one set stays empty
        lpbs = (0,)
    else:
        size, cin, cout = 100000, 3, 3
        paths = {(0, 0): [(0, taps(size))], (2, 0): [(5, taps(size // 2, 0.2))], (1, 1): [(0, taps(size))]}   # output 2: silence
        lpbs = (0, 4)
    sp, flt, _ = make_pair(tuned, oracle, cin, cout, size, paths)
    P, K = flt.block_size, flt.partitions
    T = K + 9
    S = 2
    lens = [T * P - 777 * s for s in range(S)]
    xs = [rng.uniform(-1, 1, (n, cin)).astype(np.float32) for n in lens]
    more = [rng.uniform(-1, 1, (19 * P - 5, cin)).astype(np.float32) for _ in range(S)]
    tuned.set_tuning(mac_form=1)
    st = [flt.open_stream(T) for _ in range(S)]
    ref1 = fa.batch_process(st, xs)
    ref2 = fa.batch_process(st, more)
    y64 = oracle.linear_convolution_f64(xs[1], dense_taps(paths, size), cout)
    assert _rms(ref1[1] - y64) <= TOL
    for lpb in lpbs:
        for tiles in (1, 3):
            for fma in (4, 3):
                tuned.set_tuning(mac_form=100, walk_lpb=lpb, walk_tiles=tiles, walk_fma=fma)
                st = [flt.open_stream(T) for _ in range(S)]
                y1 = fa.batch_process(st, xs)
                assert tuned.last_kernels()["mac"].startswith("mac_walk3_kernel<" if fma == 3 else "mac_walk_kernel<"), tuned.last_kernels()
                y2 = fa.batch_process(st, more)
                for s_ in range(S):
                    assert _rms(y1[s_] - ref1[s_]) <= 2e-6, (lpb, tiles, fma, s_)
                    assert _rms(y2[s_] - ref2[s_]) <= 2e-6, (lpb, tiles, fma, s_)
                assert _rms(y1[1] - y64) <= TOL and _rms(y1[1] - y64) / _rms(y64) <= TOL, (lpb, tiles, fma)
                if cout == 3:
                    assert not y1[0][:, 2].any()
    sp.reset()
    assert _rms(ref1[1] - sp.run(xs[1])) <= TOL


def test_walk_window_for_a_short_filter_matrix(tuned, oracle):
    """A 2 x 2 matrix of ~2 s reverbs (K + 1 = 12 rows): two path sets of ONE lane each.  One lane per bin has a finer ladder
    of windows (13 / 21 / 26 / 29 rows) that the path-set forms do not: the launcher must take the 17-row window here, not fall
    through to the 33-row one (twice the arithmetic), and a lone stereo stream with the same 12 rows takes the 13-row one."""
    rng = np.random.default_rng(12)
    size = 90000                                                        # 11 partitions, 12 rows of G
    paths = {(i, o): [(0, (rng.standard_normal(size) * (1.0 if i == o else 0.3) / np.sqrt(size)).astype(np.float32))]
             for i in range(2) for o in range(2)}
    _, flt, _ = make_pair(tuned, oracle, 2, 2, size, paths)
    P, T, S = flt.block_size, 40, 2
    xs = [rng.uniform(-1, 1, (T * P - 31 * s, 2)).astype(np.float32) for s in range(S)]
    tuned.set_tuning(mac_form=1)
    ref = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
    for fma, name in ((4, "mac_walk_kernel<17, 15, true, 4, 2, 2>"), (3, "mac_walk3_kernel<17, 15, true, 2, 2>")):
        tuned.set_tuning(mac_form=100, walk_fma=fma, walk_lpb=2)
        ys = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
        assert tuned.last_kernels()["mac"].replace("false", "true") == name, tuned.last_kernels()   # (false: the NO_PIN fallback build)
        for s in range(S):
            assert _rms(ys[s] - ref[s]) <= 2e-6
    diag = {(c, c): paths[(c, c)] for c in range(2)}
    _, flt1, _ = make_pair(tuned, oracle, 2, 2, size, diag)
    tuned.set_tuning(mac_form=100, walk_fma=4, walk_lpb=1)
    fa.batch_process([flt1.open_stream(T) for _ in range(S)], xs)
    assert _walk_name(tuned.last_kernels()["mac"]) == "mac_walk_kernel<13, 7, true, 4, 1, 1>", tuned.last_kernels()
    tuned.set_tuning(mac_form=100, walk_fma=3, walk_lpb=1)
    fa.batch_process([flt1.open_stream(T) for _ in range(S)], xs)
    assert _walk_name(tuned.last_kernels()["mac"]) == "mac_walk3_kernel<13, 7, true, 1, 1>", tuned.last_kernels()


@pytest.mark.parametrize("size,window", [(65536, 9), (98304, 13), (131072, 17), (163840, 21), (204800, 26), (229376, 29), (262144, 33)])
def test_walk_window_ladder_in_both_forms(tuned, oracle, size, window):
    """One lane per bin has a ladder of windows (9 / 13 / 17 / 21 / 26 / 29 / 33 rows of G): every rung, in the four-FMA and
    in the three-FMA form (whose prefetch depth differs at 26 rows: the ring must have an even number of slots), launches
    the instantiation it should and agrees with the general kernel; odd tile lengths, a ragged last block."""
    rng = np.random.default_rng(size)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    _, flt, _ = make_pair(tuned, oracle, 2, 2, size, paths)
    P, K = flt.block_size, flt.partitions
    assert K + 1 <= window
    T, S = 2 * K + 5, 3
    xs = [rng.uniform(-1, 1, (T * P - 501 * s - 1, 2)).astype(np.float32) for s in range(S)]
    tuned.set_tuning(mac_form=1)
    ref = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
    for fma, name in ((4, "mac_walk_kernel<%d, 7, true, 4, 1, 1>" % window),
                      (3, "mac_walk3_kernel<%d, %d, true, 1, 1>" % (window, 8 if window == 26 else 7))):
        for tiles in (1, 3):
            tuned.set_tuning(mac_form=100, walk_lpb=1, walk_tiles=tiles, walk_fma=fma)
            ys = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
            assert tuned.last_kernels()["mac"].replace("false", "true") == name, tuned.last_kernels()   # (false: the NO_PIN fallback build)
            for s in range(S):
                assert _rms(ys[s] - ref[s]) <= 2e-6, (fma, tiles, s)


def test_the_streaming_form_of_the_walk_is_the_same_arithmetic(tuned, oracle):
    """`mac_walk3_nt_kernel<33, 7, true, 1, 1>`: the 33-row walk whose row loads and stores carry the non-temporal hint —
    chosen by itself where a launch's rows of Y exceed 192 MB (cfg3's batch: test_benchmarked_shape_parity), pinned here on a
    small batch (FE_TUNE_WALK_NT = 2) and held against the plain form (= 1): the same instructions but for a cache-policy
    bit, so the same bits; several tile counts, ragged tails."""
    size = 262144
    rng = np.random.default_rng(77)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    _, flt, _ = make_pair(tuned, oracle, 2, 2, size, paths)
    P, K = flt.block_size, flt.partitions
    T, S = K + 9, 3
    xs = [rng.uniform(-1, 1, (T * P - 777 * s - 1, 2)).astype(np.float32) for s in range(S)]
    for tiles in (1, 3):
        outs = {}
        for nt, name in ((1, "mac_walk3_kernel<33, 7, true, 1, 1>"), (2, "mac_walk3_nt_kernel<33, 7, true, 1, 1>")):
            tuned.set_tuning(mac_form=100, walk_lpb=1, walk_tiles=tiles, walk_fma=3, walk_nt=nt)
            outs[nt] = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
            got = tuned.last_kernels()["mac"]
            assert _walk_name(got) == (name.replace("_nt_", "_") if _is_fallback_build(got) else name), tuned.last_kernels()
        for s in range(S):
            assert np.array_equal(outs[1][s], outs[2][s]), (tiles, s)
    for bad in (-1, 3):
        with pytest.raises(fa.FolveError):
            tuned.set_tuning(walk_nt=bad)                  # FE_ERR_PARAM: 0 (by the launch's bytes), 1 (never), 2 (always)
    tuned.set_tuning(mac_form=100, walk_lpb=1, walk_tiles=0, walk_fma=3, walk_nt=0)      # by itself: a batch this small stays plain
    fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
    assert tuned.last_kernels()["mac"].startswith("mac_walk3_kernel<33,")
    # every rung of the one-lane ladder has the streaming form (a big batch through a SHORT filter is memory-bound too)
    size = 65536
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    _, flt9, _ = make_pair(tuned, oracle, 2, 2, size, paths)
    xs9 = [rng.uniform(-1, 1, (21 * P - 5, 2)).astype(np.float32) for _ in range(2)]
    outs = {}
    for nt, name in ((1, "mac_walk3_kernel<9, 7, true, 1, 1>"), (2, "mac_walk3_nt_kernel<9, 7, true, 1, 1>")):
        tuned.set_tuning(mac_form=100, walk_lpb=1, walk_tiles=2, walk_fma=3, walk_nt=nt)
        outs[nt] = fa.batch_process([flt9.open_stream(21) for _ in range(2)], xs9)
        got = tuned.last_kernels()["mac"]
        assert _walk_name(got) == (name.replace("_nt_", "_") if _is_fallback_build(got) else name), tuned.last_kernels()
    assert all(np.array_equal(a, b) for a, b in zip(outs[1], outs[2]))


@pytest.mark.parametrize("size,block", [(256, 256), (300, 512), (700, 1024), (1500, 2048), (3000, 4096), (4097, 8192)])
def test_the_walk_at_short_partitions(tuned, oracle, size, block):
    """Filters of up to 4096 taps get partitions of 256 .. 4096 frames (zita-fconfig.cc:74-77): the walk's row is P * 8 bytes
    then, its buffer offsets and the range-checked end of a tile scale with it.  Both walk forms, several tile counts, odd
    call lengths and a ragged last block against the general kernels."""
    rng = np.random.default_rng(size)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    paths[(0, 1)] = [(3, (rng.standard_normal(size - 3) * 0.05).astype(np.float32))]     # two paths into output 1: the lane sets
    _, flt, _ = make_pair(tuned, oracle, 2, 2, size, paths)
    P = flt.block_size
    assert P == block
    T, S = 45, 3
    xs = [rng.uniform(-1, 1, (T * P - 7 * s - 1, 2)).astype(np.float32) for s in range(S)]
    tuned.set_tuning(mac_form=1)
    ref = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
    for fma in (4, 3):
        for tiles in (1, 2, 5):
            tuned.set_tuning(mac_form=100, walk_tiles=tiles, walk_fma=fma)
            ys = fa.batch_process([flt.open_stream(T) for _ in range(S)], xs)
            assert tuned.last_kernels()["mac"].startswith("mac_walk3_kernel<" if fma == 3 else "mac_walk_kernel<"), tuned.last_kernels()
            for s in range(S):
                assert _rms(ys[s] - ref[s]) <= 2e-6, (fma, tiles, s)


def test_three_fma_walk_at_cfg3_and_cfg4_shapes_against_float64(tuned, oracle):
    """The three-FMA walk's rounding (its three sums have the magnitude |x||g|, the four-FMA form's |Re|, |Im|) where it
    counts: cfg3's filter (K = 32, one lane per bin: 33 rows) and cfg4's (K = 64, two lanes) over calls long enough for the
    sums to fill, against the float64 linear convolution and against the four-FMA walk."""
    rng = np.random.default_rng(34)
    for size, C, lpb in ((262144, 2, 1), (524288, 2, 2)):
        paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
        _, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
        P, K = flt.block_size, flt.partitions
        T = K + 40
        x = rng.uniform(-1, 1, (T * P - 17, C)).astype(np.float32)
        out = {}
        for fma in (4, 3):
            tuned.set_tuning(mac_form=100, walk_lpb=lpb, walk_fma=fma)
            out[fma] = flt.open_stream(T).process_blocks(x)
        y64 = oracle.linear_convolution_f64(x, dense_taps(paths, size), C)
        e3, e4 = _rms(out[3] - y64), _rms(out[4] - y64)
        assert e4 <= 1e-6 and e3 <= 1e-6 and e3 <= 3 * e4 + 1e-7, (size, e3, e4)
        assert _rms(out[3] - out[4]) <= 2e-6


def test_automatic_walk_shapes_for_one_stream_calls(tuned, oracle):
    """What the launchers pick by themselves for cfg2's and cfg4's shapes at a 128-block call (one stream: K2's lanes per
    bin and time tiles so that the launch fills the chip) and for cfg4's at the benchmarked 256 blocks (1 024 (block, pair)
    units: K1 / K3 walk in channel-pair mode) agrees with the general kernels."""
    rng = np.random.default_rng(77)
    for size, C, T in ((204800, 2, 128), (524288, 8, 128), (524288, 8, 256)):
        paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
        _, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
        P = flt.block_size
        x = rng.uniform(-1, 1, (T * P - 99, C)).astype(np.float32)
        tuned.set_tuning(mac_form=0, fft_form=0, walk_lpb=0, walk_tiles=0)
        y_auto = flt.open_stream(T).process_blocks(x)
        tuned.set_tuning(mac_form=1, fft_form=1)
        y_gen = flt.open_stream(T).process_blocks(x)
        assert _rms(y_auto - y_gen) <= 2e-6
        c = C - 1
        y64 = oracle.linear_convolution_f64(x[:40 * P], {(c, c): dense_taps(paths, size)[(c, c)]}, C)[:, c]
        assert _rms(y_auto[:40 * P, c] - y64) <= TOL


def test_lone_stream_measurement_knobs_keep_the_results(tuned, oracle):
    """The two forms round 4 measured for a lone stream's long call and did not adopt stay
    selectable for measurements, so they stay correct: FE_TUNE_SPLIT — the call's blocks as 2 .. 5 time tiles whose
    K1 -> K2 -> K3 chains alternate between the two launch lanes, tile c's K2 behind tile c - 1's K1 by event — and
    fft_form = 4, the one-block pair kernels for a stereo stream's whole call.  Two calls each, the second in the automatic
    form on the state the first left (ring positions, cross-lane ordering): equal to the automatic form within float32
    rounding, and right."""
    rng = np.random.default_rng(404)
    for size, C, T in ((204800, 2, 96), (524288, 8, 64)):
        paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
        _, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
        P = flt.block_size
        x1 = rng.uniform(-1, 1, (T * P - 321, C)).astype(np.float32)        # (a short last block: time advances by whole blocks)
        x2 = rng.uniform(-1, 1, (17 * P, C)).astype(np.float32)
        tuned.set_tuning(split=1, fft_form=0)
        st = flt.open_stream(T)
        ref = [st.process_blocks(x1), st.process_blocks(x2)]
        knobs = [dict(split=2), dict(split=3), dict(split=5)] + ([dict(split=1, fft_form=4)] if C == 2 else [])
        for kn in knobs:
            tuned.set_tuning(split=1, fft_form=0)
            tuned.set_tuning(**kn)
            st = flt.open_stream(T)
            y1 = st.process_blocks(x1)
            tuned.set_tuning(split=1, fft_form=0)
            y2 = st.process_blocks(x2)
            assert _rms(y1 - ref[0]) <= 2e-6 and _rms(y2 - ref[1]) <= 2e-6, (C, kn)
        c = C - 1
        pad = (-len(x1)) % P
        xx = np.concatenate([x1, np.zeros((pad, C), np.float32), x2])
        y64 = oracle.linear_convolution_f64(xx, {(c, c): dense_taps(paths, size)[(c, c)]}, C)[:, c]
        assert _rms(ref[0][:, c] - y64[:len(x1)]) <= TOL and _rms(ref[1][:, c] - y64[len(x1) + pad:]) <= TOL
    tuned.set_tuning(split=0, fft_form=0)


@pytest.mark.parametrize("channels,size", [(4, 20000), (8, 20000), (6, 3000), (64, 128)])
def test_channel_pair_kernels_match_the_per_channel_ones(tuned, oracle, channels, size):
    """Streams of four or more channels: forward_chpair / inverse_chpair (a workgroup per channel pair, 8-byte loads
    and stores of interleaved frames) against the per-channel general kernels — the same butterflies on the same
    numbers, hence the same bits — for P = 8192, 2048 and 128, ragged tails, two calls."""
    rng = np.random.default_rng(channels * 1000 + size)
    paths = {}
    for c in range(channels):
        paths[(c, (c + 1) % channels)] = [(0, (rng.standard_normal(min(size, 4000)) * 0.05).astype(np.float32))]
        if c % 3 == 0:
            paths[(c, c)] = [(size // 2, (rng.standard_normal(size // 2) / np.sqrt(size)).astype(np.float32))]
    sp, flt, _ = make_pair(tuned, oracle, channels, channels, size, paths)
    P = flt.block_size
    lens = [11 * P, 7 * P + 1234 % P, 3 * P - 1]
    xs = [rng.uniform(-1, 1, (n, channels)).astype(np.float32) for n in lens]
    more = [rng.uniform(-1, 1, (2 * P + 3, channels)).astype(np.float32) for _ in lens]
    outs = {}
    for form in (3, 1):                                                 # 3: the pair kernels, never the pair walkers
        tuned.set_tuning(fft_form=form)
        st = [flt.open_stream(11) for _ in lens]
        outs[form] = (fa.batch_process(st, xs), fa.batch_process(st, more), [s_.peaks() for s_ in st])
    for s_ in range(len(lens)):
        assert np.array_equal(outs[3][0][s_], outs[1][0][s_])
        assert np.array_equal(outs[3][1][s_], outs[1][1][s_])
        assert outs[3][2][s_] == outs[1][2][s_]
    sp.reset()
    assert _rms(outs[3][0][1] - sp.run(xs[1])) <= TOL
    y64 = oracle.linear_convolution_f64(xs[0], dense_taps(paths, size), channels)
    assert _rms(outs[3][0][0] - y64) <= TOL


@pytest.mark.parametrize("channels,runlen,nblocks", [(4, 1, 5), (4, 2, 19), (8, 4, 11), (8, 2, 23), (6, 8, 19), (8, 32, 40)])
def test_channel_pair_walkers(tuned, oracle, channels, runlen, nblocks):
    """Streams of four or more channels at P = 8192: forward_walker<13, MC> / inverse_walker<13, 2, MC> — a workgroup
    walks `runlen` consecutive blocks of one channel pair, 8-byte loads and stores at the frame stride — against the
    general kernels, the float64 convolution and the reference restatement: both grid orders (8 or more runs: the
    XCD-local one, with its empty trailing workgroups), walks that end inside a stream, the peeled short last block,
    state carried into a second call, peaks."""
    size = 20000                                                       # P = 8192, K = 3
    rng = np.random.default_rng(channels * 10000 + runlen * 100 + nblocks)
    paths = {}
    for c in range(channels):
        paths[(c, (c + 1) % channels)] = [(0, (rng.standard_normal(4000) * 0.05).astype(np.float32))]
        if c % 3 == 0:
            paths[(c, c)] = [(size // 2, (rng.standard_normal(size // 2) / np.sqrt(size)).astype(np.float32))]
    sp, flt, _ = make_pair(tuned, oracle, channels, channels, size, paths)
    P = flt.block_size
    lens = [nblocks * P, nblocks * P - 4321, (nblocks - 1) * P - 7, (runlen + 1) * P]
    xs = [rng.uniform(-1, 1, (n, channels)).astype(np.float32) for n in lens]
    more = [rng.uniform(-1, 1, (2 * P + 9, channels)).astype(np.float32) for _ in lens]
    hd = dense_taps(paths, size)

    tuned.set_tuning(fft_form=2, fwd_run=runlen, inv_run=runlen)
    walk = [flt.open_stream(nblocks) for _ in lens]
    ys = fa.batch_process(walk, xs)
    ys2 = fa.batch_process(walk, more)
    tuned.set_tuning(fft_form=1)
    gen = [flt.open_stream(nblocks) for _ in lens]
    yg = fa.batch_process(gen, xs)
    yg2 = fa.batch_process(gen, more)
    for s in range(len(lens)):
        assert _rms(ys[s] - yg[s]) <= 2e-6 and _rms(ys2[s] - yg2[s]) <= 2e-6, s
        pw, pg = walk[s].peaks(), gen[s].peaks()
        assert abs(pw[0] - pg[0]) <= 1e-5 and abs(pw[1] - pg[1]) <= 1e-5
    for s in (0, 1):
        y64 = oracle.linear_convolution_f64(xs[s], hd, channels)
        assert _rms(ys[s] - y64) <= TOL and _rms(ys[s] - y64) / _rms(y64) <= TOL, s
        both = np.concatenate([ys[s], ys2[s]])
        pk = walk[s].peaks()
        assert abs(pk[0] - max(0.0, float(both.max()))) <= 1e-6 and abs(pk[1] - float(np.abs(both).max())) <= 1e-6
    sp.reset()
    assert _rms(ys[1] - sp.run(xs[1])) <= TOL


def test_block_peaks_of_a_submitted_batch(tuned, oracle):
    """fe_batch_submit_peaks: every block's signed maximum (never below 0) and maximum magnitude, reduced by K3 on the
    GPU, for small batches (pair / general kernels) and for one big enough for the duplex pipeline (walkers, DMA both
    ways); streams that do not ask get nothing; a stream's running peaks are unchanged by it."""
    import ctypes
    L = fa.lib()
    size, C = 60000, 2
    rng = np.random.default_rng(123)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
    _, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
    P = flt.block_size
    for S, nblk in ((3, 2), (5, 24), (40, 16)):
        bufs, arrs, pk = [], [], []
        streams = [flt.open_stream(nblk) for _ in range(S)]
        for s_ in streams:
            b = ctypes.c_void_p()
            assert L.fe_host_alloc(nblk * P * C * 4, ctypes.byref(b)) == 0
            assert L.fe_stream_bind_host_buffer(s_.h, b, nblk * P * C * 4) == 0
            bufs.append(b)
            a = np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_float)), shape=(nblk * P, C))
            a[:] = rng.uniform(-1, 1, (nblk * P, C)).astype(np.float32) * rng.uniform(0.2, 1.0)
            arrs.append(a)
            pk.append(np.full((nblk, 2), -7.0, np.float32))
        xs = [a.copy() for a in arrs]
        ss = (ctypes.c_void_p * S)(*[s_.h for s_ in streams])
        pp = (ctypes.c_void_p * S)(*[b.value for b in bufs])
        nn = (ctypes.c_longlong * S)(*([nblk * P] * S))
        kk = (ctypes.c_void_p * S)(*[pk[i].ctypes.data if i != 1 else None for i in range(S)])   # stream 1 does not ask
        t = ctypes.c_void_p()
        assert L.fe_batch_submit_peaks(ss, S, pp, nn, pp, kk, ctypes.byref(t)) == 0
        assert L.fe_ticket_wait(t) == 0
        for i in range(S):
            y = arrs[i].reshape(nblk, P * C)
            if i == 1:
                assert np.all(pk[i] == -7.0)
                continue
            assert np.array_equal(pk[i][:, 0], np.maximum(0.0, y.max(axis=1))), (S, i)
            assert np.array_equal(pk[i][:, 1], np.abs(y).max(axis=1)), (S, i)
            run = streams[i].peaks()
            assert run[0] == max(0.0, float(y.max())) and run[1] == float(np.abs(y).max())
        y64 = oracle.linear_convolution_f64(xs[0], dense_taps(paths, size), C)
        assert _rms(arrs[0] - y64) <= TOL
        for s_ in streams:
            s_.close()
        for b in bufs:
            L.fe_host_free(b)


def test_a_batch_the_duplex_staging_cannot_hold_runs_zero_copy(tuned, oracle):
    """engine.cpp run_duplex: the staging of the duplex pipeline is capped (FE_TUNE_DUPLEX_CAP_MB; 8 GB by default) and a
    batch beyond the cap — or one the device memory cannot be had for — does not fail: it runs on the zero-copy kernels, on
    one lane, with the same results within float32 rounding and the per-block maxima still delivered; a stream that then
    goes on (state carried) in an ordinary duplex batch continues correctly, whichever lanes the two batches used."""
    import ctypes
    L = fa.lib()
    size, C, S, nblk = 60000, 2, 40, 16                       # 40 x 16 stereo blocks: 42 MB each way, a duplex batch
    rng = np.random.default_rng(321)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(C)}
    _, flt, _ = make_pair(tuned, oracle, C, C, size, paths)
    P = flt.block_size
    streams = [flt.open_stream(nblk) for _ in range(S)]
    bufs, arrs = [], []
    for s_ in streams:
        b = ctypes.c_void_p()
        assert L.fe_host_alloc(nblk * P * C * 4, ctypes.byref(b)) == 0
        assert L.fe_stream_bind_host_buffer(s_.h, b, nblk * P * C * 4) == 0
        bufs.append(b)
        arrs.append(np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_float)), shape=(nblk * P, C)))
    x1 = [rng.uniform(-1, 1, (nblk * P, C)).astype(np.float32) for _ in range(S)]
    x2 = [rng.uniform(-1, 1, (nblk * P, C)).astype(np.float32) for _ in range(S)]
    ss = (ctypes.c_void_p * S)(*[s_.h for s_ in streams])
    pp = (ctypes.c_void_p * S)(*[b.value for b in bufs])
    nn = (ctypes.c_longlong * S)(*([nblk * P] * S))

    def batch(xs):
        pk = [np.full((nblk, 2), -7.0, np.float32) for _ in range(S)]
        kk = (ctypes.c_void_p * S)(*[p.ctypes.data for p in pk])
        for a, x in zip(arrs, xs):
            a[:] = x
        t = ctypes.c_void_p()
        assert L.fe_batch_submit_peaks(ss, S, pp, nn, pp, kk, ctypes.byref(t)) == 0, L.fe_last_error()
        assert L.fe_ticket_wait(t) == 0
        return [a.copy() for a in arrs], pk

    try:
        tuned.set_tuning(duplex_cap_mb=8)                     # 8 MB: this batch cannot be staged
        y1, pk1 = batch(x1)
        tuned.set_tuning(duplex_cap_mb=0)
        y2, pk2 = batch(x2)                                   # the ordinary duplex pipeline, state carried
        for i in (0, 7, S - 1):
            ref = oracle.linear_convolution_f64(np.concatenate([x1[i], x2[i]]), dense_taps(paths, size), C)
            assert _rms(np.concatenate([y1[i], y2[i]]) - ref) <= TOL, i
        for i in range(S):
            for y, pk in ((y1[i], pk1[i]), (y2[i], pk2[i])):
                yb = y.reshape(nblk, P * C)
                assert np.array_equal(pk[:, 0], np.maximum(0.0, yb.max(axis=1))) and np.array_equal(pk[:, 1], np.abs(yb).max(axis=1)), i
    finally:
        tuned.set_tuning(duplex_cap_mb=0)
        for s_ in streams:
            s_.close()
        for b in bufs:
            L.fe_host_free(b)
