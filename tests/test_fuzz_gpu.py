"""GPU: randomized call patterns against the oracle's engine restatement (block by block).

Every call ends in a (possibly short, zero-padded) block and advances time by whole blocks — the
engine's contract, which is the reference's Process() contract — so the oracle is driven the same way:
one oc_process per block with the tail zero-filled."""
import numpy as np
import pytest

import folve_amd as fa

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _oracle_call(conv, x, P):
    """x: [frames, ninp] -> [frames, nout], one block at a time, zero-padded tail."""
    out = np.zeros((x.shape[0], conv.nout), np.float32)
    for a in range(0, x.shape[0], P):
        blk = np.zeros((P, conv.ninp), np.float32)
        n = min(P, x.shape[0] - a)
        blk[:n] = x[a:a + n]
        y = conv.process_block(blk.T.copy())
        out[a:a + n] = y.T[:n]
    return out


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_filters_and_call_patterns(engine, oracle, seed):
    rng = np.random.default_rng(seed)
    ninp, nout = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    size = int(rng.choice([40, 300, 1000, 3000, 5000, 9000, 20000, 40000]))
    conv = oracle.Convproc(ninp, nout, size)
    flt = fa.Filter(engine, ninp, nout, size)
    npaths = int(rng.integers(1, ninp * nout + 1))
    pairs = [(int(i), int(o)) for i in range(ninp) for o in range(nout)]
    rng.shuffle(pairs)
    for (i, o) in pairs[:npaths]:
        for _ in range(int(rng.integers(1, 3))):                     # accumulating additions
            n = int(rng.integers(1, size + 1))
            ind0 = int(rng.integers(0, size - n + 1))
            taps = (rng.standard_normal(n) / np.sqrt(n) * 0.5).astype(np.float32)
            conv.impdata_create(i, o, taps, ind0)
            flt.add(i, o, taps, ind0)
    if npaths < len(pairs) and rng.random() < 0.5:                   # a link onto an unused pair
        (i2, o2), (i1, o1) = pairs[npaths], pairs[0]
        conv.impdata_copy(i1, o1, i2, o2)
        flt.link(i1, o1, i2, o2)
    flt.commit()
    P = flt.block_size
    maxb = int(rng.integers(1, 7))
    streams = [flt.open_stream(maxb) for _ in range(3)]
    convs = None
    worst = 0.0
    scale = 0.0
    for round_ in range(6):
        if round_ == 3:                                              # reset one stream mid-way
            streams[1].reset()
        lens = [int(rng.integers(0, 4 * P + 1)) for _ in streams]
        xs = [rng.uniform(-1, 1, (n, ninp)).astype(np.float32) for n in lens]
        ys = fa.batch_process(streams, xs)
        if convs is None:
            convs = [conv] + [None, None]
        for k, (x, y) in enumerate(zip(xs, ys)):
            if k != 0 or x.shape[0] == 0:
                continue                                             # stream 0 is mirrored in the oracle
            yo = _oracle_call(conv, x, P)
            worst = max(worst, float(np.sqrt(np.mean((y.astype(np.float64) - yo) ** 2))) if x.shape[0] else 0.0)
            scale = max(scale, float(np.abs(yo).max()))
        assert all(np.isfinite(y).all() for y in ys)
    assert worst <= TOL, (worst, scale)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18])
def test_random_one_block_calls_on_a_bound_buffer(engine, oracle, seed):
    """The drop-in call pattern (SoundProcessor::Process, /root/reference/sound-processor.cc:98-127): one
    synchronous block at a time, in place, on a page-locked buffer bound to the stream — the zero-copy path
    with its latency kernels (stereo: the 1024-thread pair forms) — against the oracle block by block;
    a short last block, then reset and a second file."""
    import ctypes
    L = fa.lib()
    rng = np.random.default_rng(seed)
    ninp, nout = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    if seed % 2 == 0:
        ninp = nout = 2
    size = int(rng.choice([5000, 9000, 20000, 40000, 70000]))
    conv = oracle.Convproc(ninp, nout, size)
    flt = fa.Filter(engine, ninp, nout, size)
    for i in range(ninp):
        for o in range(nout):
            if i == o or rng.random() < 0.4:
                n = int(rng.integers(1, size + 1))
                ind0 = int(rng.integers(0, size - n + 1))
                taps = (rng.standard_normal(n) / np.sqrt(n) * 0.5).astype(np.float32)
                conv.impdata_create(i, o, taps, ind0)
                flt.add(i, o, taps, ind0)
    flt.commit()
    P = flt.block_size
    C = max(ninp, nout)
    buf = ctypes.c_void_p()
    assert L.fe_host_alloc(P * C * 4, ctypes.byref(buf)) == 0
    flat = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), shape=(P * C,))
    st = flt.open_stream(1)
    try:
        assert L.fe_stream_bind_host_buffer(st.h, buf, P * C * 4) == 0
        worst = 0.0
        for _file in range(2):
            nblocks = int(rng.integers(2, 7))
            for b in range(nblocks):
                v = P if b + 1 < nblocks else int(rng.integers(1, P + 1))
                x = rng.uniform(-1, 1, (v, ninp)).astype(np.float32)
                flat[:] = 0
                flat[:v * ninp] = x.reshape(-1)                          # interleaved [frames][ninp], as buffer_ holds it
                assert L.fe_stream_process(st.h, buf, v, buf, None, None) == 0
                y = flat[:v * nout].reshape(v, nout).copy()
                yo = _oracle_call(conv, x, P)
                worst = max(worst, float(np.sqrt(np.mean((y.astype(np.float64) - yo) ** 2))))
                assert np.isfinite(y).all()
            st.reset()
            conv.reset()
        assert worst <= TOL, worst
    finally:
        st.close()
        L.fe_host_free(buf)
