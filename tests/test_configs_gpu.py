"""GPU: BASELINE.json's configurations as parity cases (bench.py measures cfg3), plus
size-independent properties at full size and the reference's extreme sizes."""
import numpy as np
import pytest

import folve_amd as fa
from helpers import dense_taps, make_pair

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


def test_cfg2_one_stream_santalucia_shape(engine, oracle):
    """cfg2: one 44.1 kHz stereo stream, SantaLucia-shaped filter (size 204800, K = 25, 22 populated)."""
    rng = np.random.default_rng(2)
    n = 178193
    paths = {}
    for c in range(2):
        ir = rng.standard_normal(n) * np.exp(-np.arange(n) / 40000.0)
        ir = (ir / np.linalg.norm(ir)).astype(np.float32)
        paths[(c, c)] = [(500, ir), (0, [0.4])]
    sp, flt, st = make_pair(engine, oracle, 2, 2, 204800, paths, max_blocks=8)
    assert flt.partitions == 25 and flt.path_partitions(0, 0) == 22
    x = np.random.default_rng(1).uniform(-1, 1, (40 * 8192 + 8176, 2)).astype(np.float32)
    y = st.process_blocks(x)
    y64 = oracle.linear_convolution_f64(x, dense_taps(paths, 204800), 2)
    assert _rms(y - y64) <= TOL and _rms(y - y64) / _rms(y64) <= TOL
    assert _rms(y[: 12 * 8192] - sp.run(x[: 12 * 8192])) <= TOL


def test_cfg3_batch_of_streams_sharing_one_filter(engine, oracle):
    """cfg3 shape at test size: many stereo streams, one shared 2-path 262144-tap filter (K = 32)."""
    rng = np.random.default_rng(3)
    size = 262144
    paths = {}
    for c in range(2):
        h = rng.standard_normal(size).astype(np.float32)
        paths[(c, c)] = [(0, h / np.linalg.norm(h))]
    _, flt, _ = make_pair(engine, oracle, 2, 2, size, paths)
    S, T = 8, 36                                        # longer than K so every partition is exercised
    streams = [flt.open_stream(16) for _ in range(S)]
    xs = [np.random.default_rng(100 + s).uniform(-1, 1, (T * 8192, 2)).astype(np.float32) for s in range(S)]
    ys = fa.batch_process(streams, xs)
    hd = dense_taps(paths, size)
    for s in (0, 3, 7):                                 # subset recomputed in float64 on the host
        y64 = oracle.linear_convolution_f64(xs[s], hd, 2)
        assert _rms(ys[s] - y64) <= TOL and _rms(ys[s] - y64) / _rms(y64) <= TOL
    # a checksum of checksums over the whole batch: sum(y) == sum_t x * sum(h) up to edge effects is weak;
    # use linearity instead: stream(a*x0 + b*x1) == a*y0 + b*y1
    mix = flt.open_stream(16)
    ymix = mix.process_blocks(0.5 * xs[0] - 0.25 * xs[1])
    assert _rms(ymix - (0.5 * ys[0] - 0.25 * ys[1])) <= TOL


def test_cfg4_eight_channels_512k_taps(engine, oracle):
    """cfg4: 96 kHz / 8 channels, 8 diagonal 524288-tap paths (K = 64)."""
    rng = np.random.default_rng(4)
    size, C = 524288, 8
    paths = {}
    for c in range(C):
        h = rng.standard_normal(size).astype(np.float32)
        paths[(c, c)] = [(0, h / np.linalg.norm(h))]
    _, flt, st = make_pair(engine, oracle, C, C, size, paths, max_blocks=16)
    assert flt.partitions == 64 and flt.block_size == 8192
    x = rng.uniform(-1, 1, (70 * 8192 + 100, C)).astype(np.float32)
    y = st.process_blocks(x)
    hd = dense_taps(paths, size)
    for c in (0, 5, 7):
        y64 = oracle.linear_convolution_f64(x, {(c, c): hd[(c, c)]}, C)[:, c]
        assert _rms(y[:, c] - y64) <= TOL and _rms(y[:, c] - y64) / _rms(y64) <= TOL


def test_maximum_size_and_channel_matrix(engine, oracle):
    """MAXSIZE = 2^20 taps (K = 128, zita-config.h:61) and a 3x5 matrix with every pair populated."""
    rng = np.random.default_rng(5)
    size = 0x100000
    h = rng.standard_normal(size).astype(np.float32)
    h /= np.linalg.norm(h)
    paths = {(0, 0): [(0, h)]}
    _, flt, st = make_pair(engine, oracle, 1, 1, size, paths, max_blocks=8)
    assert flt.partitions == 128
    x = np.zeros((130 * 8192, 1), np.float32)
    x[5, 0] = 1.0                                        # impulse in -> h out, delayed by 5
    y = st.process_blocks(x)
    assert np.abs(y[5:5 + size, 0] - h).max() <= 2e-6
    assert np.abs(y[5 + size:, 0]).max() <= 2e-6 and np.abs(y[:5, 0]).max() <= 1e-7
    # dense 3 -> 5 matrix, short taps
    pm = {(i, o): [(0, (rng.standard_normal(700) * 0.05).astype(np.float32))] for i in range(3) for o in range(5)}
    sp, flt2, st2 = make_pair(engine, oracle, 3, 5, 700, pm, max_blocks=4)
    assert flt2.block_size == 1024
    x = rng.uniform(-1, 1, (9 * 1024 + 7, 3)).astype(np.float32)
    y = st2.process_blocks(x)
    assert _rms(y - sp.run(x)) <= TOL
    assert _rms(y - oracle.linear_convolution_f64(x, dense_taps(pm, 700), 5)) <= TOL


def test_64_by_64_channels(engine, oracle):
    """MAXINP x MAXOUT = 64 x 64 (zita-fconfig.cc:49,55) with a sparse set of pairs."""
    rng = np.random.default_rng(6)
    pm = {}
    for o in range(64):
        for i in ((o * 7) % 64, (o * 11 + 3) % 64):
            pm[(i, o)] = [(int(rng.integers(0, 50)), (rng.standard_normal(60) * 0.1).astype(np.float32))]
    sp, flt, st = make_pair(engine, oracle, 64, 64, 128, pm, max_blocks=2)
    assert flt.block_size == 128
    x = rng.uniform(-1, 1, (5 * 128 + 3, 64)).astype(np.float32)
    y = st.process_blocks(x)
    assert _rms(y - sp.run(x)) <= TOL
    assert _rms(y - oracle.linear_convolution_f64(x, dense_taps(pm, 128), 64)) <= TOL


def test_empty_and_degenerate_inputs(engine, oracle):
    paths = {(0, 0): [(0, [1.0, 0.5])]}
    sp, flt, st = make_pair(engine, oracle, 1, 1, 64, paths, max_blocks=2)
    assert st.process_blocks(np.zeros((0, 1), np.float32)).shape == (0, 1)     # empty call: no-op
    assert fa.batch_process([], []) == []
    y = st.process_blocks(np.ones((1, 1), np.float32))                         # one frame
    assert y.shape == (1, 1) and abs(y[0, 0] - 1.0) < 1e-6
    with pytest.raises(fa.FolveError):
        st.process(np.zeros((65, 1), np.float32))                              # more than a block
    with pytest.raises(fa.FolveError):
        fa.batch_process([st, st], [np.zeros((4, 1), np.float32)] * 2)         # a stream twice in one batch
    # a filter with no impulse at all: outputs are silence
    flt0 = fa.Filter(engine, 2, 2, 1000).commit()
    y0 = flt0.open_stream(1).process_blocks(np.ones((100, 2), np.float32))
    assert np.all(y0 == 0)
