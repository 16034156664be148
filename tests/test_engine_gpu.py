"""GPU parity: HIP engine (through the C ABI) vs the CPU oracle and the float64 ground truth.

Tolerance: BASELINE.json north_star — <= 1e-5 RMS (absolute, signals O(1)) and
<= 1e-5 relative RMS against the reference algorithm; measured ~1e-7.
"""
import numpy as np
import pytest

import folve_amd as fa
from helpers import dense_taps, make_pair, run_engine_like_reference

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _check(y, y_oracle, y64):
    from oracle.oracle import rms
    assert np.isfinite(y).all()
    r = rms(y64)
    assert rms(y - y_oracle) <= TOL, ("vs oracle", rms(y - y_oracle))
    assert rms(y - y64) <= TOL, ("vs float64", rms(y - y64))
    if r > 0:
        assert rms(y - y64) / r <= TOL, ("relative", rms(y - y64) / r)


@pytest.mark.parametrize("size,expect_P", [(32, 64), (100, 128), (256, 256), (512, 512), (700, 1024),
                                           (1500, 2048), (4096, 4096), (4097, 8192), (20000, 8192)])
def test_block_sizes_single_block_calls(engine, oracle, size, expect_P):
    rng = np.random.default_rng(size)
    h0 = (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32)
    h1 = (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32)
    paths = {(0, 0): [(0, h0)], (1, 1): [(0, h1)]}
    sp, flt, st = make_pair(engine, oracle, 2, 2, size, paths)
    assert flt.block_size == expect_P == sp.fragm
    P = flt.block_size
    n = 5 * P + P // 3            # ragged last block
    x = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    y, peak = run_engine_like_reference(st, x)
    yo = sp.run(x)
    y64 = oracle.linear_convolution_f64(x, dense_taps(paths, size), 2)
    _check(y, yo, y64)
    assert abs(peak[0] - max(0.0, float(y.max()))) <= 1e-6
    assert abs(peak[1] - float(np.abs(y).max())) <= 1e-6
    assert abs(peak[0] - sp.max_output_value()) <= 1e-5


def test_echo_closed_form(engine, oracle):
    """demo-filters/echo/filter-44100.conf: y = 0.7 x[n] + 0.3 x[n-22050], no cross-talk."""
    paths = {(0, 0): [(0, [0.7]), (22050, [0.3])], (1, 1): [(0, [0.7]), (22050, [0.3])]}
    sp, flt, st = make_pair(engine, oracle, 2, 2, 204800, paths)
    assert flt.block_size == 8192 and flt.partitions == 25
    assert flt.path_partitions(0, 0) == 2 and flt.path_partitions(0, 1) == 0
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (3 * 8192 + 1000, 2)).astype(np.float32)
    y, _ = run_engine_like_reference(st, x)
    exp = 0.7 * x.astype(np.float64)
    exp[22050:] += 0.3 * x[:-22050]
    _check(y, sp.run(x), exp)


def test_full_matrix_links_and_accumulation(engine, oracle):
    """2x3 matrix, overlapping additions on one pair, a link, and an unfed output."""
    rng = np.random.default_rng(7)
    size = 30000
    a = (rng.standard_normal(9000) * 0.01).astype(np.float32)
    b = (rng.standard_normal(12000) * 0.01).astype(np.float32)
    c = (rng.standard_normal(5000) * 0.01).astype(np.float32)
    paths = {(0, 0): [(0, a), (4000, b), (29999, [0.5])], (1, 0): [(100, c)], (1, 1): [(8192, c)]}
    links = [(0, 0, 0, 1)]          # (0,1) shares (0,0)
    sp, flt, st = make_pair(engine, oracle, 2, 3, size, paths, links, max_blocks=3)
    h = dense_taps(paths, size)
    h[(0, 1)] = h[(0, 0)]
    x = rng.uniform(-1, 1, (7 * 8192 + 77, 2)).astype(np.float32)
    y = st.process_blocks(x)       # split internally into calls of <= 3 blocks
    y64 = oracle.linear_convolution_f64(x, h, 3)
    _check(y, sp.run(x), y64)
    assert np.all(y[:, 2] == 0.0)  # output 3 has no path


@pytest.mark.parametrize("max_blocks", [1, 2, 5, 16, 40])
def test_multi_block_equals_single_block(engine, oracle, max_blocks):
    rng = np.random.default_rng(max_blocks)
    size = 70000
    h = (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32)
    paths = {(0, 0): [(0, h)]}
    sp, flt, st = make_pair(engine, oracle, 1, 1, size, paths, max_blocks=max_blocks)
    x = rng.uniform(-1, 1, (23 * 8192 + 5, 1)).astype(np.float32)
    y = st.process_blocks(x)
    _check(y, sp.run(x), oracle.linear_convolution_f64(x, dense_taps(paths, size), 1))


def test_reset_replays_identically_and_state_carries(engine, oracle):
    rng = np.random.default_rng(3)
    size = 50000
    h = (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32)
    paths = {(0, 0): [(0, h)], (1, 1): [(0, h[::-1].copy())]}
    sp, flt, st = make_pair(engine, oracle, 2, 2, size, paths, max_blocks=4)
    x = rng.uniform(-1, 1, (6 * 8192, 2)).astype(np.float32)
    y1 = st.process_blocks(x)
    assert st.blocks_done() == 6
    y2 = st.process_blocks(x)      # state carried: differs from y1 (tail of the first pass)
    assert not np.array_equal(y1, y2)
    st.reset()
    assert st.peaks() == (0.0, 0.0)
    y3 = st.process_blocks(x)
    assert np.array_equal(y1, y3)  # bit-identical after reset


def test_batch_of_ragged_streams_matches_individual_runs(engine, oracle):
    rng = np.random.default_rng(11)
    size = 40000
    h = (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32)
    paths = {(0, 0): [(0, h)], (1, 1): [(0, h)]}
    _, flt, _ = make_pair(engine, oracle, 2, 2, size, paths)
    lens = [8192 * 4, 8192 * 2 + 17, 100, 8192 * 7 + 8191, 8192]
    xs = [rng.uniform(-1, 1, (n, 2)).astype(np.float32) for n in lens]
    batch_streams = [flt.open_stream(3) for _ in lens]
    ys = fa.batch_process(batch_streams, xs)
    for x, y in zip(xs, ys):
        solo = flt.open_stream(3)
        assert np.array_equal(solo.process_blocks(x), y)    # bit-for-bit
        y64 = oracle.linear_convolution_f64(x, dense_taps(paths, size), 2)
        assert oracle.rms(y - y64) <= TOL


def test_streams_of_two_filters_in_one_batch(engine, oracle):
    rng = np.random.default_rng(5)
    pa = {(0, 0): [(0, (rng.standard_normal(3000) * 0.02).astype(np.float32))]}
    pb = {(0, 0): [(0, (rng.standard_normal(9000) * 0.02).astype(np.float32))], (0, 1): [(5, [1.0])]}
    spa, fa_, sa = make_pair(engine, oracle, 1, 1, 3000, pa, max_blocks=2)
    spb, fb_, sb = make_pair(engine, oracle, 1, 2, 9000, pb, max_blocks=2)
    xa = rng.uniform(-1, 1, (3 * 4096 + 9, 1)).astype(np.float32)
    xb = rng.uniform(-1, 1, (2 * 8192 + 1, 1)).astype(np.float32)
    ya, yb = fa.batch_process([sa, sb], [xa, xb])
    assert oracle.rms(ya - spa.run(xa)) <= TOL
    assert oracle.rms(yb - spb.run(xb)) <= TOL


def test_streams_of_four_filters_ragged_in_one_batch_bit_identical_to_per_filter_calls(engine, oracle):
    """A batch that mixes filters (the reference resolves one configuration per rate / channels / bits,
    /root/reference/processor-pool.cc:53-61) is launched in per-filter groups — on both launch lanes when the call holds
    several (engine.cpp run_groups).  Whatever lane a group lands on, its streams' outputs are the SAME BITS as when that
    filter's streams are batched alone; 4 filters (K = 1 / 3 / 5 / 9), 3 streams each, ragged lengths, two calls with state
    carried, through the host-pointer, the device-pointer and the submitted zero-copy forms."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(77)
    sizes = [3000, 20000, 40000, 70000]
    filters, ref_procs = [], []
    for size in sizes:
        paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
        filters.append(paths)
    lens = [[5 * 8192 + 77, 2 * 8192, 8192 - 5], [3 * 8192 + 1, 9 * 8192 + 4000, 100], [8192, 6 * 8192 + 3, 4 * 8192 + 4095]]
    P_of = []
    # inputs: [filter][stream] -> two consecutive calls
    xs = []
    for fi, size in enumerate(sizes):
        row = []
        for si in range(3):
            n = lens[si][fi % 3] if size > 4096 else lens[si][fi % 3] // 2 + 13
            row.append([rng.uniform(-1, 1, (n, 2)).astype(np.float32), rng.uniform(-1, 1, (n // 2 + 7, 2)).astype(np.float32)])
        xs.append(row)

    def fresh():
        out = []
        for fi, size in enumerate(sizes):
            sp, flt, st = make_pair(engine, oracle, 2, 2, size, filters[fi], max_blocks=4)
            more = [flt.open_stream(4) for _ in range(2)]
            out.append((sp, flt, [st] + more))
        return out

    # (a) per filter, alone
    alone = []
    for fi, (sp, flt, sts) in enumerate(fresh()):
        P_of.append(flt.block_size)
        first = fa.batch_process(sts, [xs[fi][si][0] for si in range(3)])
        second = fa.batch_process(sts, [xs[fi][si][1] for si in range(3)])
        alone.append((first, second))
        # (and right: the first stream against the oracle over both calls; a call's short last block is zero-padded —
        # time advances by whole blocks — so the oracle sees the padded concatenation)
        P = flt.block_size
        a, b = xs[fi][0]
        pad = (-len(a)) % P
        both = np.concatenate([a, np.zeros((pad, 2), np.float32), b])
        yo = sp.run(both)
        assert oracle.rms(first[0] - yo[:len(a)]) <= TOL and oracle.rms(second[0] - yo[len(a) + pad:]) <= TOL
    assert P_of == [4096, 8192, 8192, 8192]
    # (b) all twelve streams interleaved in one call, host pointers
    mixed = fresh()
    order = [(fi, si) for si in range(3) for fi in range(4)]
    sts = [mixed[fi][2][si] for fi, si in order]
    y1 = fa.batch_process(sts, [xs[fi][si][0] for fi, si in order])
    y2 = fa.batch_process(sts, [xs[fi][si][1] for fi, si in order])
    for k, (fi, si) in enumerate(order):
        assert np.array_equal(y1[k], alone[fi][0][si]) and np.array_equal(y2[k], alone[fi][1][si]), (fi, si)
    # (c) the same through device pointers (what bench.py's mixed_filters leg times)
    mixed = fresh()
    sts = [mixed[fi][2][si] for fi, si in order]
    for call in range(2):
        xd = [torch.from_numpy(xs[fi][si][call]).cuda() for fi, si in order]
        yd = [torch.empty_like(t) for t in xd]
        fa.batch_process(sts, xd, yd, device=True)
        for k, (fi, si) in enumerate(order):
            assert np.array_equal(yd[k].cpu().numpy(), alone[fi][call][si]), (call, fi, si)


def test_device_pointer_batch(engine, oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(13)
    size = 262144
    h = rng.standard_normal(size).astype(np.float32)
    h /= np.linalg.norm(h)
    paths = {(0, 0): [(0, h)], (1, 1): [(0, h)]}
    sp, flt, st = make_pair(engine, oracle, 2, 2, size, paths, max_blocks=8)
    assert flt.partitions == 32
    x = rng.uniform(-1, 1, (8 * 8192, 2)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    fa.batch_process([st], [xd], [yd], device=True)
    y = yd.cpu().numpy()
    _check(y, sp.run(x), oracle.linear_convolution_f64(x, dense_taps(paths, size), 2))
