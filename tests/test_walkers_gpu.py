"""GPU: large batches (>= 256 blocks per launch) through the fast kernel forms — the stereo forward
walker (forward_walker_kernel<13>), the one-transform stereo form of the shorter blocks
(forward_dual_kernel<11|12>: one 2P-point complex FFT per block), the mono forward path, and the
inverse walker (inverse_walker_kernel<13, 1|2>) — against the oracle and against the general
kernels (the same streams run one at a time in small calls)."""
import numpy as np
import pytest

import folve_amd as fa
from helpers import dense_taps, make_pair

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


@pytest.mark.parametrize("channels,size,nstreams,nblocks", [
    (2, 20000, 16, 17),     # P = 8192, K = 3: forward_walker<13> + inverse_walker<13,2>
    (1, 20000, 16, 17),     # mono: general forward + inverse_walker<13,1>
    (2, 4000, 16, 20),      # P = 4096: forward_dual<12>, general inverse
    (1, 3000, 20, 16),      # P = 4096 mono
    (2, 1500, 32, 12),      # P = 2048: forward_dual<11>
    (1, 2000, 16, 20),      # P = 2048 mono
    (2, 262144, 11, 24),    # cfg3 shape, K = 32: two time tiles, the downward-walking one half full
])
def test_large_batches_match_oracle_and_general_kernels(engine, oracle, channels, size, nstreams, nblocks):
    rng = np.random.default_rng(size + channels)
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(channels)}
    if channels == 2:
        paths[(0, 1)] = [(7, (rng.standard_normal(size // 3) * 0.02).astype(np.float32))]   # a cross path
    sp, flt, _ = make_pair(engine, oracle, channels, channels, size, paths)
    P = flt.block_size
    assert nstreams * nblocks >= 256
    lens = [nblocks * P - (0 if s % 3 else 1234 + s) for s in range(nstreams)]              # some ragged tails
    xs = [rng.uniform(-1, 1, (n, channels)).astype(np.float32) for n in lens]
    big = [flt.open_stream(nblocks) for _ in range(nstreams)]
    ys = fa.batch_process(big, xs)                                   # one launch round: fast forms
    hd = dense_taps(paths, size)
    for s in (0, 1, nstreams - 1):
        sp.reset()
        yo = sp.run(xs[s])
        assert _rms(ys[s] - yo) <= TOL
        y64 = oracle.linear_convolution_f64(xs[s], hd, channels)
        assert _rms(ys[s] - y64) <= TOL and _rms(ys[s] - y64) / _rms(y64) <= TOL
    # the same streams through the general kernels (single stream, few blocks per call)
    for s in (2, 5):
        small = flt.open_stream(2)
        yg = small.process_blocks(xs[s])
        assert _rms(ys[s] - yg) <= 2e-6
    # state carries across walker calls: second call continues the convolution
    more = [rng.uniform(-1, 1, (3 * P + 5, channels)).astype(np.float32) for _ in range(nstreams)]
    ys2 = fa.batch_process(big, more)
    full = oracle.linear_convolution_f64(np.concatenate([np.pad(xs[1], ((0, nblocks * P - lens[1]), (0, 0))), more[1]]),
                                         hd, channels)
    assert _rms(ys2[1] - full[nblocks * P:]) <= TOL
    pk = big[1].peaks()
    both = np.concatenate([ys[1], ys2[1]])
    assert abs(pk[0] - max(0.0, float(both.max()))) <= 1e-6 and abs(pk[1] - float(np.abs(both).max())) <= 1e-6


def test_unaligned_device_pointers_fall_back_to_general_kernels(engine, oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(9)
    size = 20000
    paths = {(c, c): [(0, (rng.standard_normal(size) / np.sqrt(size)).astype(np.float32))] for c in range(2)}
    _, flt, _ = make_pair(engine, oracle, 2, 2, size, paths)
    P, S, T = flt.block_size, 16, 16
    x = rng.uniform(-1, 1, (S, T * P + 1, 2)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    out_a = torch.zeros(S, T * P, 2, device="cuda")
    out_u = torch.zeros(S, T * P + 1, 2, device="cuda")
    sa = [flt.open_stream(T) for _ in range(S)]
    su = [flt.open_stream(T) for _ in range(S)]
    aligned_in = [xd[s, :T * P].contiguous() for s in range(S)]
    fa.batch_process(sa, aligned_in, [out_a[s] for s in range(S)], device=True)             # fast forms
    # views starting one frame (8 bytes) into the buffers: not 16-byte aligned
    fa.batch_process(su, [xd[s, 1:] for s in range(S)], [out_u[s, 1:] for s in range(S)], device=True)
    ya = out_a.cpu().numpy()
    yu = out_u.cpu().numpy()[:, 1:]
    hd = dense_taps(paths, size)
    y64 = oracle.linear_convolution_f64(x[3, :T * P], hd, 2)
    assert _rms(ya[3] - y64) <= TOL
    y64u = oracle.linear_convolution_f64(x[3, 1:], hd, 2)
    assert _rms(yu[3] - y64u) <= TOL


def test_pipelined_host_batch_mixed_filters_and_rounds(engine, oracle):
    """A host-pointer batch above 16 MB is cut into chunks of streams that ride the bus in, compute
    and ride out on three HIP streams (engine.cpp run_pipelined).  Streams of two filters interleaved,
    ragged lengths, more blocks than a launch round carries, page-locked and pageable callers: every
    stream equals its solo run to rounding and the float64 convolution to the parity bound."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(23)
    sizes = (20000, 9000)                                              # K = 3 and K = 2 at P = 8192
    flts, taps = [], []
    for size in sizes:
        h = [(rng.standard_normal(size) / np.sqrt(size)).astype(np.float32) for _ in range(2)]
        f = fa.Filter(engine, 2, 2, size)
        for c in range(2):
            f.add(c, c, h[c])
        f.commit()
        flts.append(f); taps.append(h)
    P, S, maxb = 8192, 14, 6
    lens = [int(20 * P - 1000 * s - (s % 3)) for s in range(S)]        # 20 blocks: four launch rounds of <= 6
    xs = [rng.uniform(-1, 1, (n, 2)).astype(np.float32) for n in lens]
    assert sum(x.nbytes for x in xs) * 2 >= 16 << 20
    which = [s % 2 for s in range(S)]
    streams = [flts[w].open_stream(maxb) for w in which]
    ys = fa.batch_process(streams, xs)                                 # pageable numpy buffers
    pinned = [torch.from_numpy(x).pin_memory() for x in xs]
    outs = [torch.zeros(n, 2).pin_memory() for n in lens]
    streams2 = [flts[w].open_stream(maxb) for w in which]
    from folve_amd.capi import BatchPlan, FE_HOST_PTRS
    BatchPlan(streams2, [t.data_ptr() for t in pinned], [t.data_ptr() for t in outs], lens, FE_HOST_PTRS).run()
    for s in range(S):
        assert np.array_equal(ys[s], outs[s].numpy()), s               # same chunks, same kernels
    for s in (0, 1, 6, 13):
        solo = flts[which[s]].open_stream(maxb).process_blocks(xs[s])
        assert _rms(ys[s] - solo) <= 2e-6
        hd = {(c, c): taps[which[s]][c] for c in range(2)}
        y64 = oracle.linear_convolution_f64(xs[s], hd, 2)
        assert _rms(ys[s] - y64) <= TOL and _rms(ys[s] - y64) / _rms(y64) <= TOL
    # state carried: a second, small (unpipelined) call continues every stream
    more = [rng.uniform(-1, 1, (P + 3, 2)).astype(np.float32) for _ in range(S)]
    ys2 = fa.batch_process(streams, more)
    s = 5
    hd = {(c, c): taps[which[s]][c] for c in range(2)}
    pad = (-lens[s]) % P
    full = oracle.linear_convolution_f64(np.concatenate([np.pad(xs[s], ((0, pad), (0, 0))), more[s]]), hd, 2)
    assert _rms(ys2[s] - full[lens[s] + pad:]) <= TOL
