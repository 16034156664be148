"""GPU: size-independent properties of the path at BASELINE.json's FULL sizes, checked on the device.

The oracle finishes cfg3's 64 x 2 x 256 blocks in minutes, not seconds, so beside the parity tests at oracle-sized
shapes (and test_forms_gpu.py::test_benchmarked_shape_parity, which compares the whole benchmarked batch with an
independent float64 FFT convolution) the full shapes are also held to what a causal linear time-invariant convolver must
satisfy, whatever its size:

  * impulse in -> the filter out (every path, every tap: the assembled taps come back exactly where they were put);
  * linearity: conv(a x1 + b x2) = a conv(x1) + b conv(x2);
  * causality and state: a signal cut into calls of any lengths gives the same output as one call (within the float32
    rounding of different kernel forms), and the same call pattern gives the same BITS (idempotence after reset);
  * independence: a stream's output does not depend on what else is in its batch.

Shapes: cfg3 (64 stereo streams, 262 144 taps, 256-block calls), cfg4 (96 kHz x 8 channels x 524 288 taps), cfg5's
per-GPU share is cfg3's.  Reference lines: /root/reference/sound-processor.cc:98-127 (Process), zita-fconfig.cc:74-81."""
import numpy as np
import pytest

import folve_amd as fa
from folve_amd.capi import BatchPlan, FE_DEVICE_PTRS

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROUND = 2e-6           # two float32 evaluations of the same convolution through different kernel forms


def _filter(engine, C, size, seed, full=False):
    rng = np.random.default_rng(seed)
    flt = fa.Filter(engine, C, C, size)
    taps = {}
    for i in range(C):
        for o in range(C):
            if i == o or full:
                h = rng.standard_normal(size).astype(np.float32)
                h /= np.float32(np.linalg.norm(h) * (2.0 if full else 1.0))
                flt.add(i, o, h)
                taps[(i, o)] = h
    flt.commit()
    return flt, taps


def _run(streams, xs, ys, lens=None):
    import torch
    torch.cuda.synchronize()       # the engine launches on its own HIP stream: torch's pending work on the inputs comes first
    BatchPlan(streams, [x.data_ptr() for x in xs], [y.data_ptr() for y in ys],
              lens or [int(x.shape[0]) for x in xs], FE_DEVICE_PTRS).run()


def _rms(t):
    return float((t.double() ** 2).mean().sqrt())


@pytest.mark.parametrize("shape", ["cfg3", "cfg4", "cfg3-matrix"])
def test_impulse_linearity_and_independence_at_full_size(engine, shape):
    torch = pytest.importorskip("torch")
    S, C, size, T = {"cfg3": (64, 2, 262144, 256), "cfg4": (1, 8, 524288, 256), "cfg3-matrix": (64, 2, 262144, 256)}[shape]
    flt, taps = _filter(engine, C, size, 7, full=shape == "cfg3-matrix")
    P = flt.block_size
    K = flt.partitions
    assert P == 8192 and T >= K + 2
    n = T * P
    streams = [flt.open_stream(T) for _ in range(S)]
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    x1 = [torch.rand(n, C, device="cuda", generator=g) * 2 - 1 for _ in range(S)]
    x2 = [torch.rand(n, C, device="cuda", generator=g) * 2 - 1 for _ in range(S)]
    y1 = [torch.empty(n, C, device="cuda") for _ in range(S)]
    y2 = [torch.empty(n, C, device="cuda") for _ in range(S)]
    y3 = [torch.empty(n, C, device="cuda") for _ in range(S)]
    # ---- impulse in -> filter out: stream 0 gets a unit impulse on input c at frame 5 (other streams carry noise) ----
    for c in (0, C - 1):
        for s_ in streams:
            s_.reset()
        imp = torch.zeros(n, C, device="cuda")
        imp[5, c] = 1.0
        _run(streams, [imp] + x1[1:], y1)
        out = y1[0].cpu().numpy()
        for o in range(C):
            want = np.zeros(n, np.float32)
            if (c, o) in taps:
                want[5:5 + size] = taps[(c, o)]
            err = float(np.sqrt(np.mean((out[:, o].astype(np.float64) - want) ** 2)))
            assert err <= 2e-8, (shape, c, o, err)                    # taps of ~2e-3: relative 1e-5
    # ---- linearity ----
    a, b = 0.75, -0.5
    for s_ in streams:
        s_.reset()
    _run(streams, x1, y1)
    for s_ in streams:
        s_.reset()
    _run(streams, x2, y2)
    for s_ in streams:
        s_.reset()
    _run(streams, [a * u + b * v for u, v in zip(x1, x2)], y3)
    worst = max(_rms(y3[s] - (a * y1[s] + b * y2[s])) for s in range(S))
    assert worst <= ROUND, (shape, worst)
    assert min(_rms(y1[s]) for s in range(S)) > 0.1                     # (the outputs are not trivially small)
    # ---- idempotence: the same call pattern after a reset gives the same bits ----
    for s_ in streams:
        s_.reset()
    _run(streams, x1, y3)
    assert all(torch.equal(y3[s], y1[s]) for s in range(S)), shape
    # ---- independence: stream 0 alone (another batch shape, other kernel forms) agrees within rounding ----
    if S > 1:
        streams[0].reset()
        _run(streams[:1], x1[:1], y3[:1])
        assert _rms(y3[0] - y1[0]) <= ROUND, shape
    for s_ in streams:
        s_.close()


@pytest.mark.parametrize("shape", ["cfg3", "cfg4"])
def test_any_cut_into_calls_gives_the_same_output_at_full_size(engine, shape):
    """State carried across calls (the FDL ring): 256 blocks in one call against the same blocks as calls of 1, 3, 64, 17,
    .. blocks (ring wrap-arounds included: the stream is opened for 64-block calls, its ring holds K + 64 rows)."""
    torch = pytest.importorskip("torch")
    S, C, size, T = {"cfg3": (16, 2, 262144, 256), "cfg4": (1, 8, 524288, 256)}[shape]
    flt, _ = _filter(engine, C, size, 9)
    P = flt.block_size
    n = T * P
    whole = [flt.open_stream(T) for _ in range(S)]
    cut = [flt.open_stream(64) for _ in range(S)]
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    xs = [torch.rand(n, C, device="cuda", generator=g) * 2 - 1 for _ in range(S)]
    yw = [torch.empty(n, C, device="cuda") for _ in range(S)]
    yc = [torch.empty(n, C, device="cuda") for _ in range(S)]
    _run(whole, xs, yw)
    pos = 0
    for nb in (1, 3, 64, 17, 1, 1, 40, 64, 2, 63):
        assert pos + nb <= T
        _run(cut, [x[pos * P:(pos + nb) * P] for x in xs], [y[pos * P:(pos + nb) * P] for y in yc])
        pos += nb
    assert pos == T
    worst = max(_rms(yc[s] - yw[s]) for s in range(S))
    assert worst <= ROUND, (shape, worst)
    for s_ in whole + cut:
        s_.close()
