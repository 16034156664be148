"""Shared builders for the parity tests: the same filter in the oracle and in the engine."""
import numpy as np

import folve_amd as fa


def make_pair(engine, O, ninp, nout, size, paths, links=(), max_blocks=1):
    """paths: {(inp, out): [(ind0, taps), ...]}; links: [(inp1, out1, inp2, out2)].
    Returns (oracle SoundProcessor, engine Filter, engine Stream)."""
    conv = O.Convproc(ninp, nout, size)
    flt = fa.Filter(engine, ninp, nout, size)
    for (i, o), chunks in paths.items():
        for ind0, taps in chunks:
            taps = np.asarray(taps, np.float32)
            conv.impdata_create(i, o, taps, ind0)
            flt.add(i, o, taps, ind0)
    for (i1, o1, i2, o2) in links:
        conv.impdata_copy(i1, o1, i2, o2)
        flt.link(i1, o1, i2, o2)
    flt.commit()
    return O.SoundProcessor.wrap(conv), flt, flt.open_stream(max_blocks)


def dense_taps(paths, size):
    """{(i,o): [(ind0, taps)]} -> {(i,o): dense float32 h[size]} (float32 accumulation like the engine)."""
    out = {}
    for key, chunks in paths.items():
        h = np.zeros(size, np.float32)
        for ind0, taps in chunks:
            taps = np.asarray(taps, np.float32)
            n = min(len(taps), size - ind0)
            h[ind0:ind0 + n] += taps[:n]
        out[key] = h
    return out


def run_engine_like_reference(stream, x):
    """Feed x block by block through fe_stream_process (SoundProcessor::Process semantics)."""
    P = stream.filter.block_size
    outs = []
    peak = (0.0, 0.0)
    for a in range(0, x.shape[0], P):
        y, ps, pa = stream.process(x[a:a + P])
        outs.append(y)
        peak = (ps, pa)
    return np.concatenate(outs, 0), peak
