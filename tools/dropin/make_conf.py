#!/usr/bin/env python3
"""Writes cfg3's filter (2 diagonal paths of `taps` random taps, seed 3) as filter-44100.conf + ir.wav into a directory,
the way bench.py's drop_in_threads leg does: the harness then goes through the real loader.   usage: make_conf.py DIR [taps]"""
import os
import sys

import numpy as np


def write(d, size=262144, C=2, FS=44100):
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(3)
    taps = []
    for _ in range(C):
        h = rng.standard_normal(size).astype(np.float32)
        taps.append(h / np.linalg.norm(h))
    ir = np.stack(taps, axis=1).astype(np.float64)
    ir16 = np.round(ir / np.abs(ir).max() * 0.9 * 32767).astype("<i2")
    data = ir16.tobytes()
    with open(os.path.join(d, "ir.wav"), "wb") as f:
        f.write(b"RIFF" + (36 + len(data)).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") +
                (1).to_bytes(2, "little") + (C).to_bytes(2, "little") + (FS).to_bytes(4, "little") +
                (FS * C * 2).to_bytes(4, "little") + (C * 2).to_bytes(2, "little") + (16).to_bytes(2, "little") +
                b"data" + len(data).to_bytes(4, "little") + data)
    conf = os.path.join(d, "filter-44100.conf")
    with open(conf, "w") as f:
        f.write("/convolver/new %d %d 256 %d\n" % (C, C, size))
        for c in range(C):
            f.write("/impulse/read %d %d 2e-3 0 0 0 %d ir.wav\n" % (c + 1, c + 1, c + 1))
    return conf


if __name__ == "__main__":
    print(write(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 262144))
