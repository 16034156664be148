#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's demo filters.  Runs ONLY in the
authoring container (needs /root/reference); the fixtures it writes are data: tap
values, seeded-input output samples and checksums — no reference source text.

  demo_lowpass.npz / demo_highpass.npz
      the non-zero int16 taps of lowpass_44.wav / highpass_44.wav (<= 131 per channel,
      /root/reference/demo-filters/{low,high}pass/*.wav), the /impulse/read parameters
      of their filter-44100.conf (gain, channel), and float64 expected output of a
      seeded stereo signal: y_c = x_c * (gain * wav[:, chan-1] / 32768).
  demo_echo.npz
      parameters of echo/filter-44100.conf and filter-192000.conf (closed form).
  demo_santalucia.npz
      checksums of the assembled impulse response of SantaLucia/filter-44100.conf
      (sum, l2, selected taps, populated partitions) and ~1.2k output samples around
      block seams for a seeded input, computed in float64 from santalucia.wav.
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/demo-filters"
OUT = os.path.join(ROOT, "tests", "golden")


def read_wav16(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fmt = 12, None
    while pos < len(b):
        cid, ln = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
        elif cid == b"data":
            assert fmt[0] == 1 and fmt[5] == 16
            d = np.frombuffer(b[pos + 8:pos + 8 + ln], "<i2").reshape(-1, fmt[1])
            return fmt[2], d
        pos += 8 + ln + (ln & 1)
    raise ValueError(path)


def seeded_input(seed, frames, ch):
    return np.random.default_rng(seed).uniform(-1, 1, (frames, ch)).astype(np.float32)


def conv64(x, h):
    from scipy.signal import fftconvolve
    nz = np.flatnonzero(h)
    return fftconvolve(x.astype(np.float64), h[: nz[-1] + 1].astype(np.float64))[: len(x)]


def seam_indices(frames):
    parts = [np.arange(0, 200), np.arange(frames - 200, frames)]
    for seam in range(8192, frames, 8192):
        parts.append(np.arange(seam - 100, min(seam + 100, frames)))
    return np.unique(np.concatenate(parts))


def main():
    os.makedirs(OUT, exist_ok=True)
    frames = 3 * 8192 + 1000
    for name, gain in (("lowpass", 0.75), ("highpass", 0.55)):
        rate, d = read_wav16("%s/%s/%s_44.wav" % (REF, name, name))
        nz = np.flatnonzero(np.abs(d).sum(1))
        ntaps = int(nz[-1]) + 1
        taps = d[:ntaps].copy()                       # int16 [ntaps, 2]
        x = seeded_input(21, frames, 2)
        h = (np.float32(gain) * (d[:, 0].astype(np.float32) / np.float32(32768.0))).astype(np.float32)
        y = np.stack([conv64(x[:, c], h) for c in range(2)], 1)
        idx = seam_indices(frames)
        np.savez_compressed(os.path.join(OUT, "demo_%s.npz" % name), taps_int16=taps, wav_frames=d.shape[0],
                            wav_rate=rate, gain=np.float32(gain), file_channel=1, size=65536, seed=21,
                            frames=frames, out_idx=idx, out_expected=y[idx], out_rms=float(np.sqrt(np.mean(y * y))), h_sum=float(h.astype(np.float64).sum()),
                            h_l2=float(np.linalg.norm(h.astype(np.float64))))
        print(name, "taps", ntaps, "sum", h.sum(), "l2", np.linalg.norm(h))
    np.savez_compressed(os.path.join(OUT, "demo_echo.npz"), size=204800, gains=np.float32([0.7, 0.3]),
                        delay_44100=22050, delay_192000=96000)
    # SantaLucia: h_c[0] += 0.4 ; h_c[500 + i] += 4e-3 * wav[1400 + i, c] / 32768
    rate, d = read_wav16("%s/SantaLucia/santalucia.wav" % REF)
    size, delay, offset = 204800, 500, 1400
    n = d.shape[0] - offset
    hs = []
    for c in range(2):
        h = np.zeros(size, np.float32)
        h[delay:delay + n] += np.float32(4e-3) * (d[offset:, c].astype(np.float32) / np.float32(32768.0))
        h[0] += np.float32(0.4)
        hs.append(h)
    x = seeded_input(22, 4 * 8192 + 500, 2)
    y = np.stack([conv64(x[:, c], hs[c]) for c in range(2)], 1)
    idx = seam_indices(len(x))
    np.savez_compressed(os.path.join(OUT, "demo_santalucia.npz"), wav_frames=d.shape[0], wav_rate=rate, size=size,
                        delay=delay, offset=offset, ntaps=n, last_tap=delay + n - 1,
                        populated_partitions=int((delay + n - 1) // 8192 + 1),
                        h_sum=np.array([h.astype(np.float64).sum() for h in hs]),
                        h_l2=np.array([np.linalg.norm(h.astype(np.float64)) for h in hs]),
                        h_probe_idx=np.array([0, 500, 501, 9000, 100000, delay + n - 1]),
                        h_probe=np.array([[h[i] for i in (0, 500, 501, 9000, 100000, delay + n - 1)] for h in hs], np.float32),
                        seed=22, frames=len(x), out_idx=idx, out_expected=y[idx],
                        out_rms=float(np.sqrt(np.mean(y * y))))
    print("santalucia", d.shape, "last tap", delay + n - 1, "partitions", (delay + n - 1) // 8192 + 1)


if __name__ == "__main__":
    main()
