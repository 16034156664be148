"""Materialise filter directories from tests/golden/*.npz (no access to /root/reference needed)."""
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_DEMO = "/root/reference/demo-filters"


def write_wav(path, data, rate=44100, fmt="pcm16"):
    """data: [frames, ch]; int16 for pcm16, float for the others."""
    data = np.asarray(data)
    if data.ndim == 1:
        data = data[:, None]
    ch = data.shape[1]
    fmt_ext = fmt.startswith("wavex-")              # "wavex-pcm24": the same samples in a WAVE_FORMAT_EXTENSIBLE file
    if fmt_ext:
        fmt = fmt[6:]
    if fmt == "pcm16":
        raw, tag, bits = data.astype("<i2").tobytes(), 1, 16
    elif fmt == "pcm24":
        v = np.clip(np.round(data * 8388608.0), -8388608, 8388607).astype(np.int32)
        b = v.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        raw, tag, bits = b.tobytes(), 1, 24
    elif fmt == "pcm32":
        raw, tag, bits = np.clip(np.round(data * 2147483648.0), -2**31, 2**31 - 1).astype("<i4").tobytes(), 1, 32
    elif fmt == "float32":
        raw, tag, bits = data.astype("<f4").tobytes(), 3, 32
    elif fmt == "pcm8":
        raw, tag, bits = np.clip(np.round(data * 128.0) + 128, 0, 255).astype(np.uint8).tobytes(), 1, 8
    else:
        raise ValueError(fmt)
    align = ch * bits // 8
    if fmt_ext:
        # WAVE_FORMAT_EXTENSIBLE: 40-byte fmt chunk, the real format tag is the first word of the sub-format GUID
        guid = struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        hdr = struct.pack("<4sI4s4sIHHIIHHHHI", b"RIFF", 60 + len(raw), b"WAVE", b"fmt ", 40, 0xFFFE, ch, rate,
                          rate * align, align, bits, 22, bits, (1 << ch) - 1) + guid + struct.pack("<4sI", b"data", len(raw))
        with open(path, "wb") as f:
            f.write(hdr + raw)
        return
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(raw), b"WAVE", b"fmt ", 16, tag, ch, rate,
                      rate * align, align, bits, b"data", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def _ext80(x):
    """IEEE 754 80-bit extended, big-endian (the AIFF sample-rate field)."""
    import math
    if x == 0:
        return b"\0" * 10
    m, e = math.frexp(x)                                   # x = m * 2**e, 0.5 <= m < 1
    return struct.pack(">HQ", e - 1 + 16383, int(m * (1 << 64)))


def write_aiff(path, data, rate=44100, fmt="pcm16"):
    """AIFF (pcm8/pcm16/pcm24/pcm32, big-endian) or AIFF-C (sowt16 = little-endian PCM, fl32).  data: float [frames, ch]."""
    data = np.asarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    frames, ch = data.shape
    comp = None
    if fmt == "pcm8":
        raw, bits = np.clip(np.round(data * 128.0), -128, 127).astype(np.int8).tobytes(), 8
    elif fmt == "pcm16":
        raw, bits = np.clip(np.round(data * 32768.0), -32768, 32767).astype(">i2").tobytes(), 16
    elif fmt == "pcm24":
        v = np.clip(np.round(data * 8388608.0), -8388608, 8388607).astype(">i4")
        raw, bits = v.view(np.uint8).reshape(-1, 4)[:, 1:].tobytes(), 24
    elif fmt == "pcm32":
        raw, bits = np.clip(np.round(data * 2147483648.0), -2**31, 2**31 - 1).astype(">i4").tobytes(), 32
    elif fmt == "sowt16":
        raw, bits, comp = np.clip(np.round(data * 32768.0), -32768, 32767).astype("<i2").tobytes(), 16, b"sowt"
    elif fmt == "fl32":
        raw, bits, comp = data.astype(">f4").tobytes(), 32, b"fl32"
    else:
        raise ValueError(fmt)
    comm = struct.pack(">hIh", ch, frames, bits) + _ext80(float(rate))
    if comp:
        comm += comp + b"\x00\x00"                          # empty pascal string + pad
    chunks = b"COMM" + struct.pack(">I", len(comm)) + comm
    chunks += b"ANNO" + struct.pack(">I", 3) + b"abc\0"    # an odd-sized chunk before the samples
    chunks += b"SSND" + struct.pack(">III", len(raw) + 8, 0, 0) + raw + (b"\0" if len(raw) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"FORM" + struct.pack(">I", 4 + len(chunks)) + (b"AIFC" if comp else b"AIFF") + chunks)


def write_caf(path, data, rate=44100, fmt="f32le", open_ended=False):
    """Core Audio Format, linear PCM: f32le, f32be, i16be, i24le, i8.  data: float [frames, ch]."""
    data = np.asarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    ch = data.shape[1]
    if fmt == "f32le":
        raw, flags, bits = data.astype("<f4").tobytes(), 3, 32
    elif fmt == "f32be":
        raw, flags, bits = data.astype(">f4").tobytes(), 1, 32
    elif fmt == "i16be":
        raw, flags, bits = np.clip(np.round(data * 32768.0), -32768, 32767).astype(">i2").tobytes(), 0, 16
    elif fmt == "i24le":
        v = np.clip(np.round(data * 8388608.0), -8388608, 8388607).astype("<i4")
        raw, flags, bits = v.view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 2, 24
    elif fmt == "i8":
        raw, flags, bits = np.clip(np.round(data * 128.0), -128, 127).astype(np.int8).tobytes(), 0, 8
    else:
        raise ValueError(fmt)
    desc = struct.pack(">d4sIIIII", float(rate), b"lpcm", flags, ch * bits // 8, 1, ch, bits)
    body = b"desc" + struct.pack(">q", len(desc)) + desc
    body += b"free" + struct.pack(">q", 5) + b"\0" * 5       # CAF chunks are not padded
    body += b"data" + struct.pack(">q", -1 if open_ended else len(raw) + 4) + struct.pack(">I", 0) + raw
    with open(path, "wb") as f:
        f.write(b"caff" + struct.pack(">HH", 1, 0) + body)


def _wav_body(data, fmt):
    data = np.asarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    if fmt == "pcm16":
        return np.clip(np.round(data * 32768.0), -32768, 32767).astype("<i2").tobytes(), 1, 16, data.shape[1]
    if fmt == "pcm24":
        v = np.clip(np.round(data * 8388608.0), -8388608, 8388607).astype("<i4")
        return v.view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 1, 24, data.shape[1]
    if fmt == "float32":
        return data.astype("<f4").tobytes(), 3, 32, data.shape[1]
    raise ValueError(fmt)


def write_rf64(path, data, rate=44100, fmt="pcm16", magic=b"RF64"):
    """RF64 / BW64 (EBU Tech 3306): RIFF with a 'ds64' chunk carrying the 64-bit sizes; the 32-bit fields say 0xffffffff."""
    raw, tag, bits, ch = _wav_body(data, fmt)
    align = ch * bits // 8
    frames = len(raw) // align
    ds64 = struct.pack("<QQQI", 36 + 28 + len(raw), len(raw), frames, 0)
    with open(path, "wb") as f:
        f.write(magic + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + b"ds64" + struct.pack("<I", len(ds64)) + ds64 +
                b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, rate, rate * align, align, bits) +
                b"data" + struct.pack("<I", 0xFFFFFFFF) + raw)


_W64_TAIL = bytes([0xf3, 0xac, 0xd3, 0x11, 0x8c, 0xd1, 0x00, 0xc0, 0x4f, 0x8e, 0xdb, 0x8a])


def write_w64(path, data, rate=44100, fmt="pcm16"):
    """Sony Wave64: GUID chunk ids, 64-bit chunk sizes that include the 24-byte chunk header, chunks padded to 8 bytes."""
    raw, tag, bits, ch = _wav_body(data, fmt)
    align = ch * bits // 8

    def chunk(cc, body):
        b = cc + _W64_TAIL + struct.pack("<Q", 24 + len(body)) + body
        return b + b"\0" * ((8 - len(b) % 8) % 8)
    fmtc = chunk(b"fmt ", struct.pack("<HHIIHH", tag, ch, rate, rate * align, align, bits))
    junk = chunk(b"junk", b"abcde")                             # an unknown chunk of odd size in front of the samples
    datac = chunk(b"data", raw)
    riff = b"riff" + bytes([0x2e, 0x91, 0xcf, 0x11, 0xa5, 0xd6, 0x28, 0xdb, 0x04, 0xc1, 0x00, 0x00])
    wave = b"wave" + _W64_TAIL
    total = 40 + len(fmtc) + len(junk) + len(datac)
    with open(path, "wb") as f:
        f.write(riff + struct.pack("<Q", total) + wave + fmtc + junk + datac)


def write_au(path, data, rate=44100, fmt="pcm16", open_ended=False):
    """Sun / NeXT .au, big-endian: pcm8 / pcm16 / pcm24 / pcm32 / float32 / float64 (encodings 2 .. 7)."""
    data = np.asarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    ch = data.shape[1]
    if fmt == "pcm8":
        raw, enc = np.clip(np.round(data * 128.0), -128, 127).astype(np.int8).tobytes(), 2
    elif fmt == "pcm16":
        raw, enc = np.clip(np.round(data * 32768.0), -32768, 32767).astype(">i2").tobytes(), 3
    elif fmt == "pcm24":
        v = np.clip(np.round(data * 8388608.0), -8388608, 8388607).astype(">i4")
        raw, enc = v.view(np.uint8).reshape(-1, 4)[:, 1:].tobytes(), 4
    elif fmt == "pcm32":
        raw, enc = np.clip(np.round(data * 2147483648.0), -2**31, 2**31 - 1).astype(">i4").tobytes(), 5
    elif fmt == "float32":
        raw, enc = data.astype(">f4").tobytes(), 6
    elif fmt == "float64":
        raw, enc = data.astype(">f8").tobytes(), 7
    elif fmt == "ulaw":
        raw, enc = bytes(len(data) * ch), 1                     # (content irrelevant: the reader must refuse it)
    else:
        raise ValueError(fmt)
    note = b"folve test\0\0"                                     # annotation: the data offset is past the 24-byte header
    with open(path, "wb") as f:
        f.write(b".snd" + struct.pack(">IIIII", 24 + len(note), 0xFFFFFFFF if open_ended else len(raw), enc, rate, ch) + note + raw)


def golden(name):
    return np.load(os.path.join(GOLDEN, "demo_%s.npz" % name))


def make_pass_filter_dir(tmp, name):
    """lowpass / highpass demo filter rebuilt from the golden taps: same WAV shape
    (65536-frame 16-bit stereo, zeros after the taps), same commands."""
    g = golden(name)
    d = os.path.join(str(tmp), name)
    os.makedirs(d, exist_ok=True)
    wav = np.zeros((int(g["wav_frames"]), 2), np.int16)
    taps = g["taps_int16"]
    wav[: len(taps)] = taps
    write_wav(os.path.join(d, "%s_44.wav" % name), wav, int(g["wav_rate"]))
    gain = {"lowpass": "0.75", "highpass": "0.55"}[name]
    with open(os.path.join(d, "filter-44100.conf"), "w") as f:
        f.write("#                 in  out     partition    maxsize\n")
        f.write("/convolver/new    2    2        1024        65536\n\n")
        f.write("/impulse/read    1   1  %s    0      0       0       1     %s_44.wav\n" % (gain, name))
        f.write("/impulse/read    2   2  %s    0      0       0       1     %s_44.wav\n#\n" % (gain, name))
    return d


def make_echo_filter_dir(tmp):
    g = golden("echo")
    d = os.path.join(str(tmp), "echo")
    os.makedirs(d, exist_ok=True)
    for rate, delay in ((44100, int(g["delay_44100"])), (192000, int(g["delay_192000"]))):
        with open(os.path.join(d, "filter-%d.conf" % rate), "w") as f:
            f.write("/convolver/new    2    2         256     204800        0.5\n")
            f.write("/impulse/dirac   1   1   0.7       0\n/impulse/dirac   2   2   0.7       0\n")
            f.write("/impulse/dirac   1   1   0.3       %d\n/impulse/dirac   2   2   0.3       %d\n" % (delay, delay))
    return d


def make_santalucia_shaped_dir(tmp, seed=2):
    """A synthetic IR with the shape of the SantaLucia demo (2 paths, 178193 taps at
    delay 500 from offset 1400 of a 179593-frame file, + dirac 0.4 @0, size 204800).
    Returns (dir, {(i,o): dense float32 h})."""
    g = golden("santalucia")
    frames, size, delay, offset = int(g["wav_frames"]), int(g["size"]), int(g["delay"]), int(g["offset"])
    rng = np.random.default_rng(seed)
    env = np.exp(-np.arange(frames) / 40000.0)
    ir = rng.standard_normal((frames, 2)) * env[:, None]
    ir = ir / np.abs(ir).max() * 0.9
    wav = np.round(ir * 32767).astype(np.int16)
    d = os.path.join(str(tmp), "SantaLuciaShaped")
    os.makedirs(d, exist_ok=True)
    write_wav(os.path.join(d, "ir.wav"), wav, 44100)
    with open(os.path.join(d, "filter-44100.conf"), "w") as f:
        f.write("/convolver/new    2    2         256     %d        0.5\n" % size)
        f.write("/impulse/read    1   1   4e-3     %d    %d       0    1   ir.wav\n" % (delay, offset))
        f.write("/impulse/read    2   2   4e-3     %d    %d       0    2   ir.wav\n" % (delay, offset))
        f.write("/impulse/dirac   1   1   0.4       0\n/impulse/dirac   2   2   0.4       0\n")
    hs = {}
    n = frames - offset
    for c in range(2):
        h = np.zeros(size, np.float32)
        h[delay:delay + n] += np.float32(4e-3) * (wav[offset:, c].astype(np.float32) / np.float32(32768.0))
        h[0] += np.float32(0.4)
        hs[(c, c)] = h
    return d, hs


def seeded_input(seed, frames, ch):
    return np.random.default_rng(seed).uniform(-1, 1, (frames, ch)).astype(np.float32)
