"""CPU: the product's host logic that needs no GPU — C-ABI surface, config loader, path choice."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import folve_amd as fa
from folve_amd import host as H
from fixtures import (REF_DEMO, golden, make_echo_filter_dir, make_pass_filter_dir, make_santalucia_shaped_dir,
                      write_wav)
from test_oracle_cpu import SSTRING_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = fa.lib()
    declared = set()
    for hdr in ("folve_engine.h", "folve_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(f[eh]_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 55
    from folve_amd.capi import ENGINE_SYMBOLS
    bound = {n for n, _, _ in ENGINE_SYMBOLS} | {n for n, _, _ in H.HOST_SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(L, name), name


def test_no_cpu_fallback_fails_loudly():
    if fa.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fa.FolveError) as ei:
        fa.Engine(0)
    assert ei.value.code == -4
    flt = fa.Filter(None, 1, 1, 100)
    flt.add(0, 0, [1.0])
    with pytest.raises(fa.FolveError):
        flt.commit()
    assert H.SoundProcessor.create(os.path.join(ROOT, "tests", "nonexistent.conf"), 44100, 2) is None


def test_product_never_links_the_oracle():
    so = fa.lib_path()
    out = os.popen("readelf -d %s" % so).read()
    assert "liboracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "folve_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hpp", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("import oracle", "from oracle", "liboracle", "oracle_", "oracle."):
                    assert needle not in txt, (f, needle)


@pytest.mark.parametrize("size", [1024, 8, 1, 0])
def test_host_sstring_equals_oracle(oracle, size):
    rng = np.random.default_rng(1)
    alphabet = b"ab \t'\"\\\n\x00x"
    cases = list(SSTRING_CASES) + [bytes(rng.choice(list(alphabet), rng.integers(1, 12))) for _ in range(2000)]
    for src in cases:
        assert H.sstring(src, size) == oracle.sstring(src, size), (src, size)


def test_filter_limits_and_block_size():
    assert fa.fragm_for_size(100) == 128 and fa.fragm_for_size(4097) == 8192 and fa.fragm_for_size(1) == 64
    for bad in [(0, 1, 100), (65, 1, 100), (1, 0, 100), (1, 65, 100), (1, 1, 0), (1, 1, 0x100001)]:
        with pytest.raises(fa.FolveError):
            fa.Filter(None, *bad)
    f = fa.Filter(None, 64, 64, 0x100000)
    assert f.block_size == 8192 and f.partitions == 128
    with pytest.raises(fa.FolveError):
        f.add(64, 0, [1.0])
    with pytest.raises(fa.FolveError):
        f.link(0, 0, 0, 0)
    f.add(0, 0, [1.0])
    f.link(0, 0, 1, 1)                  # (1,1) -> (0,0)
    f.link(1, 1, 2, 2)                  # (2,2) -> (1,1) -> (0,0)
    with pytest.raises(fa.FolveError):
        f.link(2, 2, 1, 1)              # (1,1) -> (2,2) -> (1,1): a cycle


def test_link_ordering_follows_zita(oracle):
    """Convproc::impdata_copy as zita implements it (zita-config.cc:274 is the call site): no effect when
    the source pair has no data yet or the target already has data; a linked target ignores data added
    to it; data added to the source afterwards is shared.  Engine and oracle restate the same rules."""
    f = fa.Filter(None, 2, 2, 1000)
    c = oracle.Convproc(2, 2, 1000)
    f.link(0, 0, 0, 1); c.impdata_copy(0, 0, 0, 1)      # copy BEFORE the source has data: nothing happens
    f.add(0, 0, [1.0, 2.0]); c.impdata_create(0, 0, np.float32([1.0, 2.0]), 0)
    assert f.path_partitions(0, 1) == 0 and c.path_partitions(0, 1) == 0
    f.add(1, 1, [5.0]); c.impdata_create(1, 1, np.float32([5.0]), 0)
    f.link(0, 0, 1, 1); c.impdata_copy(0, 0, 1, 1)      # target has data of its own: nothing happens
    assert f.taps(1, 1, 4).tolist() == [5.0, 0, 0, 0]
    f.link(0, 0, 1, 0); c.impdata_copy(0, 0, 1, 0)      # a real link
    f.add(1, 0, [9.0]); c.impdata_create(1, 0, np.float32([9.0]), 0)         # ignored: (1,0) is a link
    f.add(0, 0, [0.5], 2); c.impdata_create(0, 0, np.float32([0.5]), 2)      # seen through the link
    assert f.taps(1, 0, 4).tolist() == [1.0, 2.0, 0.5, 0] and f.taps(0, 0, 4).tolist() == [1.0, 2.0, 0.5, 0]
    assert c.path_partitions(1, 0) == 1 and f.path_partitions(1, 0) == 1
    # the oracle's spectra agree: an impulse through (1 -> 0) returns the shared taps
    sp = oracle.SoundProcessor.wrap(c)
    x = np.zeros((sp.fragm, 2), np.float32)
    x[0, 1] = 1.0
    y = sp.run(x)
    assert np.allclose(y[:4, 0], [1.0, 2.0, 0.5, 0], atol=1e-6) and np.allclose(y[:2, 1], [5.0, 0.0], atol=1e-6)


def test_accumulate_link_and_masks():
    f = fa.Filter(None, 2, 2, 30000)
    f.add(0, 0, np.ones(10, np.float32), 8190)          # straddles partitions 0 and 1
    f.add(0, 0, np.full(5, 2.0, np.float32), 8192)
    f.add(0, 0, [9.0], 29999)
    f.add(0, 0, [7.0, 7.0, 7.0], 32767)                 # beyond K*P = 32768: clipped
    f.link(0, 0, 1, 1)
    f.add(1, 1, [0.25], 0)                              # a linked pair takes no data of its own (zita): no effect
    f.add(0, 0, [0.5], 0)                               # a later addition to the SOURCE is shared (README.CONFIG.txt:91-97)
    t = f.taps(0, 0)
    assert t[8190] == 1 and t[8192] == 3 and t[8197] == 1 and t[29999] == 9 and t[32767] == 7 and t[0] == 0.5
    assert np.array_equal(f.taps(1, 1), t)
    assert f.path_partitions(0, 0) == 3 and f.path_partitions(1, 0) == 0   # partitions 0, 1 and 3
    # strided source (zita-config.cc:163 passes step = nchan)
    g = fa.Filter(None, 1, 1, 100)
    g.add(0, 0, np.arange(10, dtype=np.float32), 3, step=2)
    assert np.array_equal(g.taps(0, 0)[3:8], [0, 2, 4, 6, 8])


@pytest.mark.parametrize("name", ["lowpass", "highpass"])
def test_loader_assembles_pass_filters_bit_exactly(tmp_path, name):
    g = golden(name)
    d = make_pass_filter_dir(tmp_path, name)
    st, flt, z = H.config_load(os.path.join(d, "filter-44100.conf"))
    assert st == 0 and z == {"fragm": 8192, "ninp": 2, "nout": 2, "size": 65536}
    taps = g["taps_int16"]
    h = np.zeros(65536, np.float32)
    h[: len(taps)] = np.float32(g["gain"]) * (taps[:, 0].astype(np.float32) / np.float32(32768.0))
    for c in range(2):              # both outputs use file channel 1
        assert np.array_equal(flt.taps(c, c), h)
    assert flt.path_partitions(0, 0) == 8 and flt.path_partitions(0, 1) == 0   # zeros still populate partitions
    assert float(h.astype(np.float64).sum()) == pytest.approx(float(g["h_sum"]), abs=1e-9)


def test_loader_santalucia_semantics(tmp_path):
    d, hs = make_santalucia_shaped_dir(tmp_path)
    st, flt, z = H.config_load(os.path.join(d, "filter-44100.conf"))
    assert st == 0 and z["size"] == 204800 and flt.partitions == 25
    for c in range(2):
        assert np.array_equal(flt.taps(c, c, 204800), hs[(c, c)])      # delay 500, offset 1400, dirac accumulates
        assert flt.path_partitions(c, c) == 22


@pytest.mark.skipif(not os.path.isdir(REF_DEMO), reason="needs /root/reference")
def test_loader_on_the_reference_demo_filters():
    g = golden("santalucia")
    st, flt, z = H.config_load(os.path.join(REF_DEMO, "SantaLucia", "filter-44100.conf"))
    assert st == 0
    for c in range(2):
        h = flt.taps(c, c, 204800)
        assert np.array_equal(h[g["h_probe_idx"]], g["h_probe"][c])
        assert float(h.astype(np.float64).sum()) == pytest.approx(float(g["h_sum"][c]), rel=1e-12)
        assert float(np.linalg.norm(h.astype(np.float64))) == pytest.approx(float(g["h_l2"][c]), rel=1e-12)
        assert np.flatnonzero(h)[-1] <= int(g["last_tap"])
    assert flt.path_partitions(0, 0) == 22
    for name, k in (("echo", 2), ("lowpass", 8), ("highpass", 8)):
        st, flt, _ = H.config_load(os.path.join(REF_DEMO, name, "filter-44100.conf"))
        assert st == 0 and flt.path_partitions(0, 0) == k and flt.path_partitions(1, 0) == 0


def _conf(tmp_path, text, name="filter-44100.conf"):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(text)
    return p


def test_loader_grammar_errors_and_commands(tmp_path, oracle):
    new = "/convolver/new 2 2 256 20000\n"
    cases = [
        ("garbage line\n", H.ERR_SYNTAX),
        ("   # comment only\n\n" + new, 0),
        (new + "/bogus/command 1\n", H.ERR_COMMAND),
        ("/impulse/dirac 1 1 0.5 0\n", H.ERR_NOCONV),
        (new + "/impulse/dirac 3 1 0.5 0\n", H.ERR_IONUM),
        (new + "/impulse/dirac 1 1 0.5\n", H.ERR_PARAM),
        (new + "/impulse/copy 1 1 1 1\n", H.ERR_PARAM),
        (new + "/impulse/hilbert 1 1 1.0 100 32\n", H.ERR_PARAM),        # length < 64
        (new + "/impulse/read 1 1 1.0 0 0 0 1 missing.wav\n", 0),        # ERR_OTHER is swallowed
        ("/convolver/new 2 2 256\n", H.ERR_PARAM),
        ("/convolver/new 99 2 256 1000\n", 0),                           # out of range -> ERR_OTHER -> 0, no filter
        (new + "/input/name 1 in.L\n/output/name 2 out.R\n", 0),
        (new + "/cd\n", H.ERR_PARAM),
    ]
    for i, (text, want) in enumerate(cases):
        p = _conf(tmp_path, text, "c%d.conf" % i)
        st, flt, _ = H.config_load(p)
        assert st == want, (text, st)
        # the oracle restatement agrees on every status
        sp_ok = oracle.SoundProcessor.create(p, 44100, 2) is not None
        assert sp_ok == (st == 0 and flt is not None), text
    st, flt, _ = H.config_load(os.path.join(str(tmp_path), "does-not-exist.conf"))
    assert st == -1 and flt is None


def test_loader_hilbert_copy_cd_quoting_and_formats(tmp_path, oracle):
    sub = os.path.join(str(tmp_path), "ir dir")
    os.makedirs(sub)
    rng = np.random.default_rng(4)
    ir = rng.uniform(-0.5, 0.5, (3000, 3))
    write_wav(os.path.join(sub, "f32.wav"), ir, 48000, "float32")
    write_wav(os.path.join(sub, "p24.wav"), ir, 44100, "pcm24")
    write_wav(os.path.join(sub, "p32.wav"), ir, 44100, "pcm32")
    write_wav(os.path.join(sub, "p8.wav"), ir, 44100, "pcm8")
    text = ("/convolver/new 2 3 512 16000 0.3\n"
            "/cd \"ir dir\"\n"
            "/impulse/read 1 1 0.5 10 100 1000 3 f32.wav\n"          # rate mismatch is only logged
            "/impulse/read 1 1 0.25 20 0 0 1 'p24.wav'\n"             # accumulates onto the same pair
            "/impulse/read 2 2 1.0 0 2990 500 2 p32.wav\n"            # length beyond EOF: ends at EOF
            "/impulse/read 2 3 1.0 15990 0 0 1 p8.wav\n"              # truncated to size - delay
            "/impulse/hilbert 2 1 0.8 300 256\n"
            "/impulse/hilbert 2 1 0.8 100 256\n"                      # delay < length/2: skipped
            "/impulse/dirac 2 1 0.1 16000\n"                          # delay >= size: ignored
            "/impulse/copy 1 3 1 1\n")
    p = _conf(tmp_path, text)
    st, flt, z = H.config_load(p)
    assert st == 0 and z["ninp"] == 2 and z["nout"] == 3 and flt.block_size == 8192
    f32 = ir.astype(np.float32)
    h11 = np.zeros(16000, np.float32)
    h11[10:1010] += f32[100:1100, 2] * np.float32(0.5)
    p24 = (np.clip(np.round(ir * 8388608.0), -8388608, 8388607).astype(np.int32) * 256).astype(np.float32) / np.float32(2147483648.0)
    h11[20:3020] += p24[:, 0] * np.float32(0.25)
    assert np.array_equal(flt.taps(0, 0, 16000), h11)
    assert np.array_equal(flt.taps(0, 2, 16000), h11)                  # copy shares the data
    p32 = np.clip(np.round(ir * 2147483648.0), -2**31, 2**31 - 1).astype(np.int32).astype(np.float32) / np.float32(2147483648.0)
    h22 = np.zeros(16000, np.float32); h22[0:10] = p32[2990:3000, 1]
    assert np.array_equal(flt.taps(1, 1, 16000), h22)
    p8 = (np.clip(np.round(ir * 128.0) + 128, 0, 255).astype(np.int32) - 128).astype(np.float32) / np.float32(128.0)
    h23 = np.zeros(16000, np.float32); h23[15990:16000] = p8[:10, 0]
    assert np.array_equal(flt.taps(1, 2, 16000), h23)
    hil = flt.taps(1, 0, 16000)
    nz = np.flatnonzero(hil)
    assert nz.min() >= 300 - 128 and nz.max() < 300 + 128 and hil[300] == 0
    assert np.allclose(hil[300 + 1], -hil[300 - 1]) and hil[300 - 1] > 0      # antisymmetric about the delay
    # the oracle's loader builds the same convolver: compare impulse responses
    sp = oracle.SoundProcessor.create(p, 44100, 2)
    x = np.zeros((2 * 8192, 2), np.float32); x[0, 0] = 1.0
    y = sp.run(x)
    assert np.abs(y[:16000, 0] - h11).max() < 1e-6 and np.abs(y[:16000, 2] - h11).max() < 1e-6
    x = np.zeros((2 * 8192, 2), np.float32); x[0, 1] = 1.0
    y = sp.run(x)
    assert np.abs(y[:16000, 0] - hil).max() < 1e-6 and np.abs(y[:16000, 1] - h22).max() < 1e-6
    assert np.abs(y[:16000, 2] - h23).max() < 1e-6


def test_loader_reads_rf64_w64_and_au_impulse_files(tmp_path):
    """zita-audiofile.cc:51-99 takes whatever sf_open opens.  Beside WAVE / AIFF / CAF the loader reads the other
    uncompressed containers itself — RF64 and BW64 (64-bit sizes in 'ds64'), Sony Wave64 (GUID chunks, 8-byte padding), Sun
    .au (PCM 8 / 16 / 24 / 32, float, double) — normalised as sf_readf_float normalises; a companded .au is refused here
    (it belongs to the fallback opener: tests/test_adapter_gpu.py)."""
    from fixtures import write_au, write_rf64, write_w64
    rng = np.random.default_rng(19)
    ir = rng.uniform(-0.9, 0.9, (611, 3))
    d = str(tmp_path)
    files = []
    for fmt, magic in (("pcm16", b"RF64"), ("pcm24", b"RF64"), ("float32", b"BW64")):
        name = "r_%s_%s.wav" % (fmt, magic.decode())
        write_rf64(os.path.join(d, name), ir, 48000, fmt, magic)
        files.append((name, fmt))
    for fmt in ("pcm16", "pcm24", "float32"):
        write_w64(os.path.join(d, "w_%s.w64" % fmt), ir, 96000, fmt)
        files.append(("w_%s.w64" % fmt, fmt))
    for fmt in ("pcm8", "pcm16", "pcm24", "pcm32", "float32", "float64"):
        write_au(os.path.join(d, "u_%s.au" % fmt), ir, 44100, fmt, open_ended=(fmt == "pcm24"))
        files.append(("u_%s.au" % fmt, fmt))
    text = "/convolver/new 1 %d 512 900\n" % len(files)
    for k, (name, _) in enumerate(files):
        text += "/impulse/read 1 %d 1.0 %d 7 0 3 %s\n" % (k + 1, k, name)      # offset 7, third channel, delay k
    st, flt, z = H.config_load(_conf(tmp_path, text))
    assert st == 0 and z["nout"] == len(files)

    def q(scale, lo, hi):
        return np.clip(np.round(ir * scale), lo, hi).astype(np.int64)
    expect = {
        "pcm8": q(128.0, -128, 127).astype(np.float32) / np.float32(128.0),
        "pcm16": q(32768.0, -32768, 32767).astype(np.float32) / np.float32(32768.0),
        "pcm24": (q(8388608.0, -8388608, 8388607) * 256).astype(np.float32) / np.float32(2147483648.0),
        "pcm32": q(2147483648.0, -2**31, 2**31 - 1).astype(np.float32) / np.float32(2147483648.0),
        "float32": ir.astype(np.float32), "float64": ir.astype(np.float32),
    }
    for k, (name, fmt) in enumerate(files):
        h = np.zeros(900, np.float32)
        h[k:k + 604] = expect[fmt][7:, 2]
        assert np.array_equal(flt.taps(0, k, 900), h), name
    # refused: u-law .au (an encoding for the fallback opener), an RF64 without its ds64, a Wave64 cut inside its header
    write_au(os.path.join(d, "ulaw.au"), ir, 8000, "ulaw")
    open(os.path.join(d, "nods64.wav"), "wb").write(b"RF64\xff\xff\xff\xffWAVEfmt \x10\0\0\0\1\0\1\0\x44\xac\0\0\x88\x58\1\0\2\0\x10\0data\xff\xff\xff\xff" + b"\1\0" * 10)
    blob = open(os.path.join(d, "w_pcm16.w64"), "rb").read()
    open(os.path.join(d, "cut.w64"), "wb").write(blob[:50])
    for bad in ("ulaw.au", "cut.w64"):
        st, flt, _ = H.config_load(_conf(tmp_path, "/convolver/new 1 1 512 900\n/impulse/read 1 1 1 0 0 0 1 %s\n" % bad))
        assert st != 0 or flt.path_partitions(0, 0) == 0, bad
    # (an RF64 whose data size is unknown is read to the end of the file, as libsndfile does)
    st, flt, _ = H.config_load(_conf(tmp_path, "/convolver/new 1 1 512 900\n/impulse/read 1 1 1 0 0 0 1 nods64.wav\n"))
    assert st == 0 and np.array_equal(flt.taps(0, 0, 12)[:10], np.full(10, 1.0 / 32768.0, np.float32))


def test_impulse_files_the_reader_refuses_go_to_the_libsndfile_fallback(tmp_path):
    """/root/reference/zita-audiofile.cc:51-99: an impulse file is whatever sf_open opens (FLAC, Ogg ..).  Where libsndfile
    exists (a folve build) host/sndfile_adapter.cpp registers sf_open / sf_seek / sf_readf_float / sf_close as the loader's
    fallback decoder.  tests/compile/impulse_fallback.cpp compiles the adapter against test-supplied sf_* (a made-up
    container the in-house reader refuses), links the library and loads a configuration through the real loader: offset,
    length, channel, gain and delay work as for a WAVE file; a WAVE file beside it is still read in-house; a file neither
    side can open is skipped as the reference skips it (ERR_OTHER, zita-config.cc:345)."""
    import json
    import shutil
    import struct
    import subprocess
    from fixtures import write_wav
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    import folve_amd as fa
    libdir = os.path.dirname(fa.lib_path())
    exe = os.path.join(str(tmp_path), "impulse_fallback")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "compile", "impulse_fallback.cpp"),
                        "-o", exe, fa.lib_path(), "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"],
                       capture_output=True, text=True)   # (by path: FOLVE_AMD_LIB may name another build — the sanitizer one of tools/asan_host.sh)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(23)
    ir = rng.uniform(-1, 1, (300, 2)).astype(np.float32)
    with open(os.path.join(str(tmp_path), "ir.toy"), "wb") as f:
        f.write(b"TOY1" + struct.pack("<iii", 44100, 2, 300) + ir.astype("<f4").tobytes())
    write_wav(os.path.join(str(tmp_path), "ir.wav"), np.round(ir * 32767).astype(np.int16), 44100)
    open(os.path.join(str(tmp_path), "junk.bin"), "wb").write(b"neither a sound file nor a toy" * 4)
    conf = _conf(tmp_path, "/convolver/new 1 4 64 400\n"
                           "/impulse/read 1 1 0.5 3 10 100 2 ir.toy\n"       # gain, delay 3, offset 10, length 100, channel 2
                           "/impulse/read 1 2 1.0 0 0 0 1 ir.toy\n"          # the whole file, channel 1
                           "/impulse/read 1 3 1.0 0 0 0 1 ir.wav\n"          # in-house reader, no fallback involved
                           "/impulse/read 1 4 1.0 0 0 0 1 junk.bin\n")       # nobody reads this: skipped
    r = subprocess.run([exe, conf, "400"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["status"] == 0 and out["nout"] == 4
    assert out["opens"] == 2 and out["closes"] == 2 and out["seeks"] == 1   # two .toy reads (one with an offset); the WAVE never got there
    taps = np.array(out["taps"], np.float32)
    h0 = np.zeros(400, np.float32)
    h0[3:103] = np.float32(0.5) * ir[10:110, 1]
    h1 = np.zeros(400, np.float32)
    h1[:300] = ir[:, 0]
    h2 = np.zeros(400, np.float32)
    h2[:300] = np.round(ir[:, 0] * 32767).astype(np.int16).astype(np.float32) / np.float32(32768.0)
    assert np.array_equal(taps[0], h0) and np.array_equal(taps[1], h1) and np.array_equal(taps[2], h2)
    assert not taps[3].any()


def test_loader_reads_aiff_and_caf_impulse_files(tmp_path):
    """zita-audiofile.cc:63-75 takes whatever libsndfile opens and names CAF and WAVEX; the loader
    reads the uncompressed AIFF / AIFF-C / CAF forms itself, normalised as sf_readf_float does."""
    from fixtures import write_aiff, write_caf
    rng = np.random.default_rng(9)
    ir = rng.uniform(-0.9, 0.9, (777, 2))
    d = str(tmp_path)
    files = []
    for fmt in ("pcm8", "pcm16", "pcm24", "pcm32", "sowt16", "fl32"):
        write_aiff(os.path.join(d, "a_%s.aif" % fmt), ir, 48000, fmt)
        files.append(("a_%s.aif" % fmt, fmt))
    for fmt in ("f32le", "f32be", "i16be", "i24le", "i8"):
        write_caf(os.path.join(d, "c_%s.caf" % fmt), ir, 44100, fmt, open_ended=(fmt == "i16be"))
        files.append(("c_%s.caf" % fmt, fmt))
    text = "/convolver/new 1 %d 512 1000\n" % len(files)
    for k, (name, _) in enumerate(files):
        text += "/impulse/read 1 %d 1.0 %d 5 0 2 %s\n" % (k + 1, k, name)      # offset 5, second channel, delay k
    st, flt, z = H.config_load(_conf(tmp_path, text))
    assert st == 0 and z["nout"] == len(files)

    def q(scale, lo, hi):
        return np.clip(np.round(ir * scale), lo, hi).astype(np.int64)
    expect = {
        "pcm8": q(128.0, -128, 127).astype(np.float32) / np.float32(128.0),
        "i8": q(128.0, -128, 127).astype(np.float32) / np.float32(128.0),
        "pcm16": q(32768.0, -32768, 32767).astype(np.float32) / np.float32(32768.0),
        "sowt16": q(32768.0, -32768, 32767).astype(np.float32) / np.float32(32768.0),
        "i16be": q(32768.0, -32768, 32767).astype(np.float32) / np.float32(32768.0),
        "pcm24": (q(8388608.0, -8388608, 8388607) * 256).astype(np.float32) / np.float32(2147483648.0),
        "i24le": (q(8388608.0, -8388608, 8388607) * 256).astype(np.float32) / np.float32(2147483648.0),
        "pcm32": q(2147483648.0, -2**31, 2**31 - 1).astype(np.float32) / np.float32(2147483648.0),
        "fl32": ir.astype(np.float32), "f32le": ir.astype(np.float32), "f32be": ir.astype(np.float32),
    }
    for k, (name, fmt) in enumerate(files):
        h = np.zeros(1000, np.float32)
        h[k:k + 772] = expect[fmt][5:, 1]
        assert np.array_equal(flt.taps(0, k, 1000), h), name
    # a compressed AIFF-C and a truncated header are refused (ERR_OTHER from readfile is swallowed: no path)
    open(os.path.join(d, "bad.aifc"), "wb").write(b"FORM\0\0\0\x1eAIFCCOMM\0\0\0\x16" + b"\0\1\0\0\0\1\0\x10"
                                                   + b"\x40\x0e\xac\x44\0\0\0\0\0\0" + b"ima4")
    open(os.path.join(d, "short.caf"), "wb").write(b"caff\0\1\0\0desc")
    for bad in ("bad.aifc", "short.caf"):
        st, flt, _ = H.config_load(_conf(tmp_path, "/convolver/new 1 1 512 1000\n/impulse/read 1 1 1 0 0 0 1 %s\n" % bad))
        assert st != 0 or flt.path_partitions(0, 0) == 0, bad


def test_pool_path_choice_and_error_strings(tmp_path):
    d = make_echo_filter_dir(tmp_path)
    pool = H.ProcessorPool(3)
    p, err = pool.get_or_create(d, 48000, 2, 16)
    assert p is None and err == "No filter in echo for 48.0kHz/2 ch/16 bits"
    p, err = pool.get_or_create(d, 88200, 6, 24)
    assert p is None and err == "No filter in echo for 88.2kHz/6 ch/24 bits"
    if fa.device_count() == 0:
        # most specific existing file wins: -R-C-B, then -R-C, then -R (processor-pool.cc:53-61)
        open(os.path.join(d, "filter-44100-2.conf"), "w").write("garbage\n")
        p, err = pool.get_or_create(d, 44100, 2, 16)
        assert p is None and err == "Problem parsing " + os.path.join(d, "filter-44100-2.conf")
        open(os.path.join(d, "filter-44100-2-16.conf"), "w").write("garbage\n")
        p, err = pool.get_or_create(d, 44100, 2, 16)
        assert p is None and err == "Problem parsing " + os.path.join(d, "filter-44100-2-16.conf")
        p, err = pool.get_or_create(d, 44100, 2, 24)
        assert err == "Problem parsing " + os.path.join(d, "filter-44100-2.conf")
    pool.give_back(None)        # Return(NULL) is a no-op


def test_combiner_serves_every_caller_under_contention():
    """folve::BatchScheduler without a GPU: the engine refuses every block (null stream), which is all the
    combiner's own logic needs — 16 threads x 300 blocks, every caller gets its block's status back, batches
    form under contention, a refused batch is retried block by block, nothing deadlocks."""
    import threading
    L = H._L()
    before = H.batching_stats()
    results, nthreads, ncalls = [], 16, 300
    buf = np.zeros(16, np.float32)

    def work():
        rcs = [L.fh_batcher_process(None, None, buf.ctypes.data, 8, buf.ctypes.data) for _ in range(ncalls)]
        results.append(rcs)

    th = [threading.Thread(target=work) for _ in range(nthreads)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th)
    assert len(results) == nthreads and all(rc == -2 for rcs in results for rc in rcs)      # FE_ERR_PARAM for each block


def test_combiner_ticket_path_under_thread_sanitizer(tmp_path):
    """folve::BatchScheduler's success path — tickets, waiters, per-request wake-ups, two batches in flight, the
    block-by-block retry of a refused submit — against a fake engine (tests/compile/combiner_stress.cpp), 24 threads mixing
    the synchronous and the run-ahead pattern, built with -fsanitize=thread: every request computed exactly once, no data
    race reported, no deadlock.  (With a GPU the same code is under tests/test_run_ahead_gpu.py and the harness' verify mode.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = os.path.join(str(tmp_path), "combiner_stress")
    src = os.path.join(ROOT, "tests", "compile", "combiner_stress.cpp")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", src, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("g++ without ThreadSanitizer")
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "24", "300"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["bad"] == 0 and out["done"] == 24 * 300 == out["requests"]
    assert out["batches"] < out["requests"] and out["refused"] > 0 and out["max_tickets_in_flight"] <= 2


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_layer_on_a_fake_engine_under_sanitizers(tmp_path, sanitizer):
    """The whole host layer without a GPU: the real SoundProcessor (run-ahead ring, tail, ramp, device or scanned peaks),
    ProcessorPool, DeviceRouter, combiner and jconvolver loader linked against a fake engine that computes a direct-form FIR
    (tests/compile/host_on_fake_engine.cpp).  8 file threads x 12 rounds of random-length files at run-ahead depths 1 .. 64,
    half of them handed over gaplessly to a second file (convolve-file-handler.cc:328-351,370-424): every FillBuffer returns
    what the reference's returns, pending_writes / is_input_buffer_complete agree, the output equals the convolution of the
    concatenation, max_output_value the maximum of what was written — under ThreadSanitizer and AddressSanitizer + UBSan."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    host = os.path.join(ROOT, "folve_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tests", "compile", "host_on_fake_engine.cpp")] + [
        os.path.join(host, n) for n in ("sound_processor.cpp", "processor_pool.cpp", "device_router.cpp", "batch_scheduler.cpp",
                                        "numa_placement.cpp", "zita_config.cpp", "impulse_file.cpp", "sstring.cpp")] + [
        os.path.join(ROOT, "folve_amd", "csrc", "trace.cpp")]
    exe = os.path.join(str(tmp_path), "host_fake")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-pthread", "-I" + os.path.join(ROOT, "include")] + srcs +
                       ["-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("g++ without this sanitizer")
    assert r.returncode == 0, r.stderr[-3000:]
    work = os.path.join(str(tmp_path), "filters")
    os.makedirs(work)
    r = subprocess.run([exe, work, "8", "12"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {"files": 96, "bad": 0}
    # the GPU sharder's failure handling on the same binary: eight slots, one never comes up / fails its calls / hangs its
    # probe — no open fails while any GPU works, the sick slot is fenced and gets no new files, the pool hands out nothing
    # that lives there, a probe brings it back (device_router.h; the reference: processor-pool.cc:71-77, folve-filesystem.cc:78-88)
    r = subprocess.run([exe, work, "router"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"router_scenario": "ok", "failed_checks": 0}
    # an open file survives its GPU (sound_processor.h): GPUs dying under files in mid-conversion — the files move to other
    # GPUs from their kept input and come out as if nothing had happened; silence only when no GPU is left
    # (the reference never drops a block, sound-processor.cc:98-127; state moves between owners, convolve-file-handler.cc:328-351)
    trace = os.path.join(str(tmp_path), "events.txt")
    r = subprocess.run([exe, work, "survive"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
                                FOLVE_AMD_TRACE=trace))
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"survive_scenario": "ok", "failed_checks": 0}
    # ... and the host-layer event log of that run (FOLVE_AMD_TRACE, csrc/trace.h: the counterpart of folve's -D,
    # /root/reference/folve-main.cc:63-97): "<us> <tid> <event> ...", the moves among the chunk submits and settles
    ev = [l.split(None, 3) for l in open(trace).read().splitlines()]
    assert ev and all(len(e) >= 3 and e[0].isdigit() and e[1].isdigit() for e in ev)
    kinds = {e[2] for e in ev}
    assert {"submit", "settle", "move"} <= kinds, kinds
    assert sum(e[2] == "move" for e in ev) >= 7                    # rounds 1 - 4: 1 + 1 + 2 + 3 GPUs died, each under one or two files
    assert len({e[1] for e in ev}) >= 8                             # eight file threads
