"""ctypes front-end of oracle/liboracle.so plus the float64 ground truth.

TEST INFRASTRUCTURE ONLY — see oracle/oracle_convproc.h for what is restated and
how it is pinned ("parity unpinned by reference tests": the reference has none).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAXSIZE = 0x100000


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    vp, ci, cf, cfp = C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_float)
    L.oc_convproc_new.restype = vp
    L.oc_convproc_delete.argtypes = [vp]
    L.oc_configure.argtypes = [vp, ci, ci, ci, ci, ci, ci, cf]
    L.oc_impdata_create.argtypes = [vp, ci, ci, ci, vp, ci, ci]
    L.oc_impdata_copy.argtypes = [vp, ci, ci, ci, ci]
    L.oc_inpdata.argtypes = [vp, ci]; L.oc_inpdata.restype = cfp
    L.oc_outdata.argtypes = [vp, ci]; L.oc_outdata.restype = cfp
    L.oc_process.argtypes = [vp]
    L.oc_reset.argtypes = [vp]
    L.oc_fragm.argtypes = [vp]
    L.oc_npar.argtypes = [vp]
    L.oc_path_partitions.argtypes = [vp, ci, ci]
    L.oc_sstring.argtypes = [C.c_char_p, C.c_char_p, ci]
    L.oc_fragm_for_size.argtypes = [C.c_uint]
    L.oc_sp_create.argtypes = [C.c_char_p, ci, ci]; L.oc_sp_create.restype = vp
    L.oc_sp_wrap.argtypes = [vp, ci, ci, ci]; L.oc_sp_wrap.restype = vp
    L.oc_sp_delete.argtypes = [vp]
    L.oc_sp_fill_buffer.argtypes = [vp, vp, ci]
    L.oc_sp_write_processed.argtypes = [vp, vp, ci]
    L.oc_sp_is_input_buffer_complete.argtypes = [vp]
    L.oc_sp_pending_writes.argtypes = [vp]
    L.oc_sp_reset.argtypes = [vp]
    L.oc_sp_max_output_value.argtypes = [vp]; L.oc_sp_max_output_value.restype = cf
    L.oc_sp_input_channels.argtypes = [vp]
    L.oc_sp_output_channels.argtypes = [vp]
    L.oc_sp_fragm.argtypes = [vp]
    L.oc_sp_convproc.argtypes = [vp]; L.oc_sp_convproc.restype = vp
    L.oc_sp_run.argtypes = [vp, vp, C.c_long, vp]; L.oc_sp_run.restype = C.c_long
    L.oc_bench_streams.argtypes = [ci, ci, ci, ci, ci, ci, C.c_uint]; L.oc_bench_streams.restype = C.c_double
    L.oc_wav_load.argtypes = [C.c_char_p, vp]
    L.fc_run.argtypes = [ci, ci, ci, vp, vp, C.c_long, vp]; L.fc_run.restype = ci
    L.fc_bench_streams.argtypes = [ci] * 6 + [C.c_uint]; L.fc_bench_streams.restype = C.c_double
    _LIB = L
    return L


def fragm_for_size(size):
    return lib().oc_fragm_for_size(int(size))


def sstring(src: bytes, size=1024):
    buf = C.create_string_buffer(max(size, 1) + 8)
    n = lib().oc_sstring(src, buf, size)
    return n, buf.value


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Convproc:
    """Restated zita Convproc (single level, quantum == minpart == maxpart)."""

    def __init__(self, ninp, nout, size, fragm=None, density=0.0):
        self.L = lib()
        self.h = self.L.oc_convproc_new()
        fragm = fragm or fragm_for_size(size)
        rc = self.L.oc_configure(self.h, ninp, nout, size, fragm, fragm, fragm, density)
        if rc:
            raise ValueError("oc_configure failed: %d" % rc)
        self.ninp, self.nout, self.size, self.fragm = ninp, nout, size, fragm
        self.npar = self.L.oc_npar(self.h)
        self._owned = True

    def impdata_create(self, inp, out, data, ind0, step=1):
        data = np.ascontiguousarray(data, dtype=np.float32)
        n = (len(data) + step - 1) // step
        return self.L.oc_impdata_create(self.h, inp, out, step, _fptr(data), ind0, ind0 + n)

    def impdata_copy(self, inp1, out1, inp2, out2):
        return self.L.oc_impdata_copy(self.h, inp1, out1, inp2, out2)

    def path_partitions(self, inp, out):
        return self.L.oc_path_partitions(self.h, inp, out)

    def reset(self):
        self.L.oc_reset(self.h)

    def process_block(self, x_planar):
        """x_planar: [ninp, fragm] float32 -> [nout, fragm]."""
        P = self.fragm
        for ch in range(self.ninp):
            buf = np.ctypeslib.as_array(self.L.oc_inpdata(self.h, ch), shape=(P,))
            buf[:] = x_planar[ch]
        self.L.oc_process(self.h)
        out = np.empty((self.nout, P), np.float32)
        for ch in range(self.nout):
            out[ch] = np.ctypeslib.as_array(self.L.oc_outdata(self.h, ch), shape=(P,))
        return out

    def release(self):
        self._owned = False
        return self.h

    def __del__(self):
        if getattr(self, "_owned", False) and self.h:
            self.L.oc_convproc_delete(self.h)
            self.h = None


class SoundProcessor:
    """Restated SoundProcessor (sound-processor.cc) over float spans."""

    def __init__(self, handle):
        self.L = lib()
        self.h = handle
        self.fragm = self.L.oc_sp_fragm(handle)
        self.ninp = self.L.oc_sp_input_channels(handle)
        self.nout = self.L.oc_sp_output_channels(handle)

    @classmethod
    def create(cls, config_file, samplerate, channels):
        h = lib().oc_sp_create(os.fsencode(config_file), samplerate, channels)
        return cls(h) if h else None

    @classmethod
    def wrap(cls, conv: "Convproc"):
        h = lib().oc_sp_wrap(conv.release(), conv.fragm, conv.ninp, conv.nout)
        return cls(h)

    def convproc_handle(self):
        return self.L.oc_sp_convproc(self.h)

    def path_partitions(self, inp, out):
        return self.L.oc_path_partitions(self.convproc_handle(), inp, out)

    def fill_buffer(self, src):
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, self.ninp)
        return self.L.oc_sp_fill_buffer(self.h, _fptr(src), src.shape[0])

    def write_processed(self, count):
        out = np.empty((count, self.nout), np.float32)
        self.L.oc_sp_write_processed(self.h, _fptr(out), count)
        return out

    def pending_writes(self):
        return self.L.oc_sp_pending_writes(self.h)

    def is_input_buffer_complete(self):
        return bool(self.L.oc_sp_is_input_buffer_complete(self.h))

    def max_output_value(self):
        return float(self.L.oc_sp_max_output_value(self.h))

    def reset(self):
        self.L.oc_sp_reset(self.h)

    def run(self, x):
        """x: [frames, ninp] float32 -> [frames, nout] (AddMoreSoundData loop)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.ninp)
        out = np.zeros((x.shape[0], self.nout), np.float32)
        n = self.L.oc_sp_run(self.h, _fptr(x), x.shape[0], _fptr(out))
        assert n == x.shape[0]
        return out

    def __del__(self):
        if self.h:
            self.L.oc_sp_delete(self.h)
            self.h = None


_NATIVE = None


def native_bench_lib():
    """liboracle_native.so: the same sources compiled -O3 -march=native on THIS machine (the box
    whose cores are being timed); None if it cannot be built.  Only bench.py's cpu_baseline uses it."""
    global _NATIVE
    if _NATIVE is None:
        so = os.path.join(_HERE, "liboracle_native.so")
        try:
            subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "native"], stdout=subprocess.DEVNULL)
            L = C.CDLL(so)
            L.oc_bench_streams.argtypes = [C.c_int] * 6 + [C.c_uint]
            L.oc_bench_streams.restype = C.c_double
            L.fc_bench_streams.argtypes = [C.c_int] * 6 + [C.c_uint]
            L.fc_bench_streams.restype = C.c_double
            _NATIVE = L
        except Exception:
            _NATIVE = False
    return _NATIVE or None


def bench_streams(nstreams, nblocks, nthreads, ninp=2, nout=2, size=262144, seed=3, native=False):
    L = (native_bench_lib() if native else None) or lib()
    return L.oc_bench_streams(nstreams, nblocks, nthreads, ninp, nout, size, seed)


def fast_bench_streams(nstreams, nblocks, nthreads, nch=2, size=262144, seed=3, native=False):
    """Seconds for the vectorised stand-in (oracle/fastcpu.c): same algorithm and shape as bench_streams."""
    L = (native_bench_lib() if native else None) or lib()
    return L.fc_bench_streams(nstreams, nblocks, nthreads, nch, size, fragm_for_size(size), seed)


def fast_run(x, taps, size=None):
    """x [frames (a multiple of the block), nch] through oracle/fastcpu.c with `taps` on every diagonal path."""
    x = np.ascontiguousarray(x, np.float32)
    taps = np.ascontiguousarray(taps, np.float32)
    size = size or len(taps)
    y = np.zeros_like(x)
    rc = lib().fc_run(x.shape[1], size, fragm_for_size(size), _fptr(taps), _fptr(x), x.shape[0], _fptr(y))
    assert rc == 0
    return y


# --------------------------------------------------------------------------
# float64 ground truth: exact causal linear convolution, truncated to len(x)
# ("the ground truth is the exact linear convolution evaluated
# in float64 of the float32 input with the float32-assembled h").
# --------------------------------------------------------------------------
def linear_convolution_f64(x, h_paths, nout):
    """x: [frames, ninp] float32; h_paths: {(inp, out): float32 taps}.
    Returns float64 [frames, nout]."""
    from scipy.signal import fftconvolve
    x = np.asarray(x)
    n = x.shape[0]
    y = np.zeros((n, nout), np.float64)
    for (i, o), h in h_paths.items():
        h = np.asarray(h, np.float64)
        if h.size == 0 or n == 0:
            continue
        nz = np.flatnonzero(h)
        if nz.size == 0:
            continue
        xi = x[:, i].astype(np.float64)
        if nz.size <= 64:     # sparse / short: direct sum is exact and cheap
            for t in nz:
                if t < n:
                    y[t:, o] += h[t] * xi[: n - t]
        else:
            y[:, o] += fftconvolve(xi, h[: nz[-1] + 1])[:n]
    return y


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0
