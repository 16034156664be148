"""ctypes binding of include/folve_engine.h (the C ABI of the gfx950 engine)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FE_HOST_PTRS, FE_DEVICE_PTRS, FE_ASYNC = 0, 1, 2
FE_K_FORWARD, FE_K_MAC, FE_K_INVERSE, FE_K_COUNT = 0, 1, 2, 3
FE_TUNE_FWD_RUN, FE_TUNE_INV_RUN, FE_TUNE_MAC_FORM, FE_TUNE_FFT_FORM, FE_TUNE_FAIL_NEXT, FE_TUNE_LANES = 0, 1, 2, 3, 4, 5
FE_TUNE_WALK_LPB, FE_TUNE_WALK_TILES, FE_TUNE_DUPLEX_OUT, FE_TUNE_DUPLEX_CHUNK_MB, FE_TUNE_DUPLEX_MIN_MB = 6, 7, 8, 9, 10
FE_TUNE_SPLIT, FE_TUNE_DUPLEX_CAP_MB, FE_TUNE_WALK_FMA, FE_TUNE_WALK_NT = 11, 12, 13, 14
TUNE_KNOBS = {"fwd_run": FE_TUNE_FWD_RUN, "inv_run": FE_TUNE_INV_RUN, "mac_form": FE_TUNE_MAC_FORM,
              "fft_form": FE_TUNE_FFT_FORM, "fail_next": FE_TUNE_FAIL_NEXT, "lanes": FE_TUNE_LANES,
              "walk_lpb": FE_TUNE_WALK_LPB, "walk_tiles": FE_TUNE_WALK_TILES, "duplex_out": FE_TUNE_DUPLEX_OUT, "duplex_chunk_mb": FE_TUNE_DUPLEX_CHUNK_MB, "duplex_min_mb": FE_TUNE_DUPLEX_MIN_MB, "split": FE_TUNE_SPLIT, "duplex_cap_mb": FE_TUNE_DUPLEX_CAP_MB,
              "walk_fma": FE_TUNE_WALK_FMA, "walk_nt": FE_TUNE_WALK_NT}
KERNEL_NAMES = ("forward", "mac", "inverse")


class FolveError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with %d: %s" % (what, code, lib().fe_last_error().decode(errors="replace")))
        self.code = code


def lib_path():
    # FOLVE_AMD_LIB: load another build of the same library (the sanitizer build of tools/asan_host.sh)
    return os.environ.get("FOLVE_AMD_LIB") or os.path.join(_HERE, "libfolve_amd.so")


def build_library(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])
    return lib_path()


# every symbol include/folve_engine.h declares: (name, restype, argtypes)
_vp, _i, _ll, _f = C.c_void_p, C.c_int, C.c_longlong, C.c_float
_pvp = C.POINTER(C.c_void_p)
ENGINE_SYMBOLS = [
    ("fe_device_count", _i, []),
    ("fe_engine_create", _i, [_i, _vp, _pvp]),
    ("fe_engine_destroy", None, [_vp]),
    ("fe_engine_synchronize", _i, [_vp]),
    ("fe_engine_device", _i, [_vp]),
    ("fe_engine_probe", _i, [_vp]),
    ("fe_last_error", C.c_char_p, []),
    ("fe_fragm_for_size", _i, [C.c_uint]),
    ("fe_filter_create", _i, [_vp, _i, _i, _i, _f, _pvp]),
    ("fe_filter_add", _i, [_vp, _i, _i, _i, _vp, _i, _i]),
    ("fe_filter_link", _i, [_vp, _i, _i, _i, _i]),
    ("fe_filter_commit", _i, [_vp]),
    ("fe_filter_retain", None, [_vp]),
    ("fe_filter_release", None, [_vp]),
    ("fe_filter_inputs", _i, [_vp]),
    ("fe_filter_outputs", _i, [_vp]),
    ("fe_filter_block_size", _i, [_vp]),
    ("fe_filter_partitions", _i, [_vp]),
    ("fe_filter_maxsize", _i, [_vp]),
    ("fe_filter_path_partitions", _i, [_vp, _i, _i]),
    ("fe_filter_get_taps", _i, [_vp, _i, _i, _vp, _i]),
    ("fe_stream_open", _i, [_vp, _i, _pvp]),
    ("fe_host_alloc", _i, [C.c_size_t, _pvp]),
    ("fe_host_free", None, [_vp]),
    ("fe_stream_bind_host_buffer", _i, [_vp, _vp, C.c_size_t]),
    ("fe_stream_reset", _i, [_vp]),
    ("fe_stream_close", None, [_vp]),
    ("fe_stream_process", _i, [_vp, _vp, _i, _vp, C.POINTER(_f), C.POINTER(_f)]),
    ("fe_stream_process_blocks", _i, [_vp, _vp, _ll, _vp]),
    ("fe_stream_get_peaks", _i, [_vp, C.POINTER(_f), C.POINTER(_f)]),
    ("fe_stream_reset_peaks", _i, [_vp]),
    ("fe_stream_blocks_done", _ll, [_vp]),
    ("fe_stream_block_size", _i, [_vp]),
    ("fe_stream_max_blocks", _i, [_vp]),
    ("fe_batch_process", _i, [_pvp, _i, _pvp, C.POINTER(_ll), _pvp, _i]),
    ("fe_batch_submit", _i, [_pvp, _i, _pvp, C.POINTER(_ll), _pvp, _pvp]),
    ("fe_ticket_wait", _i, [_vp]),
    ("fe_ticket_done", _i, [_vp]),
    ("fe_batch_submit_peaks", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    ("fe_filter_use_count", _i, [_vp]),
    ("fe_device_local_cpulist", _i, [_i, C.c_char_p, C.c_size_t]),
    ("fe_batch_get_peaks", _i, [_pvp, _i, C.POINTER(_f), C.POINTER(_f)]),
    ("fe_engine_set_tuning", _i, [_vp, _i, _i]),
    ("fe_debug_xlane", _i, [_vp, _vp]),
    ("fe_engine_set_profiling", _i, [_vp, _i]),
    ("fe_engine_get_profile", _i, [_vp, C.POINTER(_ll), C.POINTER(C.c_double)]),
    ("fe_engine_get_kernel_profile", _i, [_vp, C.POINTER(_ll), C.POINTER(C.c_double)]),
    ("fe_engine_reset_profile", _i, [_vp]),
    ("fe_engine_last_kernels", _i, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("fe_engine_hbm_rates", _i, [_vp, C.c_size_t, _i, C.POINTER(C.c_double)]),
    ("fe_engine_hbm_rates2", _i, [_vp, C.c_size_t, _i, C.POINTER(C.c_double)]),
    ("fe_engine_hbm_rates3", _i, [_vp, C.c_size_t, _i, C.POINTER(C.c_double)]),
]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "folve_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C folve_amd/csrc`. There is no CPU fallback." % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7;
    # importing it first lets the dynamic linker satisfy our DT_NEEDED entry with
    # that copy, so torch tensors' device pointers are valid in the engine.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    for name, res, args in ENGINE_SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    from . import host as _host
    _host.declare(L)
    _LIB = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise FolveError(rc, what)


def fragm_for_size(size):
    return lib().fe_fragm_for_size(int(size))


def device_count():
    return lib().fe_device_count()


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _host_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One GPU: HIP stream, twiddle tables, scratch."""

    def __init__(self, device=0, hip_stream=None):
        h = C.c_void_p()
        _chk(lib().fe_engine_create(int(device), C.c_void_p(hip_stream or 0), C.byref(h)), "fe_engine_create")
        self.h = h
        self.device = device

    def synchronize(self):
        _chk(lib().fe_engine_synchronize(self.h), "fe_engine_synchronize")

    def set_tuning(self, **knobs):
        """Pin kernel forms on this engine (fwd_run, inv_run, mac_form, fft_form, fail_next); 0 = automatic."""
        for name, value in knobs.items():
            _chk(lib().fe_engine_set_tuning(self.h, TUNE_KNOBS[name], int(value)), "fe_engine_set_tuning")

    def xlane_selftest(self):
        out = np.zeros(512, np.float32)
        _chk(lib().fe_debug_xlane(self.h, _host_ptr(out)), "fe_debug_xlane")
        return out

    def set_profiling(self, on):
        _chk(lib().fe_engine_set_profiling(self.h, int(on)), "fe_engine_set_profiling")

    def reset_profile(self):
        _chk(lib().fe_engine_reset_profile(self.h), "fe_engine_reset_profile")

    def get_profile(self):
        n = (C.c_longlong * FE_K_COUNT)()
        ms = (C.c_double * FE_K_COUNT)()
        _chk(lib().fe_engine_get_profile(self.h, n, ms), "fe_engine_get_profile")
        return {KERNEL_NAMES[k]: {"launches": int(n[k]), "ms": float(ms[k])} for k in range(FE_K_COUNT)}

    def get_kernel_profile(self):
        """Profiling mode 2 (set_profiling(2)): per role the dispatches timed and the sum of their own begin-to-end times."""
        n = (C.c_longlong * FE_K_COUNT)()
        ms = (C.c_double * FE_K_COUNT)()
        _chk(lib().fe_engine_get_kernel_profile(self.h, n, ms), "fe_engine_get_kernel_profile")
        return {KERNEL_NAMES[k]: {"launches": int(n[k]), "ms": float(ms[k])} for k in range(FE_K_COUNT)}

    def last_kernels(self):
        """{"forward", "mac", "inverse"}: the kernels of this engine's most recent launch round, as rocprofv3 names them."""
        bufs = [C.create_string_buffer(96) for _ in range(FE_K_COUNT)]
        _chk(lib().fe_engine_last_kernels(self.h, bufs[0], bufs[1], bufs[2], 96), "fe_engine_last_kernels")
        return {KERNEL_NAMES[k]: bufs[k].value.decode() for k in range(FE_K_COUNT)}

    def hbm_rates(self, nbytes=1 << 31, reps=20):
        """GB/s this GPU's HBM gives plain streaming kernels: {"read", "write", "copy"} (copy counts both ways)."""
        g = (C.c_double * 3)()
        _chk(lib().fe_engine_hbm_rates(self.h, nbytes, reps, g), "fe_engine_hbm_rates")
        return {"read": float(g[0]), "write": float(g[1]), "copy": float(g[2])}

    def hbm_rates2(self, nbytes=1 << 31, reps=20):
        """hbm_rates plus "write_regions" / "copy_regions": every workgroup storing into its own contiguous region."""
        g = (C.c_double * 5)()
        _chk(lib().fe_engine_hbm_rates2(self.h, nbytes, reps, g), "fe_engine_hbm_rates2")
        return {"read": float(g[0]), "write": float(g[1]), "copy": float(g[2]), "write_regions": float(g[3]), "copy_regions": float(g[4])}

    def hbm_rates3(self, nbytes=1 << 31, reps=20):
        """hbm_rates2 plus "copy_best": the best float4 copy shape found on MI355X (non-temporal both ways, 4 KiB bursts)."""
        g = (C.c_double * 6)()
        _chk(lib().fe_engine_hbm_rates3(self.h, nbytes, reps, g), "fe_engine_hbm_rates3")
        return {"read": float(g[0]), "write": float(g[1]), "copy": float(g[2]), "write_regions": float(g[3]), "copy_regions": float(g[4]),
                "copy_best": float(g[5])}

    def close(self):
        if self.h:
            lib().fe_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Filter:
    """A convolver matrix under construction / committed (what config() builds)."""

    def __init__(self, engine, ninp, nout, maxsize, density=0.0):
        h = C.c_void_p()
        eh = engine.h if engine is not None else None
        _chk(lib().fe_filter_create(eh, ninp, nout, maxsize, density, C.byref(h)), "fe_filter_create")
        self.h = h
        self.engine = engine
        self.ninp, self.nout, self.maxsize = ninp, nout, maxsize
        self.block_size = lib().fe_filter_block_size(h)
        self.partitions = lib().fe_filter_partitions(h)

    @classmethod
    def from_handle(cls, handle, engine=None):
        self = cls.__new__(cls)
        L = lib()
        L.fe_filter_retain(handle)
        self.h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.engine = engine
        self.ninp, self.nout = L.fe_filter_inputs(self.h), L.fe_filter_outputs(self.h)
        self.maxsize = L.fe_filter_maxsize(self.h)
        self.block_size = L.fe_filter_block_size(self.h)
        self.partitions = L.fe_filter_partitions(self.h)
        return self

    def add(self, inp, out, data, ind0=0, step=1):
        data = np.ascontiguousarray(data, dtype=np.float32)
        n = (data.size + step - 1) // step
        _chk(lib().fe_filter_add(self.h, inp, out, step, _host_ptr(data), ind0, ind0 + n), "fe_filter_add")

    def link(self, inp1, out1, inp2, out2):
        _chk(lib().fe_filter_link(self.h, inp1, out1, inp2, out2), "fe_filter_link")

    def commit(self):
        _chk(lib().fe_filter_commit(self.h), "fe_filter_commit")
        return self

    def path_partitions(self, inp, out):
        return lib().fe_filter_path_partitions(self.h, inp, out)

    def taps(self, inp, out, n=None):
        n = n or self.partitions * self.block_size
        dst = np.zeros(n, np.float32)
        _chk(lib().fe_filter_get_taps(self.h, inp, out, _host_ptr(dst), n), "fe_filter_get_taps")
        return dst

    def open_stream(self, max_blocks_per_call=1):
        return Stream(self, max_blocks_per_call)

    def close(self):
        if self.h:
            lib().fe_filter_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """Per-file convolver state (what a pooled SoundProcessor owns)."""

    def __init__(self, flt, max_blocks_per_call=1):
        h = C.c_void_p()
        _chk(lib().fe_stream_open(flt.h, int(max_blocks_per_call), C.byref(h)), "fe_stream_open")
        self.h = h
        self.filter = flt
        self.max_blocks = max_blocks_per_call

    def reset(self):
        _chk(lib().fe_stream_reset(self.h), "fe_stream_reset")

    def process(self, x, valid=None):
        """One block, exactly SoundProcessor::Process: x [valid, ninp] -> ([valid, nout], peak_signed, peak_abs)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.filter.ninp)
        valid = x.shape[0] if valid is None else valid
        out = np.zeros((valid, self.filter.nout), np.float32)
        ps, pa = C.c_float(), C.c_float()
        _chk(lib().fe_stream_process(self.h, _host_ptr(x), valid, _host_ptr(out), C.byref(ps), C.byref(pa)),
             "fe_stream_process")
        return out, ps.value, pa.value

    def process_blocks(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.filter.ninp)
        out = np.zeros((x.shape[0], self.filter.nout), np.float32)
        _chk(lib().fe_stream_process_blocks(self.h, _host_ptr(x), x.shape[0], _host_ptr(out)),
             "fe_stream_process_blocks")
        return out

    def peaks(self):
        ps, pa = C.c_float(), C.c_float()
        _chk(lib().fe_stream_get_peaks(self.h, C.byref(ps), C.byref(pa)), "fe_stream_get_peaks")
        return ps.value, pa.value

    def reset_peaks(self):
        _chk(lib().fe_stream_reset_peaks(self.h), "fe_stream_reset_peaks")

    def blocks_done(self):
        return lib().fe_stream_blocks_done(self.h)

    def close(self):
        if self.h:
            lib().fe_stream_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchPlan:
    """Pre-marshalled argument arrays for repeated fe_batch_process calls on the
    same buffers (keeps Python overhead out of a timed loop)."""

    def __init__(self, streams, ins, outs, nframes, flags):
        n = len(streams)
        self.n = n
        self.flags = flags
        self.sa = (C.c_void_p * n)(*[s.h for s in streams])
        self.ia = (C.c_void_p * n)(*ins)
        self.oa = (C.c_void_p * n)(*outs)
        self.na = (C.c_longlong * n)(*nframes)
        self._keep = (streams,)

    def run(self):
        _chk(lib().fe_batch_process(self.sa, self.n, self.ia, self.na, self.oa, self.flags), "fe_batch_process")


def batch_process(streams, ins, outs=None, device=False, async_=False):
    """Batched call.  Host mode: ins are numpy [frames, ninp] arrays, returns numpy outputs.
    Device mode: ins/outs are torch CUDA float32 tensors ([frames, ch], contiguous)."""
    n = len(streams)
    if device:
        assert outs is not None
        for t in list(ins) + list(outs):
            assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float32"
        nfr = [int(t.shape[0]) for t in ins]
        plan = BatchPlan(streams, [t.data_ptr() for t in ins], [t.data_ptr() for t in outs], nfr,
                         FE_DEVICE_PTRS | (FE_ASYNC if async_ else 0))
        plan.run()
        return outs
    ins = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, s.filter.ninp) for x, s in zip(ins, streams)]
    outs = [np.zeros((x.shape[0], s.filter.nout), np.float32) for x, s in zip(ins, streams)]
    plan = BatchPlan(streams, [x.ctypes.data for x in ins], [y.ctypes.data for y in outs],
                     [x.shape[0] for x in ins], FE_HOST_PTRS)
    plan.run()
    return outs
