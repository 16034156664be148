"""ctypes binding of include/folve_host.h: the SoundProcessor / ProcessorPool / config-loader mirror."""
import ctypes as C
import os

import numpy as np

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
_pi = C.POINTER(C.c_int)
HOST_SYMBOLS = [
    ("fh_sstring", _i, [C.c_char_p, C.c_char_p, _i]),
    ("fh_config_load", _i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp), _pi, _pi, _pi, _pi]),
    ("fh_processor_create", _vp, [C.c_char_p, _i, _i]),
    ("fh_processor_destroy", None, [_vp]),
    ("fh_processor_fill_buffer", _i, [_vp, _vp, _i]),
    ("fh_processor_fill_buffer2", _i, [_vp, _vp, _i, _pi]),
    ("fh_processor_write_processed", None, [_vp, _vp, _i]),
    ("fh_processor_is_input_buffer_complete", _i, [_vp]),
    ("fh_processor_pending_writes", _i, [_vp]),
    ("fh_processor_input_channels", _i, [_vp]),
    ("fh_processor_output_channels", _i, [_vp]),
    ("fh_processor_block_size", _i, [_vp]),
    ("fh_processor_max_output_value", _f, [_vp]),
    ("fh_processor_max_abs_output_value", _f, [_vp]),
    ("fh_processor_reset_max_values", None, [_vp]),
    ("fh_processor_reset", None, [_vp]),
    ("fh_processor_config_file", C.c_char_p, [_vp]),
    ("fh_processor_config_file_timestamp", _ll, [_vp]),
    ("fh_processor_config_still_up_to_date", _i, [_vp]),
    ("fh_processor_device", _i, [_vp]),
    ("fh_processor_stream", _vp, [_vp]),
    ("fh_router_cached_filters", _i, []),
    ("fh_processor_engine", _vp, [_vp]),
    ("fh_processor_ok", _i, [_vp]),
    ("fh_processor_moves", _i, [_vp]),
    ("fh_survival_set", None, [_i]),
    ("fh_pool_create", _vp, [_i]),
    ("fh_pool_destroy", None, [_vp]),
    ("fh_pool_get_or_create", _vp, [_vp, C.c_char_p, _i, _i, _i, C.c_char_p, _i]),
    ("fh_pool_return", None, [_vp, _vp]),
    ("fh_pool_pooled_count", _i, [_vp, C.c_char_p]),
    ("fh_batching_set", None, [_i, _i, _i]),
    ("fh_batching_enabled", _i, []),
    ("fh_batching_early_quarters", None, [_i]),
    ("fh_batcher_process", _i, [_vp, _vp, _vp, _i, _vp]),
    ("fh_batching_stats", None, [C.POINTER(_ll), C.POINTER(_ll), C.POINTER(_ll)]),
    ("fh_batching_stats2", None, [C.POINTER(_ll)] * 5),
    ("fh_run_ahead_set", None, [_i]),
    ("fh_device_peaks_set", None, [_i]),
    ("fh_run_ahead_get", _i, []),
    ("fh_processor_run_ahead", _i, [_vp]),
    ("fh_processor_fill_buffer_from", _i, [_vp, _vp, _vp]),
    ("fh_processor_write_processed_to", None, [_vp, _vp, _vp, _i]),
    ("fh_numa_placement_set", None, [_i]),
    ("fh_pin_thread_near_device", _i, [_i]),
    ("fh_router_device_count", _i, []),
    ("fh_router_live_streams", _i, [_i]),
    ("fh_router_slot_state", _i, [_i]),
    ("fh_router_slot_failures", _ll, [_i]),
    ("fh_router_slot_engine", _vp, [_i]),
    ("fh_router_health_policy", None, [_i, C.c_double]),
    ("fh_router_report_failure", None, [_vp]),
]

# zita-config.h:51
NOERR, ERR_OTHER, ERR_SYNTAX, ERR_PARAM, ERR_ALLOC, ERR_CANTCD, ERR_COMMAND, ERR_NOCONV, ERR_IONUM = range(9)


def declare(L):
    for name, res, args in HOST_SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


def _L():
    from .capi import lib
    return lib()


def sstring(src: bytes, size=1024):
    buf = C.create_string_buffer(max(size, 1) + 8)
    n = _L().fh_sstring(src, buf, size)
    return n, buf.value


def config_load(config_file, fsamp=44100, channels=2, engine=None):
    """Parse a jconvolver config into an uncommitted Filter (host only when engine is None).
    Returns (status, Filter or None, dict(fragm, ninp, nout, size))."""
    from .capi import Filter
    fh = C.c_void_p()
    vals = [C.c_int() for _ in range(4)]
    st = _L().fh_config_load(engine.h if engine is not None else None, os.fsencode(config_file), fsamp, channels,
                             C.byref(fh), *[C.byref(v) for v in vals])
    flt = None
    if fh.value:
        flt = Filter.from_handle(fh, engine)
        _L().fe_filter_release(fh)      # from_handle took its own reference
    return st, flt, dict(zip(("fragm", "ninp", "nout", "size"), [v.value for v in vals]))


class SoundProcessor:
    """folve::SoundProcessor driven over float spans (what ConvolveFileHandler does with SNDFILE*)."""

    def __init__(self, handle, owned=True):
        self.h = handle
        self.owned = owned
        L = _L()
        self.ninp = L.fh_processor_input_channels(handle)
        self.nout = L.fh_processor_output_channels(handle)
        self.fragm = L.fh_processor_block_size(handle)
        self._ahead = 0        # frames the processor has read beyond the ones it returned (run-ahead)

    @classmethod
    def create(cls, config_file, samplerate, channels):
        h = _L().fh_processor_create(os.fsencode(config_file), samplerate, channels)
        return cls(h) if h else None

    def fill_buffer(self, src):
        """FillBuffer over an array span.  The reference reads from a file with a position (SNDFILE*); callers here
        pass `x[done:]` with `done` advanced by the return value.  With run-ahead on the processor reads ahead of
        what it returns, so this wrapper keeps the file position: the span handed to the C++ side starts where
        the last read ended."""
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, self.ninp)
        span = src[self._ahead:]
        taken = C.c_int(0)
        r = _L().fh_processor_fill_buffer2(self.h, span.ctypes.data_as(C.c_void_p), span.shape[0], C.byref(taken))
        self._ahead += taken.value - r
        return r

    def write_processed(self, count):
        out = np.zeros((count, self.nout), np.float32)
        _L().fh_processor_write_processed(self.h, out.ctypes.data_as(C.c_void_p), count)
        return out

    def run_ahead(self):
        return _L().fh_processor_run_ahead(self.h)

    def pending_writes(self):
        return _L().fh_processor_pending_writes(self.h)

    def is_input_buffer_complete(self):
        return bool(_L().fh_processor_is_input_buffer_complete(self.h))

    def max_output_value(self):
        return float(_L().fh_processor_max_output_value(self.h))

    def max_abs_output_value(self):
        return float(_L().fh_processor_max_abs_output_value(self.h))

    def reset_max_values(self):
        _L().fh_processor_reset_max_values(self.h)

    def reset(self):
        _L().fh_processor_reset(self.h)
        self._ahead = 0

    def config_file(self):
        return os.fsdecode(_L().fh_processor_config_file(self.h))

    def config_file_timestamp(self):
        return _L().fh_processor_config_file_timestamp(self.h)

    def config_still_up_to_date(self):
        return bool(_L().fh_processor_config_still_up_to_date(self.h))

    def device(self):
        return _L().fh_processor_device(self.h)

    def run(self, x):
        """The AddMoreSoundData loop (convolve-file-handler.cc:370-424, non-gapless)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.ninp)
        outs, done = [], 0
        while done < x.shape[0]:
            r = self.fill_buffer(x[done:])
            assert r > 0
            outs.append(self.write_processed(r))
            done += r
        return np.concatenate(outs, 0) if outs else np.zeros((0, self.nout), np.float32)

    def close(self):
        if self.h and self.owned:
            _L().fh_processor_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProcessorPool:
    def __init__(self, max_per_config=3):
        self.h = _L().fh_pool_create(max_per_config)

    def get_or_create(self, base_dir, sampling_rate, channels, bits):
        err = C.create_string_buffer(2048)
        h = _L().fh_pool_get_or_create(self.h, os.fsencode(base_dir), sampling_rate, channels, bits, err, 2048)
        if not h:
            return None, err.value.decode(errors="replace")
        return SoundProcessor(h, owned=False), ""

    def give_back(self, proc):
        """ProcessorPool::Return."""
        if proc is None:
            _L().fh_pool_return(self.h, None)
            return
        h, proc.h = proc.h, None
        _L().fh_pool_return(self.h, h)

    def pooled_count(self, config_path):
        return _L().fh_pool_pooled_count(self.h, os.fsencode(config_path))

    def close(self):
        if self.h:
            _L().fh_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_batching(enabled, window_us=-1, max_batch=-1):
    """Run-ahead batcher switch (folve::BatchScheduler)."""
    _L().fh_batching_set(int(bool(enabled)), window_us, max_batch)


def batching_stats():
    v = [C.c_longlong() for _ in range(5)]
    _L().fh_batching_stats2(*[C.byref(x) for x in v])
    return dict(zip(("requests", "blocks", "batches", "largest", "overlapped"), [x.value for x in v]))


DEFAULT_RUN_AHEAD = 64     # blocks of 8192 frames; the automatic setting gives shorter blocks as many frames (up to 1024 blocks)
AUTO_RUN_AHEAD = 0         # set_run_ahead(AUTO_RUN_AHEAD): back to the automatic depth


def set_run_ahead(blocks):
    """Run-ahead depth (blocks) of SoundProcessors created from now on; 1 = off, 0 = automatic (folve::SoundProcessor::SetRunAhead)."""
    _L().fh_run_ahead_set(int(blocks))


def run_ahead():
    return _L().fh_run_ahead_get()
