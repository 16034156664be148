"""The legs through the drop-in seam, beside the resident-PCM headline: the same batch from page-locked HOST buffers (PCIe
inside the timed region), one synchronous stereo block through fe_stream_process (SoundProcessor::Process), and N file
threads each its own folve::SoundProcessor (a C++ child process over include/folve_host.h)."""
import ctypes
import json
import os
import time

import numpy as np

from .formulas import FS
from .power import usable_cpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def end_to_end(h):
    """The headline's batch from page-locked host buffers: H2D + kernels + D2H pipelined in chunks of whole streams."""
    import torch
    from folve_amd.capi import BatchPlan, FE_HOST_PTRS
    S, T, C, P = h.S, h.T, h.C, h.P
    try:
        hin = [torch.empty(T * P, C).pin_memory() for _ in range(S)]
        hout = [torch.empty(T * P, C).pin_memory() for _ in range(S)]
        for s in range(S):
            hin[s].copy_(h.xs[s])
        h.sync()
        hs = [h.flt.open_stream(T) for _ in range(S)]
        hplan = BatchPlan(hs, [t_.data_ptr() for t_ in hin], [t_.data_ptr() for t_ in hout], [T * P] * S, FE_HOST_PTRS)
        hplan.run()
        ok = bool(np.allclose(hout[0].numpy(), h.y1[0], atol=2e-6)) if 0 in h.y1 else None
        nh = 6
        th = time.perf_counter()
        for _ in range(nh):
            hplan.run()
        dh = (time.perf_counter() - th) / nh
        out = {"msamples_per_s": round(S * T * P * C / dh / 1e6, 1), "ms_per_step": round(dh * 1e3, 3),
               "buffers": "page-locked host memory, H2D + kernels + D2H pipelined in chunks of whole streams (>= 64 MB each, up to 32)",
               "pcie_GBs_each_way": round(S * T * P * C * 4 / dh / 1e9, 1), "matches_resident_run": ok}
        for s_ in hs:
            s_.close()
        del hin, hout
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def single_block(h):
    """The drop-in call: one synchronous stereo block through fe_stream_process."""
    fa, flt, P, C, K = h.fa, h.flt, h.P, h.C, h.K
    try:
        L = fa.lib()
        nbytes = P * C * 4
        buf = ctypes.c_void_p()
        assert L.fe_host_alloc(nbytes, ctypes.byref(buf)) == 0
        st = flt.open_stream(1)
        assert L.fe_stream_bind_host_buffer(st.h, buf, nbytes) == 0
        arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), shape=(P * C,))
        arr[:] = np.random.default_rng(9).uniform(-1, 1, P * C).astype(np.float32)

        def loop(n, in_p, out_p, stream):
            t_ = time.perf_counter()
            for _ in range(n):
                rc = L.fe_stream_process(stream.h, in_p, P, out_p, None, None)
                assert rc == 0
            return (time.perf_counter() - t_) / n
        loop(K + 20, buf, buf, st)
        zc = min(loop(200, buf, buf, st) for _ in range(3))
        st2 = flt.open_stream(1)
        a_in = np.random.default_rng(9).uniform(-1, 1, P * C).astype(np.float32)
        a_out = np.zeros(P * C, np.float32)
        pi, po = a_in.ctypes.data_as(ctypes.c_void_p), a_out.ctypes.data_as(ctypes.c_void_p)
        loop(K + 20, pi, po, st2)
        staged = min(loop(200, pi, po, st2) for _ in range(3))
        out = {"single_block_us": round(zc * 1e6, 1), "staged_pageable_us": round(staged * 1e6, 1),
               "what": "fe_stream_process: one synchronous 8192-frame stereo block, K = %d, host pointers; "
                       "first figure with the block buffer page-locked and bound to the stream (what "
                       "folve::SoundProcessor does), second with ordinary memory (staged copies)" % K,
               "realtime_factor": round(P / FS / zc, 0)}
        st.close(); st2.close()
        L.fe_host_free(buf)
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def _write_filter_files(d, taps, C, size):
    """The headline's filter through the real loader: a 16-bit PCM WAV, as the demo filters' impulse files, and a .conf."""
    ir = np.stack(taps, axis=1).astype(np.float64)
    ir16 = np.round(ir / np.abs(ir).max() * 0.9 * 32767).astype("<i2")
    with open(os.path.join(d, "ir.wav"), "wb") as f:
        data = ir16.tobytes()
        f.write(b"RIFF" + (36 + len(data)).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") +
                (1).to_bytes(2, "little") + (C).to_bytes(2, "little") + (FS).to_bytes(4, "little") +
                (FS * C * 2).to_bytes(4, "little") + (C * 2).to_bytes(2, "little") + (16).to_bytes(2, "little") +
                b"data" + len(data).to_bytes(4, "little") + data)
    with open(os.path.join(d, "filter-44100.conf"), "w") as f:
        f.write("/convolver/new %d %d 256 %d\n" % (C, C, size))
        for c in range(C):
            f.write("/impulse/read %d %d 2e-3 0 0 0 %d ir.wav\n" % (c + 1, c + 1, c + 1))
    return os.path.join(d, "filter-44100.conf")


def _child(exe, conf, args, timeout, env=None):
    import subprocess
    r = subprocess.run([exe, conf] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                       timeout=timeout, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else {"error": "rc %d" % r.returncode}


def drop_in_threads(h, e2e_gbs=None):
    """The drop-in call under load: N file threads, each its own folve::SoundProcessor pulling single blocks as
    ConvolveFileHandler does — a C++ host over include/folve_host.h (tools/dropin/dropin_threads.cpp, built by
    __graft_entry__.build()), run as a child process; the same filter through the real loader (.conf + WAV)."""
    import tempfile
    P, C, K = h.P, h.C, h.K
    try:
        exe = os.path.join(ROOT, "tools", "dropin", "dropin_threads")
        if not os.path.exists(exe):
            raise RuntimeError("tools/dropin/dropin_threads not built (python -c 'import __graft_entry__ as g; g.build()')")
        conf = _write_filter_files(tempfile.mkdtemp(prefix="folve_dropin_"), h.taps, C, h.size)
        runs = []
        # (threads, combiner, run-ahead depth in blocks): depth 1 is the reference's one block per Process() call
        for nt, comb, ra in ((1, 1, 1), (1, 1, 64), (16, 1, 64), (64, 1, 1), (128, 1, 1), (64, 1, 32), (64, 1, 64), (64, 1, 128), (64, 0, 1)):
            # long enough that the run-ahead ramp and the ragged end (threads finishing their last chunks) do not weigh
            nblk = (300 if not comb else 2000) if ra == 1 else (20000 if nt == 1 else 8192 if nt <= 16 else max(4096, 48 * ra))
            r = _child(exe, conf, [nt, nblk, comb, "json", "run_ahead=%d" % ra], 180)
            if "error" in r:
                r.update({"threads": nt, "combiner": bool(comb), "run_ahead": ra})
            runs.append(r)
        # The multi-GPU path folve itself would run: ONE process, ProcessorPool -> DeviceRouter spreading the open files
        # over every visible GPU (least-loaded, sticky), each file thread and its page-locked ring placed on its GPU's
        # NUMA node.  Only when more than one GPU is visible to this process.
        ndev = h.fa.lib().fe_device_count()
        multi = None
        if ndev > 1:
            nt = min(64 * ndev, 512)
            env = dict(os.environ)
            env.pop("FOLVE_AMD_DEVICES", None)
            multi = _child(exe, conf, [nt, 2048, 1, "json", "run_ahead=64", "pin=1"], 300, env)
            multi["what"] = ("one process, %d file threads over %d GPUs through folve::DeviceRouter (streams to the least-loaded "
                             "GPU, one combiner and one engine per GPU, no collective), run-ahead 64, threads and rings "
                             "NUMA-placed next to their GPU" % (nt, ndev))
        # cfg5's shape on ONE device: eight router slots (eight engines, combiners and copies of the filter) on this GPU,
        # 512 file threads, 64 per slot — everything of the 8-GPU in-process path except seven more devices and buses.
        cfg5_one = None
        if ndev == 1:
            env = dict(os.environ, FOLVE_AMD_DEVICES="0,0,0,0,0,0,0,0")
            cfg5_one = _child(exe, conf, [512, 2048, 1, "json", "run_ahead=64"], 300, env)
            cfg5_one["what"] = ("cfg5's shape on one device: 512 file threads over 8 router slots (FOLVE_AMD_DEVICES=0,0,0,0,0,0,0,0), "
                                "64 streams per slot, run-ahead 64; one GPU and one bus carry all eight slots.  NOT a stand-in for the "
                                "8-GPU rate: engines that share a device share its copy engines and its bus, and 512 threads share "
                                "this box's CPU quota; it shows that the sharder places 64 streams on every slot and that all eight "
                                "engines, combiners and pipelines run at once")
        # every run against the bus: bytes each way per second, and as a fraction of what `end_to_end` moved in this run
        for r_ in runs:
            if r_.get("blocks_per_s"):
                gbs = r_["blocks_per_s"] * P * C * 4 / 1e9
                r_["pcie_GBs_each_way"] = round(gbs, 2)
                r_["of_end_to_end"] = round(gbs / e2e_gbs, 3) if e2e_gbs else None
        return {"what": "N host threads, each a folve::SoundProcessor (page-locked ring, per-GPU combiner) pulling 8192-frame "
                        "stereo blocks as ConvolveFileHandler::AddMoreSoundData does: FillBuffer -> WriteProcessed over "
                        "sf_readf_float / sf_writef_float-shaped callbacks that copy every block in and out, K = %d; "
                        "run_ahead = blocks a processor reads ahead of its reader (1 = the reference's one block per "
                        "Process() call); child process, tools/dropin/dropin_threads.cpp" % K,
                "usable_cpus": usable_cpus()[0], "runs": runs, "multi_gpu": multi, "cfg5_shape_one_device": cfg5_one}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}
