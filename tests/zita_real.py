"""The real libzita-convolver, where a box has it: build tests/compile/zita_ref.cpp against it and run it.
Used by tests/test_zita_gpu.py (parity of the oracle and of the HIP path against the REAL library — the one route by which
the oracle's parity is ever pinned by the reference's own arithmetic) and by bench.py's cpu_baseline leg (the real library
timed through the same stream / thread shape as the restatement).  Nothing here ships zita or stands in for
its headers: without <zita-convolver.h> + libzita-convolver + libfftw3f the build says so and callers skip."""
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "compile", "zita_ref.cpp")


def build(workdir=None):
    """(path of the built program, None) or (None, why not)."""
    workdir = workdir or tempfile.mkdtemp(prefix="zita_ref_")
    exe = os.path.join(workdir, "zita_ref")
    try:
        r = subprocess.run(["g++", "-O3", "-march=native", "-std=c++17", "-pthread", SRC, "-o", exe, "-lzita-convolver", "-lfftw3f"],
                           capture_output=True, text=True, timeout=300)
    except (OSError, subprocess.TimeoutExpired) as e:
        return None, "g++ did not run: %r" % (e,)
    if r.returncode != 0:
        why = "libzita-convolver / libfftw3f not installed" if ("cannot find -lzita-convolver" in r.stderr or "cannot find -lfftw3f" in r.stderr) else r.stderr[-300:]
        return None, "zita_ref does not link here (%s)" % why.strip()
    r = subprocess.run([exe], capture_output=True, text=True)
    if r.returncode == 77:
        return None, "<zita-convolver.h> is not on this box"
    return exe, None


def run(exe, channels, size, taps, x, workdir):
    """taps [channels][size] float32, x [frames][channels] float32 -> y [frames][channels] float32 (real zita-convolver)."""
    import numpy as np
    tp, ip, op = (os.path.join(workdir, n) for n in ("taps.f32", "in.f32", "out.f32"))
    np.ascontiguousarray(taps, np.float32).tofile(tp)
    np.ascontiguousarray(x, np.float32).tofile(ip)
    r = subprocess.run([exe, "run", str(channels), str(size), tp, ip, op], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("zita_ref run failed (%d): %s" % (r.returncode, r.stderr[-500:]))
    info = json.loads(r.stdout.strip().splitlines()[-1])
    return np.fromfile(op, np.float32).reshape(-1, channels), info


def bench(exe, channels, size, streams, blocks, threads):
    r = subprocess.run([exe, "bench", str(channels), str(size), str(streams), str(blocks), str(threads)], capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("zita_ref bench failed (%d): %s" % (r.returncode, r.stderr[-500:]))
    return json.loads(r.stdout.strip().splitlines()[-1])
