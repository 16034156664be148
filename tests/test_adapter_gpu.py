"""GPU: the libsndfile adapter LINKED and RUN.  tests/compile/adapter_run.cpp compiles
host/sndfile_adapter.cpp — FillBuffer(SNDFILE*) / WriteProcessed(SNDFILE*, int), the reference's signatures
(/root/reference/sound-processor.h:35,55) — with sf_readf_float / sf_writef_float over in-memory float files inside the
test binary, links libfolve_amd.so, and drives two files through the call pattern of
/root/reference/convolve-file-handler.cc:328-351,370-424 via the reference's header names (include/dropin/).
What the output "files" received is compared with the float64 convolution.  (This exercises the adapter on the GPU;
it is not a pin of the oracle.)"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from fixtures import make_santalucia_shaped_dir, seeded_input

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5


@pytest.fixture(scope="module")
def adapter_exe(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    out = os.path.join(str(tmp_path_factory.mktemp("adapter")), "adapter_run")
    libdir = os.path.join(ROOT, "folve_amd")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread",
                        os.path.join(ROOT, "tests", "compile", "adapter_run.cpp"), "-o", out,
                        "-L" + libdir, "-lfolve_amd", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _run(exe, d, tmp, a, b, gapless, run_ahead):
    pa, pb, oa, ob = [os.path.join(str(tmp), n) for n in ("a.f32", "b.f32", "oa.f32", "ob.f32")]
    a.astype("<f4").tofile(pa)
    b.astype("<f4").tofile(pb)
    r = subprocess.run([exe, d, "44100", "2", str(int(gapless)), str(run_ahead), pa, pb, oa, ob],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    info = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return np.fromfile(oa, "<f4").reshape(-1, 2), np.fromfile(ob, "<f4").reshape(-1, 2), info


@pytest.mark.parametrize("run_ahead", [1, 8])
def test_adapter_members_linked_and_run_gapless(oracle, tmp_path, adapter_exe, run_ahead):
    d, hs = make_santalucia_shaped_dir(tmp_path)
    a = seeded_input(41, 6 * 8192 + 2345, 2)
    b = seeded_input(42, 9 * 8192 + 77, 2)
    ya, yb, info = _run(adapter_exe, d, tmp_path, a, b, True, run_ahead)
    assert ya.shape == a.shape and yb.shape == b.shape          # as many frames written as read, per file
    ref = oracle.linear_convolution_f64(np.concatenate([a, b]), hs, 2)
    assert oracle.rms(np.concatenate([ya, yb]) - ref) <= TOL     # gapless: the reverb of A continues into B
    assert info["max_a"] == pytest.approx(max(0.0, float(ya.max())), abs=1e-5)
    # without gapless the two files are convolved on their own
    ya2, yb2, _ = _run(adapter_exe, d, tmp_path, a, b, False, run_ahead)
    assert oracle.rms(ya2 - oracle.linear_convolution_f64(a, hs, 2)) <= TOL
    assert oracle.rms(yb2 - oracle.linear_convolution_f64(b, hs, 2)) <= TOL
    assert oracle.rms(yb2 - ref[len(a):]) > 1e-3
    if run_ahead > 1:
        # run-ahead asks libsndfile for many blocks at once: far fewer reads than blocks, one write per block
        assert info["reads_a"] < 6 and info["writes_a"] >= 7
