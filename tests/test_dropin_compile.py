"""CPU: the reference's call sites of the convolver seam compile, unchanged in form, against this
repo's SoundProcessor / ProcessorPool — including the two libsndfile members
(/root/reference/sound-processor.h:35,55), whose definitions (host/sndfile_adapter.cpp) are
type-checked here against the prototypes they need.  Compile only: the image has no libsndfile."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_reference_call_sites_compile_against_the_dropin_classes(tmp_path):
    src = os.path.join(ROOT, "tests", "compile", "folve_call_sites.cpp")
    obj = os.path.join(str(tmp_path), "call_sites.o")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", src, "-o", obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    # the reference-signature members are defined by the adapter, and they call libsndfile
    assert "T folve::SoundProcessor::FillBuffer(SNDFILE_tag*)" in syms
    assert "T folve::SoundProcessor::WriteProcessed(SNDFILE_tag*, int)" in syms
    assert "U sf_readf_float" in syms and "U sf_writef_float" in syms
    # ... and the impulse-file fallback over sf_open / sf_seek / sf_close (zita-audiofile.cc:51-99,170-182)
    assert "U sf_open" in syms and "U sf_seek" in syms and "U sf_close" in syms


def test_library_itself_does_not_depend_on_libsndfile():
    import folve_amd as fa
    out = subprocess.run(["nm", "-D", fa.lib_path()], capture_output=True, text=True).stdout
    assert "sf_readf_float" not in out and "sf_writef_float" not in out
