"""CPU: the register butterflies of the HIP FFT core, compiled for the host (same templates)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_radix_butterflies_and_plans(tmp_path):
    exe = os.path.join(str(tmp_path), "host_fft_check")
    subprocess.check_call(["g++", "-std=c++20", "-O2", "-o", exe, os.path.join(ROOT, "tests", "host_fft_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
