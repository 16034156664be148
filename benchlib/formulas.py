"""Byte and flop formulas of the hot path, the ground truth of the parity gates, and the peaks
every fraction divides by (/opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s, FP32 vector 157.3 TFLOP/s at 2.4 GHz)."""
import numpy as np

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md:36 — 8.0 TB/s spec (6.29 TB/s quoted for a float4 copy)
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md:41 — FP32 vector at the 2.4 GHz top clock (256 CUs x 4 SIMDs x 64 flop/clk)
SCLK_MAX_MHZ = 2400.0
FS = 44100
PARITY_TOL = 1e-5


def alg_bytes(P, K_eff, S):
    """bytes per block-channel of the STREAMING uniformly partitioned algorithm
    (every output block reads K spectra of its stream and K of the filter), and its per-kernel split."""
    fwd = 4 * P + 8 * (P + 1)                 # read the block's PCM once; write one spectrum
    mac = 8 * (P + 1) * K_eff + 8 * (P + 1) * K_eff / S   # read K spectra + the shared filter
    inv = 8 * (P + 1) + 4 * P                 # read the accumulated spectrum; write P samples
    return {"forward": fwd, "mac": mac, "inverse": inv, "total": 12 * P + 8 * (P + 1) * (K_eff + 1) + 8 * (P + 1) * K_eff / S}


def tiled_bytes(P, K, T):
    """Bytes per block-channel a run-ahead call of T blocks must move at least: every PCM sample in
    and out once, every spectrum written once and read once by K2 (plus the K history rows per call),
    every accumulated spectrum written and read once."""
    fwd = 4 * P + 8 * P
    mac = 8 * P * (T + K) / T + 8 * P
    inv = 8 * P + 4 * P
    return {"forward": fwd, "mac": mac, "inverse": inv, "total": fwd + mac + inv}


def conv_f64(x, taps):
    """Exact causal linear convolution per channel, float64, truncated to len(x) (the ground truth)."""
    from scipy.signal import fftconvolve
    n = x.shape[0]
    return np.stack([fftconvolve(x[:, c].astype(np.float64), taps[c].astype(np.float64))[:n] for c in range(x.shape[1])], 1)


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


def walk_flops(kernel_name, P, K, T, stream_outputs, paths_per_output=1):
    """Floating-point operations K2 ISSUES per launch: per (stream, output), path, block and bin one complex multiply-add
    against each of the K + 1 rows of G — three real FMAs (6 flop) in the three-FMA walk (mac_walk3.hip), four (8 flop)
    in every other form.  cfg4: 6 x 65 x 8192 x 256 x 8 = 6.54 Gflop; the 2 x 2 matrix: 6 x 33 x 8192 x 256 x 128 x 2 = 106 Gflop."""
    per_mac = 6 if "mac_walk3" in (kernel_name or "") else 8
    return per_mac * (K + 1) * P * T * stream_outputs * paths_per_output


def valu_fractions(flops, kernel_ms, sclk_mhz=None):
    """K2 against the FP32 vector roof: TFLOP/s issued, the fraction of the nominal 157.3 (2.4 GHz) and — with the shader
    clock measured while the loop ran (the socket's power cap holds it near 1.7 GHz) — of what that clock gives."""
    if not kernel_ms:
        return None
    tf = flops / (kernel_ms * 1e-3) / 1e12
    out = {"tflops": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "frac": round(tf / VALU_PEAK_TFLOPS, 4), "flops_per_launch": int(flops)}
    if sclk_mhz:
        out["sclk_mhz"] = round(sclk_mhz, 0)
        out["frac_at_sclk"] = round(tf / (VALU_PEAK_TFLOPS * sclk_mhz / SCLK_MAX_MHZ), 4)
    return out
