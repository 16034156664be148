"""numpy model of the index math used by folve_amd/csrc/kernels (design aid).

Validates, against numpy.fft, the exact formulas the HIP kernels implement:
  * Stockham autosort passes with a mixed radix plan,
  * real FFT of a 2P window through a P-point complex FFT, packed spectrum
    (bin 0 holds (DC, Nyquist)),
  * packed multiply-accumulate,
  * inverse through the P-point complex FFT, overlap-save (keep last P).
Run: python tools/fft_model.py
"""
import numpy as np


def plan(log2n):
    r, out = log2n, []
    while r >= 4 and r != 5:
        out.append(16); r -= 4
    while r >= 3:
        out.append(8); r -= 3
    if r == 2: out.append(4)
    if r == 1: out.append(2)
    assert np.prod(out) == 1 << log2n
    return out


def stockham(z, inverse=False):
    n = len(z)
    sign = 1.0 if inverse else -1.0
    a = z.astype(np.complex128).copy()
    ns = 1
    for R in plan(int(np.log2(n))):
        b = np.empty_like(a)
        j = np.arange(n // R)
        k = j % ns
        v = np.stack([a[j + r * (n // R)] * np.exp(sign * 2j * np.pi * k * r / (ns * R)) for r in range(R)])
        # R-point DFT, natural order
        W = np.exp(sign * 2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
        v = W @ v
        j0 = (j - k) * R + k
        for r in range(R):
            b[j0 + r * ns] = v[r]
        a = b
        ns *= R
    return a


def fwd_packed(win):
    """win: 2P real -> packed P complex (bin0 = DC + i*Nyq)."""
    P = len(win) // 2
    z = win[0::2] + 1j * win[1::2]
    Z = stockham(z)
    X = np.empty(P, np.complex128)
    X[0] = (Z[0].real + Z[0].imag) + 1j * (Z[0].real - Z[0].imag)
    k = np.arange(1, P)
    Zk, Zc = Z[k], np.conj(Z[P - k])
    E = 0.5 * (Zk + Zc)
    O = -0.5j * (Zk - Zc)
    X[k] = E + np.exp(-1j * np.pi * k / P) * O
    return X


def inv_packed(Y):
    """packed P complex -> 2P real * (2P) (unnormalised)."""
    P = len(Y)
    Z = np.empty(P, np.complex128)
    Z[0] = (Y[0].real + Y[0].imag) + 1j * (Y[0].real - Y[0].imag)
    k = np.arange(1, P)
    Yk, Yc = Y[k], np.conj(Y[P - k])
    E = Yk + Yc
    O = (Yk - Yc) * np.exp(1j * np.pi * k / P)
    Z[k] = E + 1j * O
    z = stockham(Z, inverse=True)
    y = np.empty(2 * P)
    y[0::2] = z.real
    y[1::2] = z.imag
    return y


def mac_packed(acc, X, H):
    acc[0] += X[0].real * H[0].real + 1j * (X[0].imag * H[0].imag)
    acc[1:] += X[1:] * H[1:]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for P in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        z = rng.standard_normal(P) + 1j * rng.standard_normal(P)
        assert np.allclose(stockham(z), np.fft.fft(z))
        assert np.allclose(stockham(z, True), np.fft.ifft(z) * P)
        w = rng.standard_normal(2 * P)
        X = fwd_packed(w)
        F = np.fft.rfft(w)
        assert np.allclose(X[1:], F[1:P]) and np.isclose(X[0].real, F[0].real) and np.isclose(X[0].imag, F[P].real)
        assert np.allclose(inv_packed(X), w * 2 * P)
        # overlap-save partitioned convolution through packed spectra
        K, nb = 3, 6
        h = rng.standard_normal(K * P)
        x = rng.standard_normal(nb * P)
        Hs = [fwd_packed(np.concatenate([h[j * P:(j + 1) * P], np.zeros(P)])) / (2 * P) for j in range(K)]
        xp = np.concatenate([np.zeros(P), x])
        Xs = [fwd_packed(xp[n * P:(n + 2) * P]) for n in range(nb)]
        y = np.zeros(nb * P)
        for n in range(nb):
            acc = np.zeros(P, np.complex128)
            for j in range(K):
                if n - j >= 0:
                    mac_packed(acc, Xs[n - j], Hs[j])
            y[n * P:(n + 1) * P] = inv_packed(acc)[P:]
        ref = np.convolve(x, h)[: nb * P]
        assert np.allclose(y, ref), (P, np.abs(y - ref).max())
        print("P=%d ok plan=%s" % (P, plan(int(np.log2(P)))))


# ---------------------------------------------------------------------------
# Model of the wave-autonomous decomposition used by kernels.hip (v2):
#   N = N1 * N2, N1 = wavefronts per workgroup, N2 = 1024 (or N when N < 1024)
#   stage A: A[k1][n2] = DFT_N1 over n1 of z[n1*N2 + n2], times W_N^(n2*k1)
#   stage B: wave k1: Z[k1 + N1*k2] = DFT_N2 over n2 of A[k1][n2]   (Stockham, in its LDS row)
# ---------------------------------------------------------------------------
def two_level(q, twoP):
    """exp(-2*pi*i*q/twoP) as coarse[q>>5] * fine[q&31], like the LDS tables."""
    q = np.asarray(q) % twoP
    coarse = np.exp(-2j * np.pi * (q >> 5) * 32 / twoP)
    fine = np.exp(-2j * np.pi * (q & 31) / twoP)
    return coarse * fine


def wave_fft(z, inverse=False):
    n = len(z)
    n1 = max(1, n // 1024)
    n2 = n // n1
    sign = 1.0 if inverse else -1.0
    rows = np.empty((n1, n2), np.complex128)
    col = np.arange(n2)
    x = z.reshape(n1, n2)                       # x[n1][n2] = z[n1*N2 + n2]
    W = np.exp(sign * 2j * np.pi * np.outer(np.arange(n1), np.arange(n1)) / n1)
    a = W @ x                                   # a[k1][n2]
    for k1 in range(n1):
        tw = two_level(col * k1 * 2, 2 * n)     # W_N^(n2*k1) as a 2N-th root index
        rows[k1] = a[k1] * (np.conj(tw) if inverse else tw)
    out = np.empty(n, np.complex128)
    for k1 in range(n1):
        out[k1::n1] = stockham(rows[k1], inverse)   # Z[k1 + N1*k2]
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for P in (64, 256, 1024, 2048, 4096, 8192):
        z = rng.standard_normal(P) + 1j * rng.standard_normal(P)
        assert np.allclose(wave_fft(z), np.fft.fft(z))
        assert np.allclose(wave_fft(z, True), np.fft.ifft(z) * P)
    print("wave-autonomous decomposition ok")


# ---------------------------------------------------------------------------
# Model of the v3 pipeline identities:
#   (a) zero-padded forward transforms:  X(n) = FFT([x(n-1) | x(n)]) = Z(n-1) + s*Z(n),
#       Z(n) = FFT([x(n) | 0]), s_k = (-1)^k, so  Y(n) = sum_{j<=K} Z(n-j) * G(j),
#       G(j) = s*H(j) + H(j-1)  (H(-1) = H(K) = 0)
#   (b) stereo as one complex signal: FFT(L + iR) -> ZL, ZR by symmetry; IFFT(YL + i*YR) -> (yL, yR)
# ---------------------------------------------------------------------------
def check_v3(P=256, K=3, nb=7, seed=2):
    rng = np.random.default_rng(seed)
    N = 2 * P
    h = rng.standard_normal((2, K * P))
    x = rng.standard_normal((2, nb * P))
    s = (-1.0) ** np.arange(P + 1)
    H = [[np.fft.rfft(np.concatenate([h[c, j * P:(j + 1) * P], np.zeros(P)])) / N for j in range(K)] for c in range(2)]
    G = [[(s * H[c][j] if j < K else 0) + (H[c][j - 1] if j >= 1 else 0) for j in range(K + 1)] for c in range(2)]
    # (b) forward: one complex FFT for both channels of a zero-padded block
    Zs = []
    for n in range(nb):
        z = np.concatenate([x[0, n * P:(n + 1) * P] + 1j * x[1, n * P:(n + 1) * P], np.zeros(P)])
        Z = np.fft.fft(z)
        k = np.arange(P + 1)
        Zm = np.conj(Z[(N - k) % N])
        ZL, ZR = 0.5 * (Z[k] + Zm), -0.5j * (Z[k] - Zm)
        assert np.allclose(ZL, np.fft.rfft(np.concatenate([x[0, n * P:(n + 1) * P], np.zeros(P)])))
        assert np.allclose(ZR, np.fft.rfft(np.concatenate([x[1, n * P:(n + 1) * P], np.zeros(P)])))
        Zs.append((ZL, ZR))
    y = np.zeros((2, nb * P))
    for n in range(nb):
        Y = [sum(Zs[n - j][c] * G[c][j] for j in range(K + 1) if n - j >= 0) for c in range(2)]
        # (b) inverse: Z[k] = YL[k] + i YR[k] (k <= P), Z[N-k] = conj(YL[k]) + i conj(YR[k])
        Zc = np.zeros(N, complex)
        k = np.arange(P + 1)
        Zc[k] = Y[0] + 1j * Y[1]
        kk = np.arange(1, P)
        Zc[N - kk] = np.conj(Y[0][kk]) + 1j * np.conj(Y[1][kk])
        zt = np.fft.ifft(Zc) * N
        y[0, n * P:(n + 1) * P] = zt.real[P:]
        y[1, n * P:(n + 1) * P] = zt.imag[P:]
    for c in range(2):
        assert np.allclose(y[c], np.convolve(x[c], h[c])[: nb * P])
    return True


if __name__ == "__main__":
    assert check_v3() and check_v3(P=64, K=1, nb=4) and check_v3(P=128, K=5, nb=3)
    print("v3 identities (zero-padded transforms, G filter, dual-real stereo) ok")
