// Host-side check of the register butterflies in fft_core.hpp against a
// direct O(R^2) DFT in double.  Built and run by tests/test_host_fft.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../folve_amd/csrc/kernels/fft_core.hpp"

template <int R, bool INV>
static double check() {
    float2 v[R];
    double re[R], im[R];
    for (int i = 0; i < R; ++i) {
        re[i] = std::rand() / (double)RAND_MAX - 0.5;
        im[i] = std::rand() / (double)RAND_MAX - 0.5;
        v[i] = float2{(float)re[i], (float)im[i]};
        re[i] = v[i].x; im[i] = v[i].y;
    }
    fk::dft<R, INV>(v);
    double worst = 0;
    for (int k = 0; k < R; ++k) {
        double sr = 0, si = 0;
        for (int t = 0; t < R; ++t) {
            const double a = (INV ? 2.0 : -2.0) * M_PI * k * t / R;
            sr += re[t] * std::cos(a) - im[t] * std::sin(a);
            si += re[t] * std::sin(a) + im[t] * std::cos(a);
        }
        worst = std::fmax(worst, std::fmax(std::fabs(sr - v[k].x), std::fabs(si - v[k].y)));
    }
    return worst;
}

int main() {
    double w = 0;
    for (int it = 0; it < 100; ++it) {
        w = std::fmax(w, check<2, false>()); w = std::fmax(w, check<2, true>());
        w = std::fmax(w, check<4, false>()); w = std::fmax(w, check<4, true>());
        w = std::fmax(w, check<8, false>()); w = std::fmax(w, check<8, true>());
        w = std::fmax(w, check<16, false>()); w = std::fmax(w, check<16, true>());
    }
    // plan products
    for (int l = 6; l <= 13; ++l) {
        constexpr auto dummy = fk::make_plan(13); (void)dummy;
        fk::Plan p = fk::make_plan(l);
        int prod = 1;
        for (int i = 0; i < p.n; ++i) prod *= p.r[i];
        if (prod != (1 << l) || p.n < 2 || p.n > 4) { std::printf("bad plan %d\n", l); return 2; }
    }
    // the LDS image of the P = 8192 transform: element q + 512 c + 256 j lies 68 c + 34 j elements behind element q
    // (the constant offsets of the K3 walker's frame-major read-out of many-channel blocks, kernels.hip)
    {
        // WaveGeom<13>::at restated (the struct itself is device-side): rows of N2 padded elements, element k in row k % N1
        constexpr int N1 = 8, N2 = 1024, RS = fk::lds_elems(N2) + 32 / N1, P = 8192;
        auto at = [&](int k) { return (k % N1) * RS + fk::phys(k / N1); };
        for (int q = P / 2; q < P / 2 + 256; ++q)
            for (int c = 0; c < 8; ++c)
                for (int j = 0; j < 2; ++j)
                    if (at(q + 512 * c + 256 * j) != at(q) + 68 * c + 34 * j) { std::printf("bad image offset %d %d %d\n", q, c, j); return 3; }
    }
    std::printf("max_abs_err %.3e\n", w);
    return w < 2e-6 ? 0 : 1;
}
