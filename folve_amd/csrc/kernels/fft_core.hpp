// fft_core.hpp — workgroup-resident complex FFT for gfx950 (CDNA4).
//
// One workgroup transforms N = P complex points held in LDS (P = folve's
// `fragm`, 64..8192: zita-fconfig.cc:74-77 in the reference), which is the
// half-size complex transform behind the 2P-point real FFT of one partition.
// Stockham autosort passes (no bit reversal), radix 16/8/4/2 butterflies in
// registers, 64-wide wavefronts, in-place in a padded LDS image so that the
// strided writes of the early passes are bank-conflict free.
//
// The butterflies are plain C++ templates (usable from host code too, which is
// how tests/host_fft_check.cpp verifies them without a GPU).
#pragma once

#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FK_HD __host__ __device__ __forceinline__
#define FK_D __device__ __forceinline__
#else
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
#define FK_HD inline
#endif

namespace fk {

FK_HD float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
FK_HD float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
FK_HD float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
FK_HD float2 cmulc(float2 a, float2 b) { return float2{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }  // a * conj(b)

// 16th roots of unity: cos/sin(2*pi*i/16), i = 0..7.
constexpr float kCos16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
constexpr float kSin16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                             1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// x * exp(-/+ 2*pi*i*IDX/16)  (INV: +)
template <int IDX, bool INV>
FK_HD float2 mul_w16(float2 x) {
    if constexpr (IDX == 0) {
        return x;
    } else if constexpr (IDX == 4) {
        return INV ? float2{-x.y, x.x} : float2{x.y, -x.x};
    } else {
        constexpr float wr = kCos16[IDX];
        constexpr float wi = INV ? kSin16[IDX] : -kSin16[IDX];
        return float2{x.x * wr - x.y * wi, x.x * wi + x.y * wr};
    }
}

template <int R, bool INV>
FK_HD void dft(float2 (&v)[R]);

template <int R, bool INV, int... K>
FK_HD void dft_combine(float2 (&v)[R], const float2 (&e)[R / 2], const float2 (&o)[R / 2],
                       std::integer_sequence<int, K...>) {
    ((void)([&] {
         const float2 t = mul_w16<K * 16 / R, INV>(o[K]);
         v[K] = cadd(e[K], t);
         v[K + R / 2] = csub(e[K], t);
     }()),
     ...);
}

// R-point DFT, natural order in and out (decimation in time, fully unrolled).
template <int R, bool INV>
FK_HD void dft(float2 (&v)[R]) {
    static_assert(R == 2 || R == 4 || R == 8 || R == 16, "radix");
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else {
        float2 e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = v[2 * k];
            o[k] = v[2 * k + 1];
        }
        dft<R / 2, INV>(e);
        dft<R / 2, INV>(o);
        dft_combine<R, INV>(v, e, o, std::make_integer_sequence<int, R / 2>{});
    }
}

// ---- radix plan -----------------------------------------------------------
// log2(N) -> up to four radices, product N.  13 (P = 8192) -> 16,16,8,4.
struct Plan {
    int r[4];       // radix of each pass
    int ns[4];      // product of the earlier radices (twiddle period of the pass)
    int off[4];     // offset of the pass's twiddle table in the per-P pass-twiddle buffer
    int n;          // passes
    int total;      // entries in the pass-twiddle buffer
};
constexpr Plan make_plan(int log2n) {
    Plan p{{1, 1, 1, 1}, {1, 1, 1, 1}, {0, 0, 0, 0}, 0, 0};
    int left = log2n;
    while (left >= 4 && left != 5) { p.r[p.n++] = 16; left -= 4; }
    while (left >= 3) { p.r[p.n++] = 8; left -= 3; }
    if (left == 2) { p.r[p.n++] = 4; left = 0; }
    if (left == 1) { p.r[p.n++] = 2; left = 0; }
    int ns = 1, off = 0;
    for (int i = 0; i < p.n; ++i) {
        p.ns[i] = ns;
        p.off[i] = off;
        if (ns > 1) off += (p.r[i] - 1) * ns;     // table[(r-1)*ns + k] = exp(-2*pi*i*k*r/(ns*R)), r = 1..R-1
        ns *= p.r[i];
    }
    p.total = off;
    return p;
}
// Passes whose twiddle period is short read every row of their table (it stays in
// L1); longer periods read only the power-of-two rows, coalesced, and form the
// other powers as products (at most three factors).
constexpr int kTableMaxPeriod = 64;

// Threads per workgroup for an N-point transform: 16 points per thread, at
// least one wavefront.
constexpr int threads_for(int n) { return n / 16 < 64 ? 64 : n / 16; }

// LDS image: one float2 of padding after every 16, so element i lives at
// i + i/16.  Pass-1 writes (stride = radix) and every later access pattern are
// then conflict free for ds_write_b64 / ds_read_b64 lane groups.
constexpr int lds_elems(int n) { return n + n / 16; }
FK_HD int phys(int i) { return i + (i >> 4); }

#if defined(__HIPCC__)

// One Stockham pass.  src(i) yields element i of the pass input (LDS or
// global), dst(i, v) consumes element i of the pass output.  When the input is
// the LDS image that dst overwrites, SYNC_AFTER_READ separates the phases.
//   tw: this pass's table, tw[(r-1)*NS + k] = exp(-2*pi*i*k*r/(NS*R))
template <int N, int NT, int R, int NS, bool INV, bool SYNC_AFTER_READ, class Src, class Dst>
FK_D void stockham_pass(Src&& src, Dst&& dst, const float2* __restrict__ tw, int tid) {
    constexpr int NB = N / R;                       // butterflies in this pass
    constexpr int CNT = (NB + NT - 1) / NT;         // per thread
    constexpr bool GUARD = (NB % NT) != 0;
    float2 v[CNT][R];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int j = tid + c * NT;
        if (!GUARD || j < NB) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[c][r] = src(j + r * NB);
        }
    }
    if constexpr (SYNC_AFTER_READ) __syncthreads();
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int j = tid + c * NT;
        if (!GUARD || j < NB) {
            const int k = j & (NS - 1);
            if constexpr (NS > 1) {
                float2 w[R];                                  // w[r] = exp(-2*pi*i*k*r/(NS*R))
                if constexpr (NS <= kTableMaxPeriod) {
#pragma unroll
                    for (int r = 1; r < R; ++r) w[r] = tw[(r - 1) * NS + k];
                } else {
#pragma unroll
                    for (int r = 1; r < R; r *= 2) w[r] = tw[(r - 1) * NS + k];
#pragma unroll
                    for (int r = 3; r < R; ++r) {
                        if ((r & (r - 1)) != 0) {             // not a power of two: top bit * remainder
                            int top = 1;
                            while (top * 2 <= r) top *= 2;
                            w[r] = cmul(w[top], w[r - top]);
                        }
                    }
                }
#pragma unroll
                for (int r = 1; r < R; ++r) v[c][r] = INV ? cmulc(v[c][r], w[r]) : cmul(v[c][r], w[r]);
            }
            dft<R, INV>(v[c]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) dst(j0 + r * NS, v[c][r]);
        }
    }
}

// All passes of an N-point transform.  The first pass reads through `first`
// (any source), middle passes run in place in the padded LDS image `s`, the
// last pass writes through `last` (any sink).  FIRST_IN_PLACE / LAST_IN_PLACE say
// that `first` reads / `last` writes the same LDS image (a barrier then splits
// the pass).  Callers put a __syncthreads() after this if `last` wrote LDS that
// other threads will read.
template <int LOG2N, bool INV, bool FIRST_IN_PLACE, bool LAST_IN_PLACE, class First, class Last>
FK_D void fft_passes(float2* s, const float2* __restrict__ ptw, int tid, First&& first, Last&& last) {
    constexpr int N = 1 << LOG2N;
    constexpr int NT = threads_for(N);
    constexpr Plan pl = make_plan(LOG2N);
    auto lds_src = [&](int i) { return s[phys(i)]; };
    auto lds_dst = [&](int i, float2 v) { s[phys(i)] = v; };
    static_assert(pl.n >= 2 && pl.n <= 4, "plan");
    constexpr int R0 = pl.r[0], R1 = pl.r[1], R2 = pl.r[2], R3 = pl.r[3];
    stockham_pass<N, NT, R0, 1, INV, FIRST_IN_PLACE>(first, lds_dst, ptw, tid);
    __syncthreads();
    if constexpr (pl.n == 2) {
        stockham_pass<N, NT, R1, R0, INV, LAST_IN_PLACE>(lds_src, last, ptw + pl.off[1], tid);
    } else if constexpr (pl.n == 3) {
        stockham_pass<N, NT, R1, R0, INV, true>(lds_src, lds_dst, ptw + pl.off[1], tid);
        __syncthreads();
        stockham_pass<N, NT, R2, R0 * R1, INV, LAST_IN_PLACE>(lds_src, last, ptw + pl.off[2], tid);
    } else {
        stockham_pass<N, NT, R1, R0, INV, true>(lds_src, lds_dst, ptw + pl.off[1], tid);
        __syncthreads();
        stockham_pass<N, NT, R2, R0 * R1, INV, true>(lds_src, lds_dst, ptw + pl.off[2], tid);
        __syncthreads();
        stockham_pass<N, NT, R3, R0 * R1 * R2, INV, LAST_IN_PLACE>(lds_src, last, ptw + pl.off[3], tid);
    }
}

#endif  // __HIPCC__

}  // namespace fk
