/* oracle/oracle_fft.c — TEST INFRASTRUCTURE ONLY. See oracle_fft.h. */
#include "oracle_fft.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

oc_fft_plan *oc_fft_plan_create(int n) {
    if (n < 4 || (n & (n - 1))) return NULL;
    oc_fft_plan *p = (oc_fft_plan *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->n = n;
    p->h = n / 2;
    p->log2h = 0;
    while ((1 << p->log2h) < p->h) p->log2h++;
    const int h = p->h;
    /* stage s (half-size m = 2^s) uses twiddles exp(-2 pi i k / (2m)), k<m,
     * stored contiguously at offset m-1 (sum of previous stage sizes). */
    p->stage_tw = (oc_cpx *)malloc(sizeof(oc_cpx) * (size_t)(h > 1 ? h : 1));
    p->split_tw = (oc_cpx *)malloc(sizeof(oc_cpx) * (size_t)(h / 2 + 1));
    p->bitrev = (int *)malloc(sizeof(int) * (size_t)h);
    if (!p->stage_tw || !p->split_tw || !p->bitrev) { oc_fft_plan_destroy(p); return NULL; }
    for (int m = 1; m < h; m *= 2) {
        for (int k = 0; k < m; ++k) {
            const double a = -M_PI * (double)k / (double)m;
            p->stage_tw[m - 1 + k].re = (float)cos(a);
            p->stage_tw[m - 1 + k].im = (float)sin(a);
        }
    }
    for (int k = 0; k <= h / 2; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)n;
        p->split_tw[k].re = (float)cos(a);
        p->split_tw[k].im = (float)sin(a);
    }
    for (int i = 0; i < h; ++i) {
        int r = 0;
        for (int b = 0; b < p->log2h; ++b) if (i & (1 << b)) r |= 1 << (p->log2h - 1 - b);
        p->bitrev[i] = r;
    }
    return p;
}

void oc_fft_plan_destroy(oc_fft_plan *p) {
    if (!p) return;
    free(p->stage_tw);
    free(p->split_tw);
    free(p->bitrev);
    free(p);
}

/* In-place radix-2 decimation-in-time complex FFT of length h on bit-reversed
 * input; sign = -1 forward, +1 inverse (conjugated twiddles). */
static void cfft_inplace(const oc_fft_plan *p, oc_cpx *a, int inverse) {
    const int h = p->h;
    for (int m = 1; m < h; m *= 2) {
        const oc_cpx *tw = p->stage_tw + (m - 1);
        for (int g = 0; g < h; g += 2 * m) {
            oc_cpx *lo = a + g, *hi = a + g + m;
            if (!inverse) {
                for (int k = 0; k < m; ++k) {
                    const float wr = tw[k].re, wi = tw[k].im;
                    const float tr = hi[k].re * wr - hi[k].im * wi;
                    const float ti = hi[k].re * wi + hi[k].im * wr;
                    hi[k].re = lo[k].re - tr; hi[k].im = lo[k].im - ti;
                    lo[k].re += tr;           lo[k].im += ti;
                }
            } else {
                for (int k = 0; k < m; ++k) {
                    const float wr = tw[k].re, wi = -tw[k].im;
                    const float tr = hi[k].re * wr - hi[k].im * wi;
                    const float ti = hi[k].re * wi + hi[k].im * wr;
                    hi[k].re = lo[k].re - tr; hi[k].im = lo[k].im - ti;
                    lo[k].re += tr;           lo[k].im += ti;
                }
            }
        }
    }
}

void oc_fft_r2c(const oc_fft_plan *p, const float *in, oc_cpx *out, oc_cpx *work) {
    const int h = p->h;
    /* z[m] = x[2m] + i x[2m+1], loaded in bit-reversed order */
    for (int m = 0; m < h; ++m) {
        const int r = p->bitrev[m];
        work[r].re = in[2 * m];
        work[r].im = in[2 * m + 1];
    }
    cfft_inplace(p, work, 0);
    /* X[k] = E[k] + W_N^k O[k];  E = (Z[k]+conj Z[h-k])/2, O = (Z[k]-conj Z[h-k])/(2i) */
    out[0].re = work[0].re + work[0].im; out[0].im = 0.0f;
    out[h].re = work[0].re - work[0].im; out[h].im = 0.0f;
    for (int k = 1; k <= h / 2; ++k) {
        const oc_cpx a = work[k], b = work[h - k];
        const float er = 0.5f * (a.re + b.re), ei = 0.5f * (a.im - b.im);
        const float or_ = 0.5f * (a.im + b.im), oi = -0.5f * (a.re - b.re);
        const float wr = p->split_tw[k].re, wi = p->split_tw[k].im;
        const float tr = or_ * wr - oi * wi, ti = or_ * wi + oi * wr;
        out[k].re = er + tr;      out[k].im = ei + ti;
        out[h - k].re = er - tr;  out[h - k].im = -(ei - ti);
    }
}

void oc_fft_c2r(const oc_fft_plan *p, const oc_cpx *in, float *out, oc_cpx *work) {
    const int h = p->h;
    /* Z[k] = E[k] + i O[k];  E = X[k]+conj X[h-k], O = (X[k]-conj X[h-k]) W_N^{-k}
     * then z = IDFT_h(Z) (unnormalised) gives x[2m] = Re z, x[2m+1] = Im z. */
    {
        const int r0 = p->bitrev[0];
        work[r0].re = in[0].re + in[h].re;
        work[r0].im = in[0].re - in[h].re;
    }
    for (int k = 1; k <= h / 2; ++k) {
        const oc_cpx a = in[k], b = in[h - k];
        const float er = a.re + b.re, ei = a.im - b.im;
        const float dr = a.re - b.re, di = a.im + b.im;
        const float wr = p->split_tw[k].re, wi = -p->split_tw[k].im; /* W^{-k} */
        const float or_ = dr * wr - di * wi, oi = dr * wi + di * wr;
        const int rk = p->bitrev[k], rh = p->bitrev[h - k];
        /* Z[k] = E + iO ; Z[h-k] = conj(E) + i conj(O) */
        work[rk].re = er - oi;  work[rk].im = ei + or_;
        work[rh].re = er + oi;  work[rh].im = -ei + or_;
    }
    cfft_inplace(p, work, 1);
    for (int m = 0; m < h; ++m) {
        out[2 * m] = work[m].re;
        out[2 * m + 1] = work[m].im;
    }
}
