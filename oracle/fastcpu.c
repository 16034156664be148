/* oracle/fastcpu.c — TEST / MEASUREMENT INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * A FAIRER CPU stand-in for bench.py's cpu_baseline leg.  The parity oracle (oracle_convproc.c + oracle_fft.c) is a
 * scalar radix-2 restatement written to be read; timed, it is a strawman for zita-convolver + FFTW (the reference's
 * engine, configured as /root/reference/zita-fconfig.cc:74-81 does: one level, partition = 8192).  This file is the same
 * ALGORITHM — uniformly partitioned overlap-add, per input one r2c FFT of [block | 0], per output K complex
 * multiply-accumulates over P + 1 bins per path and one c2r FFT, 0.5/P folded into the filter spectra — laid out for a
 * vectorising compiler: split (structure-of-arrays) complex data, a radix-4 Stockham autosort FFT whose inner loops are
 * unit-stride, and a multiply-accumulate loop that is four FMAs per bin.  Built -O3 -march=native on the box that times
 * it (AVX2 / AVX-512 as available).  It is NOT zita-convolver and is labelled so; tests/test_oracle_cpu.py checks it
 * against the scalar oracle.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    int h;                 /* complex length (power of two >= 16) */
    int nstage;            /* radix-4 stages (+ one radix-2 stage if log2 h is odd) */
    int radix2_last;
    float **tw;            /* per radix-4 stage: 6 arrays of n/4: w1r w1i w2r w2i w3r w3i */
    float *sr, *si;        /* split twiddles exp(-2 pi i k / 2h), k <= h/2 */
    float *ar, *ai, *br, *bi;   /* ping-pong work arrays, h each */
} fc_plan;

static void *xmalloc(size_t n) {
    void *p = NULL;
    if (posix_memalign(&p, 64, (n + 63) & ~(size_t)63)) return NULL;
    return p;
}

static fc_plan *fc_plan_create(int h) {
    fc_plan *p = (fc_plan *)calloc(1, sizeof(*p));
    p->h = h;
    int l = 0;
    while ((1 << l) < h) l++;
    p->nstage = l / 2;
    p->radix2_last = l & 1;
    p->tw = (float **)calloc((size_t)p->nstage, sizeof(float *));
    int n = h;
    for (int s = 0; s < p->nstage; ++s, n /= 4) {
        const int n1 = n / 4;
        float *t = (float *)xmalloc(sizeof(float) * 6 * (size_t)n1);
        for (int q = 0; q < n1; ++q) {
            for (int r = 1; r <= 3; ++r) {
                const double a = -2.0 * M_PI * (double)r * (double)q / (double)n;
                t[(2 * (r - 1)) * n1 + q] = (float)cos(a);
                t[(2 * (r - 1) + 1) * n1 + q] = (float)sin(a);
            }
        }
        p->tw[s] = t;
    }
    p->sr = (float *)xmalloc(sizeof(float) * (size_t)(h / 2 + 1));
    p->si = (float *)xmalloc(sizeof(float) * (size_t)(h / 2 + 1));
    for (int k = 0; k <= h / 2; ++k) {
        const double a = -M_PI * (double)k / (double)h;
        p->sr[k] = (float)cos(a);
        p->si[k] = (float)sin(a);
    }
    p->ar = (float *)xmalloc(sizeof(float) * (size_t)h);
    p->ai = (float *)xmalloc(sizeof(float) * (size_t)h);
    p->br = (float *)xmalloc(sizeof(float) * (size_t)h);
    p->bi = (float *)xmalloc(sizeof(float) * (size_t)h);
    return p;
}

static void fc_plan_destroy(fc_plan *p) {
    if (!p) return;
    for (int s = 0; s < p->nstage; ++s) free(p->tw[s]);
    free(p->tw); free(p->sr); free(p->si); free(p->ar); free(p->ai); free(p->br); free(p->bi);
    free(p);
}

/* One radix-4 Stockham stage: n = current transform length, s = stride (number of interleaved transforms).
 * y[q + s*(4p + i)] = W_n^(i p) * (DFT_4 of x[q + s*(p + j n/4)])_i.   sign = -1 forward, +1 inverse. */
static void stage4(int n, int s, const float *restrict xr, const float *restrict xi, float *restrict yr, float *restrict yi,
                   const float *restrict tw, float sign) {
    const int n1 = n / 4;
    const float *w1r = tw, *w1i = tw + n1, *w2r = tw + 2 * n1, *w2i = tw + 3 * n1, *w3r = tw + 4 * n1, *w3i = tw + 5 * n1;
    if (s == 1) {
        /* unit stride in p on the input side; the four outputs of a butterfly are adjacent */
        for (int p = 0; p < n1; ++p) {
            const float ar = xr[p], ai = xi[p], br = xr[p + n1], bi = xi[p + n1];
            const float cr = xr[p + 2 * n1], ci = xi[p + 2 * n1], dr = xr[p + 3 * n1], di = xi[p + 3 * n1];
            const float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
            const float bpdr = br + dr, bpdi = bi + di;
            /* j*(b - d) with j = sign*i:  forward (sign -1): -i*(b-d) = (bmd.im, -bmd.re) */
            const float bmdr = br - dr, bmdi = bi - di;
            const float jr = -sign * bmdi, ji = sign * bmdr;
            const float v1r = amcr + jr, v1i = amci + ji, v2r = apcr - bpdr, v2i = apci - bpdi, v3r = amcr - jr, v3i = amci - ji;
            /* twiddles: forward uses W = exp(-2 pi i ..) as stored; inverse its conjugate */
            const float u1i = (sign < 0) ? w1i[p] : -w1i[p], u2i = (sign < 0) ? w2i[p] : -w2i[p], u3i = (sign < 0) ? w3i[p] : -w3i[p];
            yr[4 * p + 0] = apcr + bpdr;                  yi[4 * p + 0] = apci + bpdi;
            yr[4 * p + 1] = v1r * w1r[p] - v1i * u1i;     yi[4 * p + 1] = v1r * u1i + v1i * w1r[p];
            yr[4 * p + 2] = v2r * w2r[p] - v2i * u2i;     yi[4 * p + 2] = v2r * u2i + v2i * w2r[p];
            yr[4 * p + 3] = v3r * w3r[p] - v3i * u3i;     yi[4 * p + 3] = v3r * u3i + v3i * w3r[p];
        }
        return;
    }
    for (int p = 0; p < n1; ++p) {
        const float c1 = w1r[p], c2 = w2r[p], c3 = w3r[p];
        const float s1 = (sign < 0) ? w1i[p] : -w1i[p], s2 = (sign < 0) ? w2i[p] : -w2i[p], s3 = (sign < 0) ? w3i[p] : -w3i[p];
        const float *restrict x0r = xr + (size_t)s * p, *restrict x0i = xi + (size_t)s * p;
        const float *restrict x1r = x0r + (size_t)s * n1, *restrict x1i = x0i + (size_t)s * n1;
        const float *restrict x2r = x1r + (size_t)s * n1, *restrict x2i = x1i + (size_t)s * n1;
        const float *restrict x3r = x2r + (size_t)s * n1, *restrict x3i = x2i + (size_t)s * n1;
        float *restrict y0r = yr + (size_t)s * 4 * p, *restrict y0i = yi + (size_t)s * 4 * p;
        float *restrict y1r = y0r + s, *restrict y1i = y0i + s, *restrict y2r = y1r + s, *restrict y2i = y1i + s;
        float *restrict y3r = y2r + s, *restrict y3i = y2i + s;
#pragma GCC ivdep
        for (int q = 0; q < s; ++q) {
            const float ar = x0r[q], ai = x0i[q], br = x1r[q], bi = x1i[q], cr = x2r[q], ci = x2i[q], dr = x3r[q], di = x3i[q];
            const float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
            const float bpdr = br + dr, bpdi = bi + di, bmdr = br - dr, bmdi = bi - di;
            const float jr = -sign * bmdi, ji = sign * bmdr;
            const float v1r = amcr + jr, v1i = amci + ji, v2r = apcr - bpdr, v2i = apci - bpdi, v3r = amcr - jr, v3i = amci - ji;
            y0r[q] = apcr + bpdr;            y0i[q] = apci + bpdi;
            y1r[q] = v1r * c1 - v1i * s1;    y1i[q] = v1r * s1 + v1i * c1;
            y2r[q] = v2r * c2 - v2i * s2;    y2i[q] = v2r * s2 + v2i * c2;
            y3r[q] = v3r * c3 - v3i * s3;    y3i[q] = v3r * s3 + v3i * c3;
        }
    }
}

/* the last stage when log2 h is odd: n = 2, s = h/2, no twiddles */
static void stage2(int s, const float *restrict xr, const float *restrict xi, float *restrict yr, float *restrict yi) {
#pragma GCC ivdep
    for (int q = 0; q < s; ++q) {
        const float ar = xr[q], ai = xi[q], br = xr[q + s], bi = xi[q + s];
        yr[q] = ar + br; yi[q] = ai + bi;
        yr[q + s] = ar - br; yi[q + s] = ai - bi;
    }
}

/* complex FFT of length h on (ar, ai); result returned through *outr, *outi (one of the two work pairs) */
static void cfft(fc_plan *p, float sign, float **outr, float **outi) {
    float *xr = p->ar, *xi = p->ai, *yr = p->br, *yi = p->bi;
    int n = p->h, s = 1;
    for (int st = 0; st < p->nstage; ++st) {
        stage4(n, s, xr, xi, yr, yi, p->tw[st], sign);
        float *t;
        t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
        n /= 4;
        s *= 4;
    }
    if (p->radix2_last) {
        stage2(s, xr, xi, yr, yi);
        float *t;
        t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
    }
    *outr = xr;
    *outi = xi;
}

/* r2c of x[0 .. 2h): X[k], k = 0 .. h, split arrays (unnormalised, FFTW's convention) */
static void fc_r2c(fc_plan *p, const float *restrict x, float *restrict Xr, float *restrict Xi) {
    const int h = p->h;
    for (int m = 0; m < h; ++m) { p->ar[m] = x[2 * m]; p->ai[m] = x[2 * m + 1]; }
    float *zr, *zi;
    cfft(p, -1.0f, &zr, &zi);
    Xr[0] = zr[0] + zi[0]; Xi[0] = 0.0f;
    Xr[h] = zr[0] - zi[0]; Xi[h] = 0.0f;
    for (int k = 1; k <= h / 2; ++k) {
        const float ar = zr[k], ai = zi[k], br = zr[h - k], bi = zi[h - k];
        const float er = 0.5f * (ar + br), ei = 0.5f * (ai - bi);
        const float orr = 0.5f * (ai + bi), oi = -0.5f * (ar - br);
        const float wr = p->sr[k], wi = p->si[k];
        const float tr = orr * wr - oi * wi, ti = orr * wi + oi * wr;
        Xr[k] = er + tr;      Xi[k] = ei + ti;
        Xr[h - k] = er - tr;  Xi[h - k] = -(ei - ti);
    }
}

/* c2r: x[0 .. 2h) from X[0 .. h] (unnormalised) */
static void fc_c2r(fc_plan *p, const float *restrict Xr, const float *restrict Xi, float *restrict x) {
    const int h = p->h;
    p->ar[0] = Xr[0] + Xr[h];
    p->ai[0] = Xr[0] - Xr[h];
    for (int k = 1; k <= h / 2; ++k) {
        const float ar = Xr[k], ai = Xi[k], br = Xr[h - k], bi = Xi[h - k];
        const float er = ar + br, ei = ai - bi, dr = ar - br, di = ai + bi;
        const float wr = p->sr[k], wi = -p->si[k];
        const float orr = dr * wr - di * wi, oi = dr * wi + di * wr;
        p->ar[k] = er - oi;      p->ai[k] = ei + orr;
        p->ar[h - k] = er + oi;  p->ai[h - k] = -ei + orr;
    }
    float *zr, *zi;
    cfft(p, 1.0f, &zr, &zi);
    for (int m = 0; m < h; ++m) { x[2 * m] = zr[m]; x[2 * m + 1] = zi[m]; }
}

/* ---- one convolver: nch diagonal paths of `size` taps, partition P ---------------------------------------- */
typedef struct {
    int nch, P, K, B, pt;      /* B: padded bins per spectrum */
    fc_plan *plan;
    float *Hr, *Hi;            /* [nch][K][B] */
    float *Xr, *Xi;            /* [nch][K][B] ring */
    float *accr, *acci;        /* [B] */
    float *tbuf;               /* 2P */
    float *overlap;            /* [nch][P] */
} fc_conv;

static fc_conv *fc_conv_create(int nch, int size, int P, const float *taps /* [size], shared by the channels */) {
    fc_conv *c = (fc_conv *)calloc(1, sizeof(*c));
    c->nch = nch; c->P = P; c->K = (size + P - 1) / P;
    c->B = (P + 1 + 15) & ~15;
    c->plan = fc_plan_create(P);
    const size_t rows = (size_t)nch * c->K * c->B;
    c->Hr = (float *)xmalloc(sizeof(float) * rows); c->Hi = (float *)xmalloc(sizeof(float) * rows);
    c->Xr = (float *)xmalloc(sizeof(float) * rows); c->Xi = (float *)xmalloc(sizeof(float) * rows);
    memset(c->Hr, 0, sizeof(float) * rows); memset(c->Hi, 0, sizeof(float) * rows);
    memset(c->Xr, 0, sizeof(float) * rows); memset(c->Xi, 0, sizeof(float) * rows);
    c->accr = (float *)xmalloc(sizeof(float) * (size_t)c->B); c->acci = (float *)xmalloc(sizeof(float) * (size_t)c->B);
    c->tbuf = (float *)xmalloc(sizeof(float) * 2 * (size_t)P);
    c->overlap = (float *)xmalloc(sizeof(float) * (size_t)nch * P);
    memset(c->overlap, 0, sizeof(float) * (size_t)nch * P);
    const float norm = 0.5f / (float)P;
    for (int k = 0; k < c->K; ++k) {
        memset(c->tbuf, 0, sizeof(float) * 2 * (size_t)P);
        for (int t = 0; t < P && k * P + t < size; ++t) c->tbuf[t] = taps[k * P + t] * norm;
        float *hr = c->Hr + (size_t)k * c->B, *hi = c->Hi + (size_t)k * c->B;
        fc_r2c(c->plan, c->tbuf, hr, hi);
        for (int ch = 1; ch < nch; ++ch) {
            memcpy(c->Hr + ((size_t)ch * c->K + k) * c->B, hr, sizeof(float) * (size_t)c->B);
            memcpy(c->Hi + ((size_t)ch * c->K + k) * c->B, hi, sizeof(float) * (size_t)c->B);
        }
    }
    return c;
}

static void fc_conv_destroy(fc_conv *c) {
    if (!c) return;
    fc_plan_destroy(c->plan);
    free(c->Hr); free(c->Hi); free(c->Xr); free(c->Xi); free(c->accr); free(c->acci); free(c->tbuf); free(c->overlap);
    free(c);
}

/* one block: in / out interleaved [P][nch] */
static void fc_conv_process(fc_conv *c, const float *in, float *out) {
    const int P = c->P, K = c->K, B = c->B, nch = c->nch;
    for (int ch = 0; ch < nch; ++ch) {
        for (int t = 0; t < P; ++t) c->tbuf[t] = in[(size_t)t * nch + ch];
        memset(c->tbuf + P, 0, sizeof(float) * (size_t)P);
        fc_r2c(c->plan, c->tbuf, c->Xr + ((size_t)ch * K + c->pt) * B, c->Xi + ((size_t)ch * K + c->pt) * B);
        float *restrict ar = c->accr, *restrict ai = c->acci;
        memset(ar, 0, sizeof(float) * (size_t)B);
        memset(ai, 0, sizeof(float) * (size_t)B);
        int slot = c->pt;
        for (int j = 0; j < K; ++j) {
            const float *restrict xr = c->Xr + ((size_t)ch * K + slot) * B, *restrict xi = c->Xi + ((size_t)ch * K + slot) * B;
            const float *restrict hr = c->Hr + ((size_t)ch * K + j) * B, *restrict hi = c->Hi + ((size_t)ch * K + j) * B;
#pragma GCC ivdep
            for (int b = 0; b < B; ++b) {
                ar[b] += xr[b] * hr[b] - xi[b] * hi[b];
                ai[b] += xr[b] * hi[b] + xi[b] * hr[b];
            }
            slot = slot == 0 ? K - 1 : slot - 1;
        }
        fc_c2r(c->plan, ar, ai, c->tbuf);
        float *ov = c->overlap + (size_t)ch * P;
        for (int t = 0; t < P; ++t) out[(size_t)t * nch + ch] = ov[t] + c->tbuf[t];
        memcpy(ov, c->tbuf + P, sizeof(float) * (size_t)P);
    }
    c->pt = (c->pt + 1) % K;
}

/* ---- C entry points (ctypes) ---------------------------------------------------------------------------- */
/* run `frames` interleaved frames (a multiple of P) of x through a fresh convolver: the check against the scalar oracle */
int fc_run(int nch, int size, int P, const float *taps, const float *x, long frames, float *y) {
    if (frames % P) return -1;
    fc_conv *c = fc_conv_create(nch, size, P, taps);
    for (long f = 0; f < frames; f += P) fc_conv_process(c, x + (size_t)f * nch, y + (size_t)f * nch);
    fc_conv_destroy(c);
    return 0;
}

typedef struct { int first, stride, nstreams, nblocks, nch, P; fc_conv **cv; const float *in; float *out; } fc_arg;
static void *fc_worker(void *vp) {
    fc_arg *a = (fc_arg *)vp;
    for (int b = 0; b < a->nblocks; ++b)
        for (int s = a->first; s < a->nstreams; s += a->stride) fc_conv_process(a->cv[s], a->in, a->out);
    return NULL;
}
static unsigned fc_lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s; }

/* seconds for nstreams convolvers (one per stream, as folve has one per open file) x nblocks blocks on nthreads threads */
double fc_bench_streams(int nstreams, int nblocks, int nthreads, int nch, int size, int P, unsigned seed) {
    float *h = (float *)malloc(sizeof(float) * (size_t)size);
    for (int i = 0; i < size; ++i) h[i] = ((float)(fc_lcg(&seed) >> 8) / 8388608.0f - 1.0f) / sqrtf((float)size);
    fc_conv **cv = (fc_conv **)calloc((size_t)nstreams, sizeof(*cv));
    for (int s = 0; s < nstreams; ++s) cv[s] = fc_conv_create(nch, size, P, h);
    float *in = (float *)malloc(sizeof(float) * (size_t)P * nch);
    for (int i = 0; i < P * nch; ++i) in[i] = (float)(fc_lcg(&seed) >> 8) / 8388608.0f - 1.0f;
    float *out = (float *)malloc(sizeof(float) * (size_t)P * nch * (size_t)nthreads);
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    fc_arg *args = (fc_arg *)calloc((size_t)nthreads, sizeof(fc_arg));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; ++t) {
        args[t] = (fc_arg){t, nthreads, nstreams, nblocks, nch, P, cv, in, out + (size_t)t * P * nch};
        pthread_create(&th[t], NULL, fc_worker, &args[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int s = 0; s < nstreams; ++s) fc_conv_destroy(cv[s]);
    free(cv); free(h); free(in); free(out); free(th); free(args);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
