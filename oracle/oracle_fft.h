/* oracle/oracle_fft.h — TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * Plain-C float32 real FFT pair with FFTW's r2c/c2r conventions, standing in
 * for libfftw3f, which the reference links (Makefile:14 "-lzita-convolver
 * -lfftw3f") but which is absent from /root/reference and from this image.
 *   r2c: X[k] = sum_t x[t] exp(-2*pi*i*k*t/N),  k = 0..N/2   (unnormalised)
 *   c2r: x[t] = sum_k X[k] exp(+2*pi*i*k*t/N) over the Hermitian extension
 *        (unnormalised: c2r(r2c(x)) == N*x), exactly as fftwf_plan_dft_r2c_1d /
 *        fftwf_plan_dft_c2r_1d define them.
 */
#ifndef ORACLE_FFT_H
#define ORACLE_FFT_H

typedef struct { float re, im; } oc_cpx;

typedef struct oc_fft_plan {
    int n;            /* real length N (power of two, >= 4) */
    int h;            /* N/2 = complex FFT length            */
    int log2h;
    oc_cpx *stage_tw; /* per-stage contiguous twiddles for the length-h complex FFT */
    oc_cpx *split_tw; /* exp(-2*pi*i*k/N), k = 0..h/2 */
    int *bitrev;      /* bit reversal for length h */
} oc_fft_plan;

oc_fft_plan *oc_fft_plan_create(int n);
void oc_fft_plan_destroy(oc_fft_plan *p);
/* work: caller-provided scratch of h complex values. */
void oc_fft_r2c(const oc_fft_plan *p, const float *in, oc_cpx *out, oc_cpx *work);
void oc_fft_c2r(const oc_fft_plan *p, const oc_cpx *in, float *out, oc_cpx *work);

#endif
