/* oracle/oracle_convproc.c — TEST INFRASTRUCTURE ONLY. See oracle_convproc.h.
 *
 * Restated mechanism (zita-convolver 4.0.3, recollected; not in tree):
 *  - impdata: for every partition k overlapping [ind0,ind1): copy the taps,
 *    scaled by norm = 0.5/parsize (= 1/N, the c2r normalisation folded into
 *    H), into a zeroed 2*parsize buffer, r2c FFT it, and ADD the spectrum into
 *    the path's partition k (allocated on first touch).
 *  - process: per input r2c FFT of [block, 0..0] into ring slot `ptind`; per
 *    output acc = 0; for every path into it, for j < npar with a populated
 *    partition: acc += X[(ptind - j) mod npar] * H[j]; c2r; first half is added
 *    to the overlap saved by the previous block, second half is saved.
 */
#include "oracle_convproc.h"

#include <stdlib.h>
#include <string.h>

typedef struct oc_path {
    oc_cpx **fftb;          /* [npar] partition spectra (parsize+1), NULL = not populated */
    struct oc_path *link;   /* impdata_copy: use link->fftb instead */
    int used;               /* path exists (has data or a link) */
} oc_path;

struct oc_convproc {
    int configured, running;
    int ninp, nout, maxsize, parsize, npar;
    int ptind;
    oc_fft_plan *plan;
    oc_cpx *work;           /* parsize complex scratch for the FFT */
    float *time_data;       /* 2*parsize */
    oc_cpx *freq_data;      /* parsize+1 accumulator */
    oc_cpx *prep_freq;      /* parsize+1 for impdata */
    float **inpbuff;        /* [ninp][parsize] */
    float **outbuff;        /* [nout][parsize] */
    float **overlap;        /* [nout][parsize] */
    oc_cpx **ffta;          /* [ninp*npar] input spectra ring, each parsize+1 */
    oc_path *paths;         /* [ninp*nout] */
};

oc_convproc *oc_convproc_new(void) {
    return (oc_convproc *)calloc(1, sizeof(oc_convproc));
}

static void free_all(oc_convproc *c) {
    if (!c->configured) return;
    for (int i = 0; i < c->ninp * c->nout; ++i) {
        oc_path *p = &c->paths[i];
        if (p->fftb) {
            for (int k = 0; k < c->npar; ++k) free(p->fftb[k]);
            free(p->fftb);
        }
    }
    free(c->paths);
    for (int i = 0; i < c->ninp * c->npar; ++i) free(c->ffta[i]);
    free(c->ffta);
    for (int i = 0; i < c->ninp; ++i) free(c->inpbuff[i]);
    for (int i = 0; i < c->nout; ++i) { free(c->outbuff[i]); free(c->overlap[i]); }
    free(c->inpbuff); free(c->outbuff); free(c->overlap);
    free(c->time_data); free(c->freq_data); free(c->prep_freq); free(c->work);
    oc_fft_plan_destroy(c->plan);
    memset(c, 0, sizeof(*c));
}

void oc_convproc_delete(oc_convproc *c) {
    if (!c) return;
    free_all(c);
    free(c);
}

int oc_cleanup(oc_convproc *c) { free_all(c); return OC_OK; }

int oc_configure(oc_convproc *c, int ninp, int nout, int maxsize,
                 int quantum, int minpart, int maxpart, float density) {
    (void)density; /* a hint for multi-level partition sequences; irrelevant at one level */
    if (c->configured) return OC_BAD_STATE;
    if (ninp < 1 || ninp > OC_MAXINP || nout < 1 || nout > OC_MAXOUT) return OC_BAD_PARAM;
    if (quantum & (quantum - 1)) return OC_BAD_PARAM;
    if (quantum < OC_MINPART || quantum > OC_MAXQUANT) return OC_BAD_PARAM;
    if (minpart != quantum || maxpart != quantum) return OC_BAD_PARAM; /* single level only */
    if (maxsize < 1) return OC_BAD_PARAM;
    c->ninp = ninp; c->nout = nout; c->maxsize = maxsize;
    c->parsize = quantum;
    c->npar = (maxsize + quantum - 1) / quantum;
    c->plan = oc_fft_plan_create(2 * quantum);
    const size_t P = (size_t)quantum;
    c->work = (oc_cpx *)calloc(P, sizeof(oc_cpx));
    c->time_data = (float *)calloc(2 * P, sizeof(float));
    c->freq_data = (oc_cpx *)calloc(P + 1, sizeof(oc_cpx));
    c->prep_freq = (oc_cpx *)calloc(P + 1, sizeof(oc_cpx));
    c->inpbuff = (float **)calloc((size_t)ninp, sizeof(float *));
    c->outbuff = (float **)calloc((size_t)nout, sizeof(float *));
    c->overlap = (float **)calloc((size_t)nout, sizeof(float *));
    for (int i = 0; i < ninp; ++i) c->inpbuff[i] = (float *)calloc(P, sizeof(float));
    for (int i = 0; i < nout; ++i) {
        c->outbuff[i] = (float *)calloc(P, sizeof(float));
        c->overlap[i] = (float *)calloc(P, sizeof(float));
    }
    c->ffta = (oc_cpx **)calloc((size_t)ninp * (size_t)c->npar, sizeof(oc_cpx *));
    for (int i = 0; i < ninp * c->npar; ++i) c->ffta[i] = (oc_cpx *)calloc(P + 1, sizeof(oc_cpx));
    c->paths = (oc_path *)calloc((size_t)ninp * (size_t)nout, sizeof(oc_path));
    c->configured = 1;
    c->ptind = 0;
    return OC_OK;
}

static oc_path *path_of(const oc_convproc *c, int inp, int out) {
    return &c->paths[inp * c->nout + out];
}

static const oc_path *resolve(const oc_path *p) {
    int guard = 0;
    while (p->link && guard++ < 8192) p = p->link;
    return p;
}

int oc_impdata_create(oc_convproc *c, int inp, int out, int step,
                      const float *data, int ind0, int ind1) {
    if (!c->configured) return OC_BAD_STATE;
    if (inp < 0 || inp >= c->ninp || out < 0 || out >= c->nout) return OC_BAD_PARAM;
    if (ind0 < 0 || ind1 < ind0) return OC_BAD_PARAM;
    oc_path *p = path_of(c, inp, out);
    /* (zita, not in tree: Convlevel::impdata_write) a node that is a link takes no data of its
     * own: the call returns without effect. */
    if (p->link) return OC_OK;
    const int P = c->parsize;
    const float norm = 0.5f / (float)P;
    if (!p->fftb) {
        p->fftb = (oc_cpx **)calloc((size_t)c->npar, sizeof(oc_cpx *));
        if (!p->fftb) return OC_MEM_ALLOC;
    }
    path_of(c, inp, out)->used = 1;
    p->used = 1;
    int i0 = 0;
    for (int k = 0; k < c->npar; ++k, i0 += P) {
        const int i1 = i0 + P;
        if (i0 >= ind1 || i1 <= ind0) continue;
        if (!p->fftb[k]) {
            p->fftb[k] = (oc_cpx *)calloc((size_t)P + 1, sizeof(oc_cpx));
            if (!p->fftb[k]) return OC_MEM_ALLOC;
        }
        memset(c->time_data, 0, sizeof(float) * 2 * (size_t)P);
        const int j0 = i0 > ind0 ? i0 : ind0;
        const int j1 = i1 < ind1 ? i1 : ind1;
        for (int j = j0; j < j1; ++j) c->time_data[j - i0] = norm * data[(size_t)(j - ind0) * (size_t)step];
        oc_fft_r2c(c->plan, c->time_data, c->prep_freq, c->work);
        for (int b = 0; b <= P; ++b) {
            p->fftb[k][b].re += c->prep_freq[b].re;
            p->fftb[k][b].im += c->prep_freq[b].im;
        }
    }
    return OC_OK;
}

int oc_impdata_copy(oc_convproc *c, int inp1, int out1, int inp2, int out2) {
    if (!c->configured) return OC_BAD_STATE;
    if (inp1 < 0 || inp1 >= c->ninp || out1 < 0 || out1 >= c->nout) return OC_BAD_PARAM;
    if (inp2 < 0 || inp2 >= c->ninp || out2 < 0 || out2 >= c->nout) return OC_BAD_PARAM;
    if (inp1 == inp2 && out1 == out2) return OC_BAD_PARAM;
    oc_path *src = path_of(c, inp1, out1), *dst = path_of(c, inp2, out2);
    /* (zita, not in tree: Convlevel::impdata_copy)  M1 = findmacnode(inp1, out1, false);
     * if (!M1) return;  M2 = findmacnode(inp2, out2, true);  if (M2->_fftb) return;
     * M2->_link = M1;  — no source node yet, or a target that already has data: no effect. */
    if (!src->used) return OC_OK;
    if (dst->fftb) return OC_OK;
    for (const oc_path *at = src; at; at = at->link)
        if (at == dst) return OC_BAD_PARAM;       /* would form a cycle (undefined in zita) */
    dst->link = src;
    dst->used = 1;
    return OC_OK;
}

float *oc_inpdata(oc_convproc *c, int ch) {
    if (!c->configured || ch < 0 || ch >= c->ninp) return NULL;
    return c->inpbuff[ch];
}

float *oc_outdata(oc_convproc *c, int ch) {
    if (!c->configured || ch < 0 || ch >= c->nout) return NULL;
    return c->outbuff[ch];
}

int oc_process(oc_convproc *c) {
    if (!c->configured) return OC_BAD_STATE;
    const int P = c->parsize, K = c->npar;
    for (int i = 0; i < c->ninp; ++i) {
        memcpy(c->time_data, c->inpbuff[i], sizeof(float) * (size_t)P);
        memset(c->time_data + P, 0, sizeof(float) * (size_t)P);
        oc_fft_r2c(c->plan, c->time_data, c->ffta[i * K + c->ptind], c->work);
    }
    for (int o = 0; o < c->nout; ++o) {
        oc_cpx *acc = c->freq_data;
        memset(acc, 0, sizeof(oc_cpx) * ((size_t)P + 1));
        for (int i = 0; i < c->ninp; ++i) {
            const oc_path *p = path_of(c, i, o);
            if (!p->used) continue;
            p = resolve(p);
            if (!p->fftb) continue;
            int slot = c->ptind;
            for (int j = 0; j < K; ++j) {
                const oc_cpx *h = p->fftb[j];
                if (h) {
                    const oc_cpx *x = c->ffta[i * K + slot];
                    for (int b = 0; b <= P; ++b) {
                        acc[b].re += x[b].re * h[b].re - x[b].im * h[b].im;
                        acc[b].im += x[b].re * h[b].im + x[b].im * h[b].re;
                    }
                }
                if (slot == 0) slot = K;
                slot--;
            }
        }
        oc_fft_c2r(c->plan, acc, c->time_data, c->work);
        float *out = c->outbuff[o], *ov = c->overlap[o];
        for (int t = 0; t < P; ++t) out[t] = ov[t] + c->time_data[t];
        memcpy(ov, c->time_data + P, sizeof(float) * (size_t)P);
    }
    c->ptind = (c->ptind + 1) % K;
    return OC_OK;
}

int oc_reset(oc_convproc *c) {
    if (!c->configured) return OC_BAD_STATE;
    const size_t P = (size_t)c->parsize;
    for (int i = 0; i < c->ninp; ++i) memset(c->inpbuff[i], 0, sizeof(float) * P);
    for (int i = 0; i < c->nout; ++i) {
        memset(c->outbuff[i], 0, sizeof(float) * P);
        memset(c->overlap[i], 0, sizeof(float) * P);
    }
    for (int i = 0; i < c->ninp * c->npar; ++i) memset(c->ffta[i], 0, sizeof(oc_cpx) * (P + 1));
    c->ptind = 0;
    return OC_OK;
}

int oc_start_process(oc_convproc *c, int abspri, int policy) {
    (void)abspri; (void)policy; /* single level, synchronous: no worker threads to start */
    if (!c->configured) return OC_BAD_STATE;
    c->running = 1;
    return OC_OK;
}

int oc_stop_process(oc_convproc *c) { c->running = 0; return OC_OK; }

int oc_fragm(const oc_convproc *c) { return c->configured ? c->parsize : 0; }
int oc_npar(const oc_convproc *c) { return c->configured ? c->npar : 0; }

int oc_path_partitions(const oc_convproc *c, int inp, int out) {
    if (!c->configured || inp < 0 || inp >= c->ninp || out < 0 || out >= c->nout) return 0;
    const oc_path *p = path_of(c, inp, out);
    if (!p->used) return 0;
    p = resolve(p);
    if (!p->fftb) return 0;
    int n = 0;
    for (int k = 0; k < c->npar; ++k) n += p->fftb[k] != NULL;
    return n;
}
