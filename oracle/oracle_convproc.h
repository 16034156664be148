/* oracle/oracle_convproc.h — TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * CPU restatement of the convolution engine folve drives: libzita-convolver's
 * `Convproc`, pinned by the reference at "4.0.3 (also compatible with 3.1.0)"
 * (/root/reference/README.md:394, INSTALL.md:55), linked with -lzita-convolver
 * -lfftw3f (/root/reference/Makefile:14).  Its source is NOT under
 * /root/reference and is not installed in this image, so this file restates
 * the library's published algorithm — single-level uniformly partitioned
 * overlap-add FFT convolution, which is what folve's call
 *     configure(ninp, nout, size, fragm, fragm, fragm, dens)
 * (/root/reference/zita-fconfig.cc:80-81, quantum == minpart == maxpart)
 * reduces it to — and anchors parity on the reference's own call sites
 * (sound-processor.cc:98-127, zita-config.cc:55-279) and demo-filter fixtures.
 *
 * PARITY UNPINNED BY REFERENCE TESTS: the reference has no tests or golden
 * vectors.  The oracle is pinned instead against (1) the exact
 * float64 linear convolution that defines the path mathematically, (2) the
 * closed forms of /root/reference/demo-filters (echo: y = 0.7x[n]+0.3x[n-22050]),
 * (3) the one path TU that compiles here, zita-sstring.cc (oracle/_ref).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything in oracle/.  The product (folve_amd/) never links or calls it.
 */
#ifndef ORACLE_CONVPROC_H
#define ORACLE_CONVPROC_H

#include "oracle_fft.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Limits of zita-convolver's Convproc as used at zita-fconfig.cc:49,55,74-75. */
enum { OC_MAXINP = 64, OC_MAXOUT = 64, OC_MAXPART = 8192, OC_MAXQUANT = 8192, OC_MINPART = 64 };

/* Convproc error convention: 0 ok, negative Converror codes. folve tests != 0. */
enum { OC_OK = 0, OC_BAD_STATE = -1, OC_BAD_PARAM = -2, OC_MEM_ALLOC = -3 };

typedef struct oc_convproc oc_convproc;

oc_convproc *oc_convproc_new(void);
void oc_convproc_delete(oc_convproc *c);

/* Convproc::configure (v4 signature; zita-fconfig.cc:80). Only the single-level
 * case quantum == minpart == maxpart that folve uses is implemented. */
int oc_configure(oc_convproc *c, int ninp, int nout, int maxsize,
                 int quantum, int minpart, int maxpart, float density);
/* Convproc::impdata_create (zita-config.cc:163,203,252): ACCUMULATES
 * data[k*step], k in [0, ind1-ind0), at taps ind0..ind1-1 of path inp->out. */
int oc_impdata_create(oc_convproc *c, int inp, int out, int step,
                      const float *data, int ind0, int ind1);
/* Convproc::impdata_copy (zita-config.cc:274): (inp2,out2) shares (inp1,out1). */
int oc_impdata_copy(oc_convproc *c, int inp1, int out1, int inp2, int out2);
/* Planar block windows of `fragm` floats; NULL when unconfigured
 * (sound-processor.cc:45-46,107,117). */
float *oc_inpdata(oc_convproc *c, int ch);
float *oc_outdata(oc_convproc *c, int ch);
int oc_process(oc_convproc *c);      /* sound-processor.cc:113 */
int oc_reset(oc_convproc *c);        /* sound-processor.cc:140: zero all state */
int oc_start_process(oc_convproc *c, int abspri, int policy); /* cc:144 */
int oc_stop_process(oc_convproc *c);  /* cc:70 */
int oc_cleanup(oc_convproc *c);       /* cc:71 */

int oc_fragm(const oc_convproc *c);
int oc_npar(const oc_convproc *c);
/* number of populated partitions on path (inp,out), following links */
int oc_path_partitions(const oc_convproc *c, int inp, int out);

#ifdef __cplusplus
}
#endif
#endif
