/* oracle/oracle_config.h — TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * Restatement of folve's jconvolver-format filter loader and of the block
 * state machine around the engine:
 *   config()      /root/reference/zita-config.cc:282-378
 *   readfile()    /root/reference/zita-config.cc:55-177   (/impulse/read)
 *   impdirac()    /root/reference/zita-config.cc:180-209  (/impulse/dirac)
 *   imphilbert()  /root/reference/zita-config.cc:212-259  (/impulse/hilbert)
 *   impcopy()     /root/reference/zita-config.cc:262-279  (/impulse/copy)
 *   convnew()     /root/reference/zita-fconfig.cc:38-97   (/convolver/new)
 *   sstring()     /root/reference/zita-sstring.cc:32-116 (spec zita-sstring.h:26-43)
 *   Audiofile     /root/reference/zita-audiofile.cc:51-99,170-182 (libsndfile
 *                 is absent: restated as a RIFF/WAVE PCM/float reader with
 *                 libsndfile's float normalisation)
 *   SoundProcessor /root/reference/sound-processor.cc:34-145
 */
#ifndef ORACLE_CONFIG_H
#define ORACLE_CONFIG_H

#include "oracle_convproc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* zita-config.h:51 */
enum { OC_NOERR, OC_ERR_OTHER, OC_ERR_SYNTAX, OC_ERR_PARAM, OC_ERR_ALLOC,
       OC_ERR_CANTCD, OC_ERR_COMMAND, OC_ERR_NOCONV, OC_ERR_IONUM };
#define OC_MAXSIZE 0x00100000   /* zita-config.h:61 */

/* zita-config.h:37-49 */
typedef struct oc_zita_config {
    const char *config_file;
    oc_convproc *convproc;
    int latency, options, fsamp, fragm, ninp, nout, size;
} oc_zita_config;

int oc_sstring(const char *srce, char *dest, int size);
int oc_fragm_for_size(unsigned int size);            /* zita-fconfig.cc:74-77 */
int oc_config(oc_zita_config *cfg, const char *config_file);

/* minimal WAV reader with libsndfile's sf_readf_float normalisation */
typedef struct oc_wav { int rate, chan; unsigned int frames; float *data; } oc_wav;
int oc_wav_load(const char *path, oc_wav *w);        /* 0 ok */
void oc_wav_free(oc_wav *w);

/* SoundProcessor restatement with float spans in place of SNDFILE*. */
typedef struct oc_sound_processor oc_sound_processor;
oc_sound_processor *oc_sp_create(const char *config_file, int samplerate, int channels);
oc_sound_processor *oc_sp_wrap(oc_convproc *conv, int fragm, int ninp, int nout); /* takes ownership */
void oc_sp_delete(oc_sound_processor *sp);
int oc_sp_fill_buffer(oc_sound_processor *sp, const float *src, int frames_available);
void oc_sp_write_processed(oc_sound_processor *sp, float *dst, int sample_count);
int oc_sp_is_input_buffer_complete(const oc_sound_processor *sp);
int oc_sp_pending_writes(const oc_sound_processor *sp);
void oc_sp_reset(oc_sound_processor *sp);
float oc_sp_max_output_value(const oc_sound_processor *sp);
int oc_sp_input_channels(const oc_sound_processor *sp);
int oc_sp_output_channels(const oc_sound_processor *sp);
int oc_sp_fragm(const oc_sound_processor *sp);
oc_convproc *oc_sp_convproc(oc_sound_processor *sp);
/* Drive a whole signal the way ConvolveFileHandler::AddMoreSoundData does
 * (convolve-file-handler.cc:370-424, non-gapless): fill, process, write r. */
long oc_sp_run(oc_sound_processor *sp, const float *in, long nframes, float *out);

/* CPU baseline driver: `nstreams` independent processors (one Convproc each,
 * folve's one-per-open-file model) fed `nblocks` full blocks each by `nthreads`
 * pthreads, streams round-robin over threads.  Returns wall seconds. */
double oc_bench_streams(int nstreams, int nblocks, int nthreads, int ninp, int nout,
                        int size, unsigned seed);

#ifdef __cplusplus
}
#endif
#endif
