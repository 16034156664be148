/* folve_host.h — C view of the C++ host layer (headers under folve_amd/csrc/host).
 *
 * The host layer restates folve's SoundProcessor, ProcessorPool and config
 * loader as C++ classes (namespace folve) above the engine ABI of
 * folve_engine.h; a C++ caller — folve itself — uses those classes directly
 * (INTEGRATION.md).  This header exposes the same objects to C / ctypes so that
 * the parity tests can drive them exactly the way ConvolveFileHandler does.
 * Each function names the reference member it forwards to.
 */
#ifndef FOLVE_HOST_H
#define FOLVE_HOST_H

#include "folve_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fh_processor fh_processor;   /* folve::SoundProcessor */
typedef struct fh_pool fh_pool;             /* folve::ProcessorPool  */

/* zita-sstring.cc:32 sstring() */
int fh_sstring(const char *srce, char *dest, int size);

/* zita-config.cc:282 config(): parse `config_file` into a new, uncommitted filter on
 * `engine` (NULL: assemble on the host only).  Returns config()'s status; on
 * return *filter is the filter (or NULL) and, if non-NULL, fragm/ninp/nout/size
 * receive the ZitaConfig fields. */
int fh_config_load(fe_engine *engine, const char *config_file, int fsamp, int channels,
                   fe_filter **filter, int *fragm, int *ninp, int *nout, int *size);

/* sound-processor.cc:34 SoundProcessor::Create (GPU chosen by the router); NULL on failure */
fh_processor *fh_processor_create(const char *config_file, int samplerate, int channels);
void fh_processor_destroy(fh_processor *p);
/* FillBuffer (cc:76): reads min(frames_available, block - input_pos) frames from src; returns frames taken — never
 * more than it returns, whatever the run-ahead setting (the processor is shown a source that ends with the block), so a
 * caller may advance `src` by the return value.  This form therefore runs one block per engine call. */
int fh_processor_fill_buffer(fh_processor *p, const float *src, int frames_available);
/* The same, also telling how many frames of `src` the processor took: with run-ahead on it reads AHEAD of the
 * frames it returns (the source is a file with a position, as SNDFILE* is), so a caller that passes spans of one
 * array must start the next span `*consumed` frames further, not `return value` frames (folve_amd/host.py does). */
int fh_processor_fill_buffer2(fh_processor *p, const float *src, int frames_available, int *consumed);
/* WriteProcessed (cc:86): processes if needed, copies sample_count frames to dst */
void fh_processor_write_processed(fh_processor *p, float *dst, int sample_count);
/* The same two calls over callbacks with libsndfile's contract — read(user, dst, frames) as sf_readf_float,
 * write(user, src, frames) as sf_writef_float — for hosts whose "file" is longer than one span: with run-ahead
 * on, FillBuffer asks the source for many blocks at once (sound_processor.h). */
typedef int (*fh_read_fn)(void *user, float *dst, int frames);
typedef int (*fh_write_fn)(void *user, const float *src, int frames);
int fh_processor_fill_buffer_from(fh_processor *p, fh_read_fn read, void *user);
void fh_processor_write_processed_to(fh_processor *p, fh_write_fn write, void *user, int sample_count);
/* Run-ahead depth in blocks for processors created from now on (1 = off; 0 = automatic, the default unless FOLVE_AMD_RUN_AHEAD is set:
 * 64 blocks of 8192 frames, as many frames for shorter blocks, fewer blocks for streams of many channels);
 * fh_processor_run_ahead: the depth a given processor was created with. */
void fh_device_peaks_set(int on);       /* run-ahead blocks' maxima from the GPU (1, default) or scanned on the caller's thread (0) */
void fh_run_ahead_set(int blocks);
int fh_run_ahead_get(void);
int fh_processor_run_ahead(const fh_processor *p);
int fh_processor_is_input_buffer_complete(const fh_processor *p);
int fh_processor_pending_writes(const fh_processor *p);
int fh_processor_input_channels(const fh_processor *p);
int fh_processor_output_channels(const fh_processor *p);
int fh_processor_block_size(const fh_processor *p);
float fh_processor_max_output_value(const fh_processor *p);
float fh_processor_max_abs_output_value(const fh_processor *p);
void fh_processor_reset_max_values(fh_processor *p);
void fh_processor_reset(fh_processor *p);
const char *fh_processor_config_file(const fh_processor *p);
long long fh_processor_config_file_timestamp(const fh_processor *p);
int fh_processor_config_still_up_to_date(const fh_processor *p);
int fh_processor_device(const fh_processor *p);
fe_stream *fh_processor_stream(const fh_processor *p);
fe_engine *fh_processor_engine(const fh_processor *p);
int fh_processor_ok(const fh_processor *p);            /* 0 once the processor has emitted silence: its GPU failed and no other could take the
                                                          stream (the pool discards it) */
int fh_processor_moves(const fh_processor *p);         /* times the processor's stream has moved to another GPU after its own failed: the
                                                          processor keeps the input of its last K + run-ahead blocks, replays the K blocks of
                                                          state on the new GPU and re-runs the failed call there (sound_processor.h) */
void fh_survival_set(int on);                          /* keep that input history (default 1; 0: a GPU failure ends a file in silence, as until
                                                          round 4); applies to processors created afterwards */

/* processor-pool.cc:33 ProcessorPool(max_per_config) */
fh_pool *fh_pool_create(int max_per_config);
void fh_pool_destroy(fh_pool *pool);
/* GetOrCreate (cc:48): NULL on failure with the reference's message in errmsg */
fh_processor *fh_pool_get_or_create(fh_pool *pool, const char *base_dir, int sampling_rate, int channels,
                                    int bits, char *errmsg, int errmsg_size);
/* Return (cc:93) */
void fh_pool_return(fh_pool *pool, fh_processor *p);
int fh_pool_pooled_count(fh_pool *pool, const char *config_path);

/* the per-GPU combiner (folve_amd/csrc/host/batch_scheduler.h): Process() calls of many file threads
 * that meet on a busy GPU leave as one launch; a lone call is never delayed.  On by default.
 * window_us is ignored (there is no collection window); max_batch < 0 keeps the current value. */
void fh_batching_set(int enabled, int window_us, int max_batch);
int fh_batching_enabled(void);
void fh_batching_early_quarters(int quarters);   /* BatchScheduler::SetEarlyQuarters (experiments) */
/* one block through the combiner of `engine` (what SoundProcessor::Process calls); returns the engine's status for the block */
int fh_batcher_process(fe_engine *engine, fe_stream *s, const float *in, int valid_frames, float *out);
/* totals over all GPUs since process start */
void fh_batching_stats(long long *requests, long long *batches, long long *largest);
/* the same plus: blocks carried by the requests, and batches submitted while another was still on the GPU */
void fh_batching_stats2(long long *requests, long long *blocks, long long *batches, long long *largest, long long *overlapped);

/* NUMA placement (folve_amd/csrc/host/numa_placement.h): with it on, a processor's page-locked ring is allocated
 * next to the GPU the router picked; fh_pin_thread_near_device moves the calling thread there too (1 = moved).
 * Off by default (FOLVE_AMD_NUMA=1 turns it on). */
void fh_numa_placement_set(int on);
int fh_pin_thread_near_device(int device);

/* the process-wide GPU sharder */
int fh_router_device_count(void);
int fh_router_live_streams(int slot);
int fh_router_cached_filters(void);                     /* committed filters held: one per (configuration, slot) in use */
/* GPU health (folve_amd/csrc/host/device_router.h): 0 healthy, 1 suspect (a call failed there: new files prefer the other
 * GPUs), 2 fenced (consecutive failures: no new files until a probe succeeds).  The reference's analogue is the pool's
 * discard-and-recreate loop, processor-pool.cc:71-77; its fallback when no processor can be had is the unfiltered
 * file, folve-filesystem.cc:78-88 — which one bad GPU of eight must not trigger. */
int fh_router_slot_state(int slot);
long long fh_router_slot_failures(int slot);            /* failures reported for the slot since the process started */
fe_engine *fh_router_slot_engine(int slot);             /* NULL until the slot has been used */
/* fence_after: consecutive failures that fence a slot (<= 0: leave); reprobe_seconds: how old the last look at a sick
 * slot must be before an open probes it again (< 0: leave; 0: every open) */
void fh_router_health_policy(int fence_after, double reprobe_seconds);
void fh_router_report_failure(fe_engine *e);            /* what a processor reports when a call on `e` failed (tests) */

#ifdef __cplusplus
}
#endif
#endif /* FOLVE_HOST_H */
