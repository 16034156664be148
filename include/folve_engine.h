/* folve_engine.h — C ABI of the MI355X (gfx950) convolution engine for folve.
 *
 * This is the drop-in boundary underneath folve's SoundProcessor /
 * ProcessorPool.  The reference has no C ABI at this seam: its
 * `SoundProcessor` (sound-processor.h:28-85) drives the C++ `Convproc` object
 * of libzita-convolver directly.  Every entry point below therefore cites the
 * Convproc / SoundProcessor call it replaces.  Plain pointers and sizes only;
 * nothing throws across this boundary; every function returns 0 or a negative
 * FE_* code (folve only ever tests `!= 0`, zita-config.cc:163,203,252,274).
 *
 * Threading contract (the reference's): at most one thread is
 * inside a given stream at a time; different streams may be driven from
 * different threads; a committed filter is immutable and shareable.
 *
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails with FE_ERR_DEVICE.  A processing call that fails before its
 * kernels were enqueued leaves its streams where they were; after a failure of the
 * device itself (a kernel fault, a lost GPU) the convolver state of the streams in
 * that call is undefined: reset or close them (folve::ProcessorPool::Return
 * deletes such processors).
 */
#ifndef FOLVE_ENGINE_H
#define FOLVE_ENGINE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FE_ABI_VERSION 1

/* Limits of the engine being replaced, as used by zita-fconfig.cc:49,55,74-75
 * (Convproc::MAXINP/MAXOUT/MAXQUANT/MINPART) and zita-config.h:61 (MAXSIZE). */
#define FE_MAXINP 64
#define FE_MAXOUT 64
#define FE_MAXQUANT 8192
#define FE_MINPART 64
#define FE_MAXSIZE 0x00100000

enum {
    FE_OK = 0,
    FE_ERR_STATE = -1,   /* call not valid in this state (e.g. add after commit) */
    FE_ERR_PARAM = -2,   /* bad argument */
    FE_ERR_ALLOC = -3,   /* host or device allocation failed */
    FE_ERR_DEVICE = -4,  /* no usable HIP device / HIP runtime error */
    FE_ERR_BUSY = -5,
    FE_ERR_UNSUPPORTED = -6  /* the call's fast form does not apply to these arguments; use the general one */
};

/* flags for fe_batch_process */
enum {
    FE_HOST_PTRS = 0,    /* in/out are host pointers; the call returns when out is filled */
    FE_DEVICE_PTRS = 1,  /* in/out are device pointers on the engine's GPU */
    FE_ASYNC = 2         /* with FE_DEVICE_PTRS: enqueue only, do not wait */
};

typedef struct fe_engine fe_engine;   /* one GPU: HIP stream, tables, scratch */
typedef struct fe_filter fe_filter;   /* a configured convolver matrix (what config() builds) */
typedef struct fe_stream fe_stream;   /* per-file convolver state (what a pooled SoundProcessor owns) */

/* ---- engine ------------------------------------------------------------- */
int fe_device_count(void);
/* hip_stream: a hipStream_t to launch on, or NULL for a private stream. */
int fe_engine_create(int device, void *hip_stream, fe_engine **out);
void fe_engine_destroy(fe_engine *e);
int fe_engine_synchronize(fe_engine *e);
int fe_engine_device(const fe_engine *e);
/* Health check: one small kernel on the engine's stream and its result back on the host; 0 if the GPU answered.
 * The reference has no counterpart (a CPU does not go away); its unit of failure handling is the processor the pool
 * discards and re-creates (processor-pool.cc:71-77) — the host's GPU sharder asks this of a GPU whose calls failed
 * before it sends new files there again. */
int fe_engine_probe(fe_engine *e);
/* The host CPUs next to HIP device `device` as the kernel lists them ("0-31,128-159": sysfs local_cpulist of the
 * device's PCI function), for hosts that place file threads and page-locked buffers on the GPU's NUMA node.
 * FE_ERR_UNSUPPORTED if the system does not say. */
int fe_device_local_cpulist(int device, char *buf, size_t size);
const char *fe_last_error(void);      /* thread-local detail of the last failure */

/* ---- filter: replaces Convproc::configure / impdata_create / impdata_copy - */
/* fragm = MAXQUANT; while (fragm > MINPART && fragm >= 2*size) fragm /= 2
 * (zita-fconfig.cc:74-77). */
int fe_fragm_for_size(unsigned int maxsize);
/* Convproc::configure(ninp, nout, size, fragm, fragm, fragm, density)
 * (zita-fconfig.cc:80-81); the block size is derived as above. */
int fe_filter_create(fe_engine *e, int ninp, int nout, int maxsize, float density, fe_filter **out);
/* Convproc::impdata_create(inp, out, step, data, ind0, ind1) (zita-config.cc:163,
 * 203,252): ADDS data[k*step] at taps ind0+k; 0-based channels. */
int fe_filter_add(fe_filter *f, int inp, int out, int step, const float *data, int ind0, int ind1);
/* Convproc::impdata_copy(inp1, out1, inp2, out2) (zita-config.cc:274):
 * (inp2,out2) shares the spectra of (inp1,out1), later additions to the source
 * included (README.CONFIG.txt:91-97).  As in zita: returns 0 without linking when
 * the source pair has no data yet or the target already has data of its own, and
 * fe_filter_add on a linked target is a no-op. */
int fe_filter_link(fe_filter *f, int inp1, int out1, int inp2, int out2);
/* Transform the partitions on the GPU and make the filter immutable.  Until
 * then nothing touches the device (the loader is testable without a GPU). */
int fe_filter_commit(fe_filter *f);
/* Reference counting: create returns 1 reference; each open stream holds one. */
void fe_filter_retain(fe_filter *f);
void fe_filter_release(fe_filter *f);
int fe_filter_use_count(const fe_filter *f);      /* references held right now (a cache that sees 1 is the only holder) */
int fe_filter_inputs(const fe_filter *f);
int fe_filter_outputs(const fe_filter *f);
int fe_filter_block_size(const fe_filter *f);     /* P = fragm */
int fe_filter_partitions(const fe_filter *f);     /* K = ceil(size / P) */
int fe_filter_maxsize(const fe_filter *f);
/* populated partitions of path inp->out (links followed), 0 if no path */
int fe_filter_path_partitions(const fe_filter *f, int inp, int out);
/* copy the assembled float32 taps of path inp->out (links followed) into dst[0..n) */
int fe_filter_get_taps(const fe_filter *f, int inp, int out, float *dst, int n);

/* ---- stream: replaces Convproc's per-instance state ----------------------- */
/* max_blocks_per_call bounds how many consecutive blocks one call may carry
 * (FDL ring = K + max_blocks rows per input channel: the K + 1 rows of G reach K
 * blocks back); longer spans are split. */
int fe_stream_open(fe_filter *f, int max_blocks_per_call, fe_stream **out);
/* Page-locked host memory for block buffers (what SoundProcessor's `buffer_`, sound-processor.cc:62,
 * becomes), and its binding to a stream: host-pointer calls on that stream whose in/out lie inside
 * the bound range are read and written by the kernels directly over the bus — no staging copies.
 * The binding is a promise that the range stays allocated until it is unbound (buf = NULL) or the
 * stream is closed.  In-place calls (in == out) are fine for one block (what SoundProcessor::Process
 * does, sound-processor.cc:62-63,98-127) and for any length when ninp == nout. */
int fe_host_alloc(size_t bytes, void **out);
void fe_host_free(void *p);
int fe_stream_bind_host_buffer(fe_stream *s, void *buf, size_t bytes);
/* Convproc::reset() (sound-processor.cc:140): zero all state, zero latency. */
int fe_stream_reset(fe_stream *s);
void fe_stream_close(fe_stream *s);
/* Exactly SoundProcessor::Process() (sound-processor.cc:98-127) for one block:
 * `in` holds valid_frames (<= P) interleaved input frames, the rest of the
 * block is zero; `out` receives valid_frames interleaved output frames.  Peaks
 * are the running maxima since the last reset: signed (as cc:120-123 compares)
 * and absolute.  Host pointers; synchronous. */
int fe_stream_process(fe_stream *s, const float *in, int valid_frames, float *out,
                      float *peak_signed, float *peak_abs);
/* Run-ahead form: nframes interleaved frames = ceil(nframes/P) blocks, the last
 * zero-padded; time advances by whole blocks.  Host pointers; synchronous. */
int fe_stream_process_blocks(fe_stream *s, const float *in, long long nframes, float *out);
int fe_stream_get_peaks(fe_stream *s, float *peak_signed, float *peak_abs);
int fe_stream_reset_peaks(fe_stream *s);
long long fe_stream_blocks_done(const fe_stream *s);
int fe_stream_block_size(const fe_stream *s);     /* P of the stream's filter */
int fe_stream_max_blocks(const fe_stream *s);     /* the max_blocks_per_call it was opened with */

/* ---- batch: many independent streams in one launch ------------------------ */
/* streams[i] consumes nframes[i] interleaved frames from in[i] and produces as
 * many into out[i].  All streams must live on one engine; streams of different
 * filters are launched in groups.  flags: FE_HOST_PTRS or FE_DEVICE_PTRS[|FE_ASYNC].
 * Large host-pointer batches are pipelined (copy in / compute / copy out overlap); page-locked
 * buffers (hipHostMalloc, hipHostRegister) let both bus directions run at once. */
int fe_batch_process(fe_stream *const *streams, int n, const float *const *in, const long long *nframes,
                     float *const *out, int flags);

/* The same in two steps, for host buffers that lie in page-locked memory bound to their streams
 * (fe_stream_bind_host_buffer; otherwise FE_ERR_UNSUPPORTED and nothing is enqueued): submit enqueues the
 * kernels and returns at once with a ticket; fe_ticket_wait returns when the outputs are in the callers'
 * buffers, and consumes the ticket.  A host can keep one batch running while it assembles and submits the
 * next (folve_amd/csrc/host/batch_scheduler.cpp does).  Submitted batches go to one of two launch lanes (HIP
 * streams) and may overlap on the GPU; the calls of any ONE stream execute in the order they were made,
 * whichever lane they land on.  A stream may be in one submitted batch at a time, and a ticket must be
 * waited for before its streams are closed, reset or used in a synchronous call. */
typedef struct fe_ticket fe_ticket;
int fe_batch_submit(fe_stream *const *streams, int n, const float *const *in, const long long *nframes,
                    float *const *out, fe_ticket **ticket);
/* fe_batch_submit that also returns per-BLOCK maxima: block_peaks[i] (NULL: not wanted for stream i) receives, when the
 * ticket has been waited for, two floats for each of stream i's ceil(nframes[i] / P) blocks of this call — the maximum of
 * the block's output samples compared as sound-processor.cc:120-123 compares them (signed, never below 0) and the maximum
 * magnitude.  What folve::SoundProcessor feeds max_output_value() with, block by block, instead of rescanning its output. */
int fe_batch_submit_peaks(fe_stream *const *streams, int n, const float *const *in, const long long *nframes,
                          float *const *out, float *const *block_peaks, fe_ticket **ticket);
int fe_ticket_wait(fe_ticket *ticket);
/* 1 if fe_ticket_wait would return without waiting, 0 if the batch is still on the GPU, negative on a device
 * error; does not consume the ticket. */
int fe_ticket_done(fe_ticket *ticket);

/* Running peaks of n streams with one synchronisation (what the batcher hands back per block). */
int fe_batch_get_peaks(fe_stream *const *streams, int n, float *peak_signed, float *peak_abs);

/* ---- launch-shape overrides (tests, experiments) ---------------------------- */
/* The engine chooses kernel forms from the batch shape (walker run lengths so that every CU
 * has two workgroups, the MAC form from blocks per call, ..).  A test pins a form on ONE
 * engine to reach it with a small batch; 0 restores the automatic choice.  No process-wide
 * state, no environment variables. */
enum {
    FE_TUNE_FWD_RUN = 0,   /* K1 walker: consecutive blocks per workgroup (1..4096) */
    FE_TUNE_INV_RUN = 1,   /* K3 walker: consecutive blocks per workgroup */
    FE_TUNE_MAC_FORM = 2,  /* K2: 1 general, 4 / 8 / 16 sliding window of that many outputs, 100 whole-call walk (also for
                              sparse filters, whose empty rows of G are zeros in memory) */
    FE_TUNE_FFT_FORM = 3,  /* K1/K3: 1 general kernels only, 2 walkers whenever the shape allows, 3 no channel-pair walkers */
    FE_TUNE_FAIL_NEXT = 4, /* fault injection: the n-th launch round of this engine from now fails with FE_ERR_DEVICE (1 = the next;
                              negative: every round until the knob is set to 0) */
    FE_TUNE_WALK_LPB = 6,  /* K2 whole-call walk: lanes per bin (1, 2, 4; the filter's rows of G are spread over them) */
    FE_TUNE_WALK_TILES = 7,/* K2 whole-call walk: time tiles per call (1 ..) */
    FE_TUNE_DUPLEX_OUT = 8,/* big submitted batches (PCM comes in by DMA): results leave 0 / 2 by DMA too, 1 by K3's stores into the callers' buffers */
    FE_TUNE_DUPLEX_CHUNK_MB = 9, /* ... and the pipeline's chunk size: megabytes of PCM (in + out) per chunk, at most 16 chunks (0: 32) */
    FE_TUNE_DUPLEX_MIN_MB = 10, /* ... and the smallest batch (megabytes of PCM in + out) that takes the pipeline (0: 32) */
    FE_TUNE_DUPLEX_CAP_MB = 12, /* ... and the most device memory (megabytes per direction) the pipeline may stage a batch in (0: 8192); a
                              batch beyond it, or one the memory cannot be had for, runs with the zero-copy kernels instead of failing */
    FE_TUNE_SPLIT = 11,    /* a lone stream's long call as time tiles whose K1 -> K2 -> K3 chains alternate between the two launch lanes:
                              0 / 1 never (the default: measured slower than one chain), 2 .. 8 tiles; calls longer than the stream's
                              run-ahead depth (several launch rounds) are not split */
    FE_TUNE_WALK_FMA = 13, /* K2 whole-call walk: 3 = three multiply-adds per complex one (kernels/mac_walk3.hip), 4 = four, 0 = by shape */
    FE_TUNE_WALK_NT = 14,  /* K2 three-FMA walk on one lane per bin (any rung of its window ladder): its row loads and stores carry the non-temporal hint
                              0 = where the launch's rows of Y exceed 192 MB (they cannot stay in the 256 MB Infinity Cache), 1 = never, 2 = always */
    FE_TUNE_LANES = 5      /* fe_batch_submit: 1 = every batch on the engine's own HIP stream, 0 / 2 = two lanes (batches of different
                              streams overlap: one reads its PCM over the bus while the other writes its results back) */
};
int fe_engine_set_tuning(fe_engine *e, int knob, int value);
/* Device self-test of the cross-lane exchange the FFT rows use (kernels.h launch_xlane_selftest). */
int fe_debug_xlane(fe_engine *e, float *out512);

/* ---- measurement hooks (bench.py: per-kernel timing) ----------------------- */
enum { FE_K_FORWARD = 0, FE_K_MAC = 1, FE_K_INVERSE = 2, FE_K_COUNT = 3 };
/* on = 1: HIP events recorded between the three launches of every round, the round waited for (an event-to-event time
 *         holds the kernel AND the launch boundary behind it: 4 - 7 us on MI355X);
 * on = 2: a start and a stop event bound to each DISPATCH (hipExtLaunchKernelGGL): the command processor's begin-to-end
 *         of that packet alone, the figure rocprofv3's kernel trace prints; rounds keep running back to back (a ring of
 *         32 event sets, read back when a set comes round again or when the profile is asked for);
 * on = 0: off.  Streams on this engine keep their results either way. */
int fe_engine_set_profiling(fe_engine *e, int on);
/* accumulated since the last reset: launches and milliseconds per kernel (mode 1: event to event) */
int fe_engine_get_profile(fe_engine *e, long long launches[FE_K_COUNT], double ms[FE_K_COUNT]);
/* ... and of mode 2: each role's dispatches and the sum of their own durations */
int fe_engine_get_kernel_profile(fe_engine *e, long long launches[FE_K_COUNT], double ms[FE_K_COUNT]);
int fe_engine_reset_profile(fe_engine *e);
/* The kernels of the engine's most recent launch round, one string per role (FE_K_FORWARD / FE_K_MAC / FE_K_INVERSE): the
 * template instantiation as rocprofv3 prints it without namespaces and arguments, e.g. "mac_walk_kernel<33, 7, true, 4, 1, 1>".
 * names[k] must hold `cap` bytes each (96 suffice).  bench.py prints them and accepts a committed profile only for the same kernels. */
int fe_engine_last_kernels(fe_engine *e, char *forward, char *mac, char *inverse, size_t cap);
/* What this GPU's HBM delivers to plain streaming kernels (16 bytes per lane, `bytes` per pass, `reps`
 * passes, HIP events): gbs[0] reading, gbs[1] writing, gbs[2] copying (bytes read + written), in GB/s.
 * bench.py prints them beside the nominal 8 TB/s its roofline fraction divides by. */
int fe_engine_hbm_rates(fe_engine *e, size_t bytes, int reps, double gbs[3]);
/* The same plus two rates with every workgroup writing its OWN contiguous region of the buffer instead of one front of
 * consecutive kilobytes moving through it — the pattern of the engine's kernels, whose workgroups walk their rows:
 * gbs[3] writing, gbs[4] copying (bytes read + written).  On MI355X stores care (5.6 - 6.1 TB/s against 4.0 - 5.1), loads do
 * not (tools/micro/write_rate.hip). */
int fe_engine_hbm_rates2(fe_engine *e, size_t bytes, int reps, double gbs[5]);
/* ... plus gbs[5]: the best float4 copy shape found on MI355X (tools/micro/copy_rate.hip: non-temporal loads and stores, a 4 KiB
 * read burst per wave then its write burst; 5.9 - 6.0 TB/s counting both directions where the plain copies give 4.7 - 5.2;
 * /opt/skills/guides/MI355X_MICROARCH.md:36 quotes 6.29) — the copy yardstick of bench.py's roofline.measured_hbm. */
int fe_engine_hbm_rates3(fe_engine *e, size_t bytes, int reps, double gbs[6]);

#ifdef __cplusplus
}
#endif
#endif /* FOLVE_ENGINE_H */
