// sound_processor.cpp — the block machine of folve's SoundProcessor on top of the GPU engine.
//
// This file restates the interface and the FillBuffer / WriteProcessed / Process state machine of
// folve's sound-processor.{h,cc}, Copyright (C) 2012 Henner Zeller <h.zeller@acm.org>, which is free
// software under the GNU General Public License, version 3 or (at your option) any later version;
// this restatement is distributed under the same terms, WITHOUT ANY WARRANTY.  See
// <http://www.gnu.org/licenses/>.  The arithmetic underneath (the engine, its kernels) and the
// run-ahead ring are original work.
#include "sound_processor.h"

#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <vector>

#include "batch_scheduler.h"
#include "device_router.h"
#include "numa_placement.h"
#include "../trace.h"

namespace folve {

static time_t GetModificationTime(const std::string& filename) {
    struct stat st;
    if (stat(filename.c_str(), &st) != 0) return 0;
    return st.st_mtime;
}

namespace {
std::atomic<int> g_run_ahead{-2};          // -2: not decided yet (environment); -1: automatic (by block size)
std::atomic<bool> g_device_peaks{true};
std::atomic<int> g_survive{-1};            // -1: not decided yet (FOLVE_AMD_SURVIVE, default on)
const int kMaxRunAhead = 1024;
// 64 blocks = 12 s of 44.1 kHz audio per chunk: 16 MB of page-locked ring (8 MB without the input history: in place) and 12.6 MB of delay line per open stereo file
// at 256 k taps.  Measured with 64 file threads on one MI355X: depth 8: 5.5, 32: 8.0, 64: 9.0, 128: 9.2 Gsamples/s.
const int kDefaultRunAhead = 64;
// What one processor may pin for its two chunks: streams of many channels run ahead by fewer blocks (a block of a
// 64-channel stream is 2 MB; 64 of them twice over would pin 256 MB per open file).  Stereo and 8-channel streams at
// the default depth stay below it (8 and 32 MB).
const size_t kRingBudgetBytes = static_cast<size_t>(64) << 20;
// The input history an open file keeps so that it can move to another GPU (2K + 2 blocks, pageable, touched only where a
// block would otherwise be overwritten while it still counts): up to what the reference's own limits give — MAXSIZE = 2^20
// taps (K = 128) on MAXINP = 64 channels (zita-config.h:61, zita-fconfig.cc:49-55): 258 blocks x 8192 frames x 64 channels.
// (It was the ring's 64 MB until round 6: a 16-channel K = 128 file fell back to silence on a GPU failure.)
const size_t kHistoryBudgetBytes = static_cast<size_t>(258) * 8192 * 64 * sizeof(float);
}  // namespace

void SoundProcessor::SetDevicePeaks(bool on) { g_device_peaks.store(on); }

void SoundProcessor::SetSurvival(bool on) { g_survive.store(on ? 1 : 0); }

static bool SurvivalWanted() {
    int v = g_survive.load();
    if (v < 0) {
        const char* env = getenv("FOLVE_AMD_SURVIVE");
        v = (env && atoi(env) == 0 && env[0] == '0') ? 0 : 1;
        g_survive.store(v);
    }
    return v != 0;
}

void SoundProcessor::SetRunAhead(int blocks) { g_run_ahead.store(blocks <= 0 ? -1 : std::min(blocks, kMaxRunAhead)); }

static int ConfiguredRunAhead() {              // blocks, or -1: automatic
    int v = g_run_ahead.load();
    if (v == -2) {
        const char* env = getenv("FOLVE_AMD_RUN_AHEAD");
        v = (env && atoi(env) > 0) ? std::min(atoi(env), kMaxRunAhead) : -1;
        g_run_ahead.store(v);
    }
    return v;
}

int SoundProcessor::RunAhead() {
    const int v = ConfiguredRunAhead();
    return v < 0 ? kDefaultRunAhead : v;
}

// Automatic: the default depth is 64 blocks OF 8192 FRAMES — as many frames per chunk for shorter blocks (a 128-tap
// filter has 128-frame blocks: 64 of them are 8192 frames, an engine call for 0.2 ms of audio).  One file thread through a
// 128-tap filter: 437 Msamples/s at 64 blocks, 2 700 at 1 024; 1 000 taps: 1 835 -> 4 450; 4 000 taps: 3 150 -> 4 580.
static int RunAheadForBlock(int fragm) {
    const int v = ConfiguredRunAhead();
    if (v >= 0) return v;
    const long long frames = static_cast<long long>(kDefaultRunAhead) * 8192;
    const long long blocks = fragm > 0 ? frames / fragm : kDefaultRunAhead;
    return static_cast<int>(std::max<long long>(kDefaultRunAhead, std::min<long long>(blocks, kMaxRunAhead)));
}

SoundProcessor* SoundProcessor::Create(const std::string& config_file, int samplerate, int channels) {
    // The GPU comes from the router; if that GPU fails under the open (its engine cannot transform the filter or hold
    // the stream), the open moves on to the next one: only a broken configuration, or no GPU left, fails it.
    DeviceRouter* router = DeviceRouter::Default();
    std::vector<fe_engine*> tried;
    for (;;) {
        fe_engine* engine = router->PickEngine(&tried);
        if (!engine) {
            Logf("No usable GPU: cannot create a processor for %s (there is no CPU fallback)", config_file.c_str());
            return NULL;
        }
        bool engine_fault = false;
        SoundProcessor* p = CreateOnReserved(engine, config_file, samplerate, channels, &engine_fault);
        if (p) return p;
        router->StreamClosed(engine);                           // give the reservation back
        if (!engine_fault) return NULL;                         // the configuration's fault: the same on every GPU
        router->ReportFailure(engine);
        tried.push_back(engine);
    }
}

SoundProcessor* SoundProcessor::CreateOn(fe_engine* engine, const std::string& config_file, int samplerate,
                                         int channels) {
    bool engine_fault = false;
    SoundProcessor* p = CreateOnReserved(engine, config_file, samplerate, channels, &engine_fault);
    if (p) DeviceRouter::Default()->StreamOpened(engine);
    else if (engine_fault) DeviceRouter::Default()->ReportFailure(engine);
    return p;
}

// The caller has already accounted the stream to `engine` in the router.
SoundProcessor* SoundProcessor::CreateOnReserved(fe_engine* engine, const std::string& config_file, int samplerate,
                                                 int channels, bool* engine_fault) {
    if (!engine) return NULL;
    ZitaConfig zita;
    memset(&zita, 0, sizeof(zita));
    // Parsing and the filter's FFTs happen once per (config, mtime, GPU); every
    // processor of that configuration shares the committed spectra.
    DeviceRouter::FileStamps impulse_files;
    fe_filter* filter = DeviceRouter::Default()->GetFilter(engine, config_file, GetModificationTime(config_file),
                                                           samplerate, channels, &zita, &impulse_files, engine_fault);
    if (!filter) return NULL;
    fe_stream* stream = NULL;
    // The stream's delay line is sized for the longest call this processor will make: its run-ahead depth
    // (one block per call is the reference's contract and what depth 1 gives).
    int run_depth = RunAheadForBlock(zita.fragm);
    // (read ONCE per processor: the run-ahead depth, the chunk layout and the history all follow from this one answer — a
    // SetSurvival() from another thread between two reads would size the arena for one layout and use the other)
    const bool keep_history = SurvivalWanted();
    {
        const size_t in = static_cast<size_t>(zita.fragm) * zita.ninp, out = static_cast<size_t>(zita.fragm) * zita.nout;
        const size_t block_bytes = (zita.ninp == zita.nout && !keep_history ? in : in + out) * sizeof(float);     // (ChunkFloats, below)
        const size_t fit = block_bytes ? kRingBudgetBytes / (2 * block_bytes) : static_cast<size_t>(run_depth);
        if (fit < static_cast<size_t>(run_depth)) run_depth = static_cast<int>(std::max<size_t>(fit, 1));
    }
    const int rc = fe_stream_open(filter, run_depth, &stream);
    fe_filter_release(filter);           // the stream holds its own reference
    if (rc != 0) {
        Logf("Cannot open a convolver stream for %s: %s", config_file.c_str(), fe_last_error());
        if (engine_fault) *engine_fault = true;                 // (device memory, the device itself)
        return NULL;
    }
    zita.engine = engine;
    if (NumaPlacement()) {
        // page-locked pages land where the allocating thread runs: next to the GPU that will read them
        ScopedDeviceAffinity near_gpu(fe_engine_device(engine));
        return new SoundProcessor(zita, config_file, stream, run_depth, impulse_files, samplerate, channels, keep_history);
    }
    return new SoundProcessor(zita, config_file, stream, run_depth, impulse_files, samplerate, channels, keep_history);
}

// The block buffer (`buffer_`, sound-processor.cc:62-63: fragm * max(ninp, nout) floats, reused in
// place for the output) lives in page-locked memory bound to the stream: the kernels read the PCM
// and write the result directly in it, so a block costs three kernel launches and one wait — no
// staging copies.  If page-locked memory cannot be had the buffer is ordinary memory and the engine
// stages it.
static float* AllocBlockBuffer(size_t floats, bool* pinned) {
    void* p = NULL;
    if (fe_host_alloc(floats * sizeof(float), &p) == 0 && p) {
        *pinned = true;
        return static_cast<float*>(p);
    }
    *pinned = false;
    return new float[floats];
}

// Floats of one run-ahead chunk: `depth` blocks of input and of output.  In place when the channel counts agree and no
// input history is kept (folve_engine.h: in-place calls of any length are fine then); with the history (the default) a
// chunk's input must outlive its call — it is what a move to another GPU re-runs and replays — so the output has its own
// half and nothing has to be copied aside.
static size_t ChunkFloats(const ZitaConfig& c, int depth, bool in_place) {
    if (depth <= 1) return 0;
    const size_t in = static_cast<size_t>(depth) * c.fragm * c.ninp, out = static_cast<size_t>(depth) * c.fragm * c.nout;
    return in_place ? in : in + out;
}

SoundProcessor::SoundProcessor(const ZitaConfig& config, const std::string& cfg, fe_stream* stream, int run_depth,
                               const std::vector<std::pair<std::string, time_t>>& impulse_files, int samplerate, int channels,
                               bool keep_history)
    : zita_config_(config), config_file_(cfg), config_file_timestamp_(GetModificationTime(cfg)), impulse_files_(impulse_files),
      samplerate_(samplerate), channels_(channels), engine_(config.engine),
      stream_(stream), hist_(NULL), hist_cap_(0), hist_k_(0), blocks_fed_(0), moves_(0),
      run_depth_(run_depth),
      in_place_(config.ninp == config.nout && !keep_history),
      buffer_floats_(static_cast<size_t>(config.fragm) * std::max(config.ninp, config.nout)),
      arena_floats_(buffer_floats_ + 2 * ChunkFloats(config, run_depth, config.ninp == config.nout && !keep_history)),
      buffer_(AllocBlockBuffer(arena_floats_, &buffer_pinned_)),
      cur_(NULL), ahead_(NULL), ring_block_(NULL), tail_(NULL), tail_frames_(0), source_short_(false), depth_next_(1),
      input_pos_(0), output_pos_(0), max_out_value_observed_(0.0), max_abs_value_observed_(0.0), ok_(true),
      slot_health_(DeviceRouter::Default()->HealthFlag(config.engine)) {
    if (buffer_pinned_ && fe_stream_bind_host_buffer(stream_, buffer_, arena_floats_ * sizeof(float)) != 0) {
        Logf("Processor %p: block buffer not bound (%s): blocks will be staged", static_cast<void*>(this), fe_last_error());
    }
    if (run_depth_ > 1) {
        // [ block buffer | chunk 0 | chunk 1 ] in one page-locked allocation bound to the stream
        float* p = buffer_ + buffer_floats_;
        const size_t in = static_cast<size_t>(run_depth_) * config.fragm * config.ninp;
        for (Chunk& c : chunks_) {
            c.in = p;
            c.out = in_place_ ? p : p + in;
            p += ChunkFloats(config, run_depth_, in_place_);
        }
        tail_ = new float[static_cast<size_t>(config.fragm) * config.ninp];
        for (Chunk& c : chunks_) c.peaks = new float[2 * static_cast<size_t>(run_depth_)];
    }
    // The input history an open file needs to move to another GPU (sound_processor.h): the K blocks in front of a call and
    // the call's own input.  A run-ahead chunk's input stays where it is (its output has its own half of the chunk), so the
    // ring only ever takes what would otherwise be overwritten while still needed: single blocks (computed in place), chunks
    // shorter than K (the ramp, a file's end) and the tail of a long chunk whose buffer is about to be re-used early.
    if (keep_history) {
        hist_k_ = config.fragm > 0 ? static_cast<int>((static_cast<long long>(config.size) + config.fragm - 1) / config.fragm) : 0;
        const long long cap = 2LL * hist_k_ + 2;
        const size_t bytes = static_cast<size_t>(cap) * config.fragm * config.ninp * sizeof(float);
        if (hist_k_ > 0 && bytes <= kHistoryBudgetBytes) {
            hist_ = new (std::nothrow) float[bytes / sizeof(float)];
            if (hist_) {
                hist_cap_ = static_cast<int>(cap);
                hist_tag_.assign(static_cast<size_t>(cap), -1);
            }
        }
        if (!hist_) Logf("Processor %p: no input history kept (%zu bytes): a GPU failure under it ends in silence", static_cast<void*>(this), bytes);
    }
    Reset();
}

SoundProcessor::~SoundProcessor() {
    DrainRing();                         // a request still on the GPU writes into the ring
    fe_stream_close(stream_);
    DeviceRouter::Default()->StreamClosed(engine_);
    delete[] hist_;
    if (buffer_pinned_) fe_host_free(buffer_);
    else delete[] buffer_;
    delete[] tail_;
    for (Chunk& c : chunks_) delete[] c.peaks;
}

int SoundProcessor::device() const { return fe_engine_device(engine_); }

// What the GPU sharder hears of this processor's engine calls: every failure (at the call sites), and a success only
// while the slot is not healthy (one relaxed load otherwise).
void SoundProcessor::EngineCallSucceeded() {
    if (slot_health_ && slot_health_->load(std::memory_order_relaxed) != 0 && ok_)
        DeviceRouter::Default()->ReportSuccess(engine_);
}

// Ask the source for the next chunk (depth_next_ whole blocks).  Whole blocks stay in the chunk; frames beyond
// the last whole block — the file's short last block — are moved to tail_.  A short read stops further
// read-ahead until everything read so far has been handed out.
bool SoundProcessor::ReadChunk(FrameSource* in, Chunk* c) {
    const int P = zita_config_.fragm;
    if (c->holds) {
        // This buffer still holds the input of an earlier chunk.  Whatever of it lies among the last K blocks handed to the
        // engine is state a later call may have to replay: it goes to the ring before the buffer is re-used.  (Long chunks
        // alternating — the steady state — find nothing here: the K most recent blocks are all in the OTHER buffer.)
        const long long lo = std::max(c->first, blocks_fed_ - hist_k_), hi = c->first + c->blocks;
        if (hist_ && lo < hi && c->blocks >= hist_k_)             // (a shorter chunk was saved when it was submitted)
            SaveBlocks(c->in + static_cast<size_t>(lo - c->first) * P * input_channels(), lo, static_cast<int>(hi - lo), P);
        c->holds = false;
    }
    const int want = depth_next_ * P;
    const int got = in->ReadFrames(c->in, want);
    c->blocks = got / P;
    c->next = 0;
    const int rem = got - c->blocks * P;
    if (rem > 0) {
        memcpy(tail_, c->in + static_cast<size_t>(c->blocks) * P * input_channels(), sizeof(float) * rem * input_channels());
        tail_frames_ = rem;
    }
    if (got < want) source_short_ = true;
    depth_next_ = std::min(depth_next_ * 2, run_depth_);
    return c->blocks > 0;
}

// One engine request for the chunk's blocks.  Through the combiner the call returns at once and the chunk is
// collected by SettleChunk; with the combiner off the engine is called synchronously here.
void SoundProcessor::SubmitChunk(Chunk* c) {
    const long long frames = static_cast<long long>(c->blocks) * zita_config_.fragm;
    c->peaks_valid = false;
    c->first = blocks_fed_;
    c->holds = hist_ != NULL;                                // (its input stays in place: the output has its own half of the chunk)
    if (hist_ && c->blocks < hist_k_) SaveBlocks(c->in, c->first, c->blocks, zita_config_.fragm);   // short: the next chunks may overwrite it while it still counts
    blocks_fed_ += c->blocks;
    if (ftrace::events_on()) ftrace::event("submit processor=%p gpu=%d first_block=%lld blocks=%d", static_cast<void*>(this), device(), c->first, c->blocks);
    if (BatchScheduler::Enabled()) {
        c->request = BatchScheduler::ForEngine(engine_)->Submit(stream_, c->in, frames, c->out,
                                                                g_device_peaks.load() ? c->peaks : NULL);
        return;
    }
    c->request = NULL;
    const int rc = fe_stream_process_blocks(stream_, c->in, frames, c->out);
    if (rc != 0) {
        Logf("GPU convolution failed (%d): %s", rc, fe_last_error());
        DeviceRouter::Default()->ReportFailure(engine_);
        if (!MoveToAnotherGpu(c->first, c->blocks, frames, c->out)) {
            memset(c->out, 0, sizeof(float) * frames * output_channels());
            ok_ = false;
        }
    } else {
        EngineCallSucceeded();
    }
}

void SoundProcessor::SettleChunk(Chunk* c) {
    if (!c->request) return;
    std::string error;
    const int rc = BatchScheduler::ForEngine(engine_)->Wait(static_cast<BatchScheduler::Request*>(c->request), &error,
                                                            &c->peaks_valid);
    c->request = NULL;
    if (ftrace::events_on()) ftrace::event("settle processor=%p gpu=%d first_block=%lld blocks=%d rc=%d", static_cast<void*>(this), device(), c->first, c->blocks, rc);
    if (rc != 0) {
        Logf("GPU convolution failed (%d): %s", rc, error.c_str());
        DeviceRouter::Default()->ReportFailure(engine_);
        c->peaks_valid = false;                              // (whatever the failed call left: the blocks are scanned when handed out)
        const long long frames = static_cast<long long>(c->blocks) * zita_config_.fragm;
        if (!MoveToAnotherGpu(c->first, c->blocks, frames, c->out)) {
            memset(c->out, 0, sizeof(float) * static_cast<size_t>(frames) * output_channels());
            ok_ = false;
        }
    } else {
        EngineCallSucceeded();
    }
}

// `n` blocks of input (block numbers b0 ..; the last one `last_frames` long, padded with the zeros the engine assumes behind
// it) into the history ring: block b in slot b % hist_cap_, tagged with its number.
void SoundProcessor::SaveBlocks(const float* in, long long b0, int n, int last_frames) {
    const size_t bf = static_cast<size_t>(zita_config_.fragm) * input_channels();
    for (int b = 0; b < n; ++b) {
        const size_t slot = static_cast<size_t>((b0 + b) % hist_cap_);
        float* dst = hist_ + slot * bf;
        const size_t k = (b == n - 1 ? static_cast<size_t>(last_frames) : static_cast<size_t>(zita_config_.fragm)) * input_channels();
        memcpy(dst, in + static_cast<size_t>(b) * bf, k * sizeof(float));
        if (k < bf) memset(dst + k, 0, (bf - k) * sizeof(float));
        hist_tag_[slot] = b0 + b;
    }
}

// A single block about to be computed in place (Process()): its input goes to the ring.
void SoundProcessor::KeepInput(const float* in, int blocks, int last_frames) {
    if (hist_) SaveBlocks(in, blocks_fed_, blocks, last_frames);
    blocks_fed_ += blocks;
}

// Where block b's input is now: in a chunk buffer that still holds it, or in the ring; NULL if it is gone.
const float* SoundProcessor::BlockInput(long long b) const {
    const size_t bf = static_cast<size_t>(zita_config_.fragm) * input_channels();
    for (const Chunk& c : chunks_)
        if (c.holds && b >= c.first && b < c.first + c.blocks) return c.in + static_cast<size_t>(b - c.first) * bf;
    if (hist_ && b >= 0) {
        const size_t slot = static_cast<size_t>(b % hist_cap_);
        if (hist_tag_[slot] == b) return hist_ + slot * bf;
    }
    return NULL;
}

// See the head of sound_processor.h.  The failed call's blocks are the newest `blocks` handed to the engine, the state in
// front of them the hist_k_ blocks before.
bool SoundProcessor::MoveToAnotherGpu(long long first, int blocks, long long frames, float* out) {
    if (!hist_ || blocks <= 0 || first + blocks != blocks_fed_) return false;
    // The other GPU's filter is looked up under this processor's (configuration, mtime): if the configuration or one of its
    // impulse files has been edited since the file was opened, that lookup would parse the NEW content — other taps in the
    // middle of a file, cached under the old key.  The taps this stream runs on exist nowhere but on the failed GPU: no move.
    if (!ConfigStillUpToDate()) {
        Logf("Processor %p: %s (or an impulse file of it) changed since this file was opened: the stream cannot move to another GPU",
             static_cast<void*>(this), config_file_.c_str());
        return false;
    }
    DeviceRouter* router = DeviceRouter::Default();
    const int P = zita_config_.fragm;
    const size_t bf = static_cast<size_t>(P) * input_channels();
    const long long replay0 = std::max<long long>(0, first - hist_k_);
    const long long nrep = first - replay0;
    // the kept blocks in order, contiguous: [replay | the failed call's own input]  (a copy: the call's output may share
    // memory with its input — Process() computes in place)
    std::vector<float> in(static_cast<size_t>(nrep + blocks) * bf);
    for (long long b = replay0; b < first + blocks; ++b) {
        const float* src = BlockInput(b);
        if (!src) {
            Logf("Processor %p: block %lld of its input history is gone: the stream cannot move", static_cast<void*>(this), b);
            return false;
        }
        memcpy(in.data() + static_cast<size_t>(b - replay0) * bf, src, bf * sizeof(float));
    }
    std::vector<float> scratch(static_cast<size_t>(std::max<long long>(nrep, 1)) * P * output_channels());
    std::vector<fe_engine*> tried(1, engine_);
    for (;;) {
        fe_engine* e = router->PickEngine(&tried);          // (reserves a stream there)
        if (!e) {
            Logf("Processor %p: its GPU failed and no other can take the stream: silence from here", static_cast<void*>(this));
            return false;
        }
        bool engine_fault = false;
        ZitaConfig z;
        memset(&z, 0, sizeof(z));
        fe_filter* filter = router->GetFilter(e, config_file_, config_file_timestamp_, samplerate_, channels_, &z, NULL, &engine_fault);
        fe_stream* ns = NULL;
        bool same = filter && z.fragm == zita_config_.fragm && z.ninp == zita_config_.ninp && z.nout == zita_config_.nout &&
                    z.size == zita_config_.size;
        int rc = same ? fe_stream_open(filter, run_depth_, &ns) : -1;
        if (filter) fe_filter_release(filter);
        if (same && rc != 0) engine_fault = true;
        if (rc == 0 && nrep > 0) rc = fe_stream_process_blocks(ns, in.data(), nrep * P, scratch.data());     // the delay line, rebuilt
        if (rc == 0) rc = fe_stream_process_blocks(ns, in.data() + static_cast<size_t>(nrep) * bf, frames, out);
        if (rc == 0) {
            fe_stream_close(stream_);
            router->StreamClosed(engine_);
            Logf("Processor %p (%s): GPU %d failed under it; the stream moved to GPU %d (%lld blocks of state replayed, %d re-run)",
                 static_cast<void*>(this), config_file_.c_str(), fe_engine_device(engine_), fe_engine_device(e), nrep, blocks);
            if (ftrace::events_on()) ftrace::event("move processor=%p gpu=%d->%d replayed=%lld rerun=%d", static_cast<void*>(this), fe_engine_device(engine_), fe_engine_device(e), nrep, blocks);
            engine_ = e;
            stream_ = ns;
            slot_health_ = router->HealthFlag(e);
            if (buffer_pinned_ && fe_stream_bind_host_buffer(stream_, buffer_, arena_floats_ * sizeof(float)) != 0)
                Logf("Processor %p: block buffer not bound on the new GPU (%s): blocks will be staged", static_cast<void*>(this), fe_last_error());
            ++moves_;
            return true;
        }
        if (ns) fe_stream_close(ns);
        router->StreamClosed(e);                             // give the reservation back
        if (!same && !engine_fault) {
            Logf("Processor %p: %s no longer yields the filter this stream was opened with: the stream cannot move", static_cast<void*>(this),
                 config_file_.c_str());
            return false;
        }
        Logf("Processor %p: GPU %d could not take the stream either (%s)", static_cast<void*>(this), fe_engine_device(e), fe_last_error());
        router->ReportFailure(e);
        tried.push_back(e);
    }
}

// Forget everything read ahead (Reset, destruction); a request still on the GPU is waited for first.
void SoundProcessor::DrainRing() {
    for (Chunk& c : chunks_) {
        SettleChunk(&c);
        c.blocks = c.next = 0;
        c.holds = false;
    }
    cur_ = ahead_ = NULL;
    ring_block_ = NULL;
    tail_frames_ = 0;
    source_short_ = false;
    depth_next_ = 1;
}

int SoundProcessor::FillBuffer(FrameSource* in) {
    const int P = zita_config_.fragm;
    const int samples_needed = P - input_pos_;
    assert(samples_needed);   // Otherwise, call WriteProcessed() first.
    output_pos_ = -1;
    ring_block_ = NULL;
    if (run_depth_ > 1 && input_pos_ == 0) {
        // A fresh block: it comes out of the run-ahead ring if a whole block can be had.
        if (!(cur_ && cur_->next < cur_->blocks)) {
            cur_ = NULL;
            if (ahead_) {                                   // the chunk computed while the last one was handed out
                cur_ = ahead_;
                ahead_ = NULL;
            } else if (tail_frames_ == 0 && !source_short_) {
                Chunk* c = &chunks_[0];
                if (ReadChunk(in, c)) { SubmitChunk(c); cur_ = c; }
            }
            if (cur_) {
                SettleChunk(cur_);                          // (one request per stream at a time)
                // keep the GPU busy with the next chunk while this one is handed out block by block
                if (tail_frames_ == 0 && !source_short_) {
                    Chunk* c = cur_ == &chunks_[0] ? &chunks_[1] : &chunks_[0];
                    if (ReadChunk(in, c)) { SubmitChunk(c); ahead_ = c; }
                }
            }
        }
        if (cur_) {
            ring_block_ = cur_->out + static_cast<size_t>(cur_->next) * P * output_channels();
            cur_->next++;
            input_pos_ = P;
            return P;
        }
        // No whole block left: what remains of the read-ahead is the file's short last block.  It goes into the
        // block buffer unprocessed, exactly where the reference's FillBuffer would have put it.
        source_short_ = false;
        if (tail_frames_ > 0) {
            const int r = tail_frames_;
            memcpy(buffer_, tail_, sizeof(float) * r * input_channels());
            tail_frames_ = 0;
            input_pos_ = r;
            return r;
        }
    }
    const int r = in->ReadFrames(buffer_ + static_cast<size_t>(input_pos_) * input_channels(), samples_needed);
    input_pos_ += r;
    return r;
}

void SoundProcessor::WriteProcessed(FrameSink* out, int sample_count) {
    if (output_pos_ < 0) {
        Process();
    }
    assert(sample_count <= zita_config_.fragm - output_pos_);
    const float* block = ring_block_ ? ring_block_ : buffer_;
    out->WriteFrames(block + static_cast<size_t>(output_pos_) * output_channels(), sample_count);
    output_pos_ += sample_count;
    if (output_pos_ == zita_config_.fragm) {
        input_pos_ = 0;
    }
}

// The maximum over returned frames, signed as sound-processor.cc:120-123 compares, and the magnitude beside it.
void SoundProcessor::ScanPeaks(const float* v, size_t n) {
    float hi = max_out_value_observed_, mag = max_abs_value_observed_;
    for (size_t j = 0; j < n; ++j) {
        const float x = v[j];
        hi = x > hi ? x : hi;
        const float a = x < 0 ? -x : x;
        mag = a > mag ? a : mag;
    }
    max_out_value_observed_ = hi;
    max_abs_value_observed_ = mag;
}

// One block through the GPU.  The reference zero-fills the unread tail, splits
// the channels, runs Convproc::process(), re-interleaves input_pos_ frames and
// tracks the maximum (sound-processor.cc:98-127); K1/K2/K3 do the first four on
// the device — frames >= input_pos_ count as zero, input_pos_ frames come back —
// and the maximum is taken here over the returned frames, signed as cc:120-123 does.
void SoundProcessor::Process() {
    if (ring_block_) {
        // a block of the run-ahead ring: computed already (its chunk was settled when it became current), its maxima too
        if (cur_ && cur_->peaks_valid) {
            const float* pk = cur_->peaks + 2 * static_cast<size_t>(cur_->next - 1);
            if (pk[0] > max_out_value_observed_) max_out_value_observed_ = pk[0];
            if (pk[1] > max_abs_value_observed_) max_abs_value_observed_ = pk[1];
        } else if (ok_) {
            ScanPeaks(ring_block_, static_cast<size_t>(input_pos_) * output_channels());
        }
        output_pos_ = 0;
        return;
    }
    if (input_pos_ > 0) {
        // The call goes through the device's combiner: alone it runs at once; while another file's
        // block is in flight on this GPU it is parked and leaves with the next batch.
        std::string error;
        int rc;
        const long long first = blocks_fed_;
        KeepInput(buffer_, 1, input_pos_);                   // (the block is computed in place)
        if (BatchScheduler::Enabled()) {
            rc = BatchScheduler::ForEngine(engine_)->Process(stream_, buffer_, input_pos_, buffer_, &error);
        } else {
            rc = fe_stream_process(stream_, buffer_, input_pos_, buffer_, NULL, NULL);
            if (rc != 0) error = fe_last_error();
        }
        const size_t n = static_cast<size_t>(input_pos_) * output_channels();
        if (rc != 0) {
            Logf("GPU convolution failed (%d): %s", rc, error.c_str());
            DeviceRouter::Default()->ReportFailure(engine_);
            if (MoveToAnotherGpu(first, 1, input_pos_, buffer_)) {
                ScanPeaks(buffer_, n);
            } else {
                memset(buffer_, 0, sizeof(float) * n);
                ok_ = false;
            }
        } else {
            ScanPeaks(buffer_, n);
            EngineCallSucceeded();
        }
    }
    output_pos_ = 0;
}

// The reference compares the configuration file's timestamp and notes as a TODO that the *.wav files it mentions
// should be checked as well (sound-processor.cc:129-133): they are, here.
bool SoundProcessor::ConfigStillUpToDate() const {
    return config_file_timestamp_ == GetModificationTime(config_file_) && DeviceRouter::StampsCurrent(impulse_files_);
}

void SoundProcessor::ResetMaxValues() {
    max_out_value_observed_ = 0.0;
    max_abs_value_observed_ = 0.0;
    fe_stream_reset_peaks(stream_);
}

void SoundProcessor::Reset() {
    DrainRing();
    fe_stream_reset(stream_);
    blocks_fed_ = 0;                     // (a reset stream has no state: nothing before this point would be replayed)
    std::fill(hist_tag_.begin(), hist_tag_.end(), -1LL);
    input_pos_ = 0;
    output_pos_ = -1;
    ResetMaxValues();
}

}  // namespace folve
