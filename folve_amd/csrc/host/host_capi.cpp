// host_capi.cpp — extern "C" view of the host classes (include/folve_host.h).
#include "../../../include/folve_host.h"

#include <string.h>

#include <algorithm>
#include <string>

#include "batch_scheduler.h"
#include "device_router.h"
#include "numa_placement.h"
#include "processor_pool.h"
#include "sound_processor.h"
#include "sstring.h"
#include "zita_config.h"

using folve::ProcessorPool;
using folve::SoundProcessor;

namespace {

// FrameSource / FrameSink over caller-provided float spans.
class SpanSource : public folve::FrameSource {
public:
    SpanSource(const float* src, int frames, int channels) : src_(src), left_(frames), ch_(channels) {}
    int ReadFrames(float* dst, int frames) override {
        const int n = std::min(frames, left_);
        memcpy(dst, src_, sizeof(float) * static_cast<size_t>(n) * ch_);
        src_ += static_cast<size_t>(n) * ch_;
        left_ -= n;
        taken_ += n;
        return n;
    }
    int taken() const { return taken_; }
private:
    const float* src_;
    int left_, ch_, taken_ = 0;
};

class SpanSink : public folve::FrameSink {
public:
    SpanSink(float* dst, int channels) : dst_(dst), ch_(channels) {}
    int WriteFrames(const float* src, int frames) override {
        memcpy(dst_, src, sizeof(float) * static_cast<size_t>(frames) * ch_);
        dst_ += static_cast<size_t>(frames) * ch_;
        return frames;
    }
private:
    float* dst_;
    int ch_;
};

class CallbackSource : public folve::FrameSource {
public:
    CallbackSource(fh_read_fn fn, void* user) : fn_(fn), user_(user) {}
    int ReadFrames(float* dst, int frames) override { return fn_(user_, dst, frames); }
private:
    fh_read_fn fn_;
    void* user_;
};

class CallbackSink : public folve::FrameSink {
public:
    CallbackSink(fh_write_fn fn, void* user) : fn_(fn), user_(user) {}
    int WriteFrames(const float* src, int frames) override { return fn_(user_, src, frames); }
private:
    fh_write_fn fn_;
    void* user_;
};

inline SoundProcessor* SP(fh_processor* p) { return reinterpret_cast<SoundProcessor*>(p); }
inline const SoundProcessor* SP(const fh_processor* p) { return reinterpret_cast<const SoundProcessor*>(p); }
inline ProcessorPool* PP(fh_pool* p) { return reinterpret_cast<ProcessorPool*>(p); }

}  // namespace

extern "C" {

int fh_sstring(const char* srce, char* dest, int size) { return folve::sstring(srce, dest, size); }

int fh_config_load(fe_engine* engine, const char* config_file, int fsamp, int channels, fe_filter** filter,
                   int* fragm, int* ninp, int* nout, int* size) {
    folve::ZitaConfig z;
    memset(&z, 0, sizeof(z));
    z.engine = engine;
    z.fsamp = fsamp;
    z.ninp = channels;
    z.nout = channels;
    const int stat = folve::config(&z, config_file);
    if (filter) *filter = z.filter; else if (z.filter) fe_filter_release(z.filter);
    if (fragm) *fragm = z.fragm;
    if (ninp) *ninp = z.ninp;
    if (nout) *nout = z.nout;
    if (size) *size = z.size;
    return stat;
}

fh_processor* fh_processor_create(const char* config_file, int samplerate, int channels) {
    return reinterpret_cast<fh_processor*>(SoundProcessor::Create(config_file, samplerate, channels));
}
void fh_processor_destroy(fh_processor* p) { delete SP(p); }

// The span form keeps the reference's contract to the letter — it takes exactly the frames it returns — so the source
// it shows the processor ends where the current block ends: a processor with run-ahead then never reads beyond what
// it hands back (it sees a file that always ends at the block boundary, and falls back to one block per engine call).
// Hosts that want run-ahead use fh_processor_fill_buffer2 or the callback form.
int fh_processor_fill_buffer(fh_processor* p, const float* src, int frames_available) {
    SpanSource s(src, std::min(frames_available, SP(p)->frames_wanted()), SP(p)->input_channels());
    return SP(p)->FillBuffer(&s);
}
int fh_processor_fill_buffer2(fh_processor* p, const float* src, int frames_available, int* consumed) {
    SpanSource s(src, frames_available, SP(p)->input_channels());
    const int r = SP(p)->FillBuffer(&s);
    if (consumed) *consumed = s.taken();
    return r;
}
void fh_processor_write_processed(fh_processor* p, float* dst, int sample_count) {
    SpanSink s(dst, SP(p)->output_channels());
    SP(p)->WriteProcessed(&s, sample_count);
}
int fh_processor_fill_buffer_from(fh_processor* p, fh_read_fn read, void* user) {
    CallbackSource s(read, user);
    return SP(p)->FillBuffer(&s);
}
void fh_processor_write_processed_to(fh_processor* p, fh_write_fn write, void* user, int sample_count) {
    CallbackSink s(write, user);
    SP(p)->WriteProcessed(&s, sample_count);
}
void fh_run_ahead_set(int blocks) { SoundProcessor::SetRunAhead(blocks); }
void fh_device_peaks_set(int on) { SoundProcessor::SetDevicePeaks(on != 0); }
int fh_run_ahead_get(void) { return SoundProcessor::RunAhead(); }
int fh_processor_run_ahead(const fh_processor* p) { return SP(p)->run_ahead(); }
int fh_processor_is_input_buffer_complete(const fh_processor* p) { return SP(p)->is_input_buffer_complete(); }
int fh_processor_pending_writes(const fh_processor* p) { return SP(p)->pending_writes(); }
int fh_processor_input_channels(const fh_processor* p) { return SP(p)->input_channels(); }
int fh_processor_output_channels(const fh_processor* p) { return SP(p)->output_channels(); }
int fh_processor_block_size(const fh_processor* p) { return SP(p)->block_size(); }
float fh_processor_max_output_value(const fh_processor* p) { return SP(p)->max_output_value(); }
float fh_processor_max_abs_output_value(const fh_processor* p) { return SP(p)->max_abs_output_value(); }
void fh_processor_reset_max_values(fh_processor* p) { SP(p)->ResetMaxValues(); }
void fh_processor_reset(fh_processor* p) { SP(p)->Reset(); }
const char* fh_processor_config_file(const fh_processor* p) { return SP(p)->config_file().c_str(); }
long long fh_processor_config_file_timestamp(const fh_processor* p) { return SP(p)->config_file_timestamp(); }
int fh_processor_config_still_up_to_date(const fh_processor* p) { return SP(p)->ConfigStillUpToDate(); }
int fh_processor_device(const fh_processor* p) { return SP(p)->device(); }
fe_stream* fh_processor_stream(const fh_processor* p) { return SP(p)->stream(); }
fe_engine* fh_processor_engine(const fh_processor* p) { return SP(p)->engine(); }
int fh_processor_ok(const fh_processor* p) { return SP(p)->ok() ? 1 : 0; }
int fh_processor_moves(const fh_processor* p) { return SP(p)->moves(); }
void fh_survival_set(int on) { SoundProcessor::SetSurvival(on != 0); }

fh_pool* fh_pool_create(int max_per_config) { return reinterpret_cast<fh_pool*>(new ProcessorPool(max_per_config)); }
void fh_pool_destroy(fh_pool* pool) { delete PP(pool); }

fh_processor* fh_pool_get_or_create(fh_pool* pool, const char* base_dir, int sampling_rate, int channels, int bits,
                                    char* errmsg, int errmsg_size) {
    std::string err;
    SoundProcessor* p = PP(pool)->GetOrCreate(base_dir, sampling_rate, channels, bits, &err);
    if (errmsg && errmsg_size > 0) {
        strncpy(errmsg, err.c_str(), static_cast<size_t>(errmsg_size) - 1);
        errmsg[errmsg_size - 1] = 0;
    }
    return reinterpret_cast<fh_processor*>(p);
}
void fh_pool_return(fh_pool* pool, fh_processor* p) { PP(pool)->Return(SP(p)); }
int fh_pool_pooled_count(fh_pool* pool, const char* config_path) {
    return static_cast<int>(PP(pool)->pooled_count(config_path));
}

void fh_batching_set(int enabled, int window_us, int max_batch) {
    folve::BatchScheduler::SetEnabled(enabled != 0);
    (void)window_us;                       // the combiner has no collection window (kept in the signature)
    folve::BatchScheduler::Configure(max_batch);
}
int fh_batching_enabled(void) { return folve::BatchScheduler::Enabled(); }
void fh_batching_early_quarters(int quarters) { folve::BatchScheduler::SetEarlyQuarters(quarters); }
int fh_batcher_process(fe_engine* engine, fe_stream* s, const float* in, int valid_frames, float* out) {
    return folve::BatchScheduler::ForEngine(engine)->Process(s, in, valid_frames, out, NULL);
}
void fh_batching_stats(long long* requests, long long* batches, long long* largest) {
    fh_batching_stats2(requests, NULL, batches, largest, NULL);
}
void fh_batching_stats2(long long* requests, long long* blocks, long long* batches, long long* largest, long long* overlapped) {
    long long r = 0, b = 0, l = 0, k = 0, o = 0;
    folve::DeviceRouter* router = folve::DeviceRouter::Default();
    for (int d = 0; d < router->device_count(); ++d) {
        fe_engine* e = router->EngineIfCreated(d);
        if (!e) continue;
        const folve::BatchScheduler::Stats st = folve::BatchScheduler::ForEngine(e)->stats();
        r += st.requests; b += st.batches; if (st.largest > l) l = st.largest;
        k += st.blocks; o += st.overlapped;
    }
    if (blocks) *blocks = k;
    if (overlapped) *overlapped = o;
    if (requests) *requests = r;
    if (batches) *batches = b;
    if (largest) *largest = l;
}

void fh_numa_placement_set(int on) { folve::SetNumaPlacement(on != 0); }
int fh_pin_thread_near_device(int device) { return folve::PinThreadNearDevice(device) ? 1 : 0; }

int fh_router_device_count(void) { return folve::DeviceRouter::Default()->device_count(); }
int fh_router_live_streams(int slot) { return folve::DeviceRouter::Default()->live_streams(slot); }
int fh_router_cached_filters(void) { return folve::DeviceRouter::Default()->cached_filters(); }
int fh_router_slot_state(int slot) { return static_cast<int>(folve::DeviceRouter::Default()->slot_state(slot)); }
long long fh_router_slot_failures(int slot) { return folve::DeviceRouter::Default()->slot_failures(slot); }
fe_engine* fh_router_slot_engine(int slot) { return folve::DeviceRouter::Default()->EngineIfCreated(slot); }
void fh_router_health_policy(int fence_after, double reprobe_seconds) {
    if (fence_after > 0) folve::DeviceRouter::Default()->SetFenceAfter(fence_after);
    if (reprobe_seconds >= 0) folve::DeviceRouter::Default()->SetReprobeSeconds(reprobe_seconds);
}
void fh_router_report_failure(fe_engine* e) { folve::DeviceRouter::Default()->ReportFailure(e); }

}  // extern "C"
