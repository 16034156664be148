#include "sound_processor.h"

#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>

#include "batch_scheduler.h"
#include "device_router.h"

namespace folve {

static time_t GetModificationTime(const std::string& filename) {
    struct stat st;
    if (stat(filename.c_str(), &st) != 0) return 0;
    return st.st_mtime;
}

SoundProcessor* SoundProcessor::Create(const std::string& config_file, int samplerate, int channels) {
    fe_engine* engine = DeviceRouter::Default()->PickEngine();
    if (!engine) {
        Logf("No usable GPU: cannot create a processor for %s (there is no CPU fallback)", config_file.c_str());
        return NULL;
    }
    SoundProcessor* p = CreateOnReserved(engine, config_file, samplerate, channels);
    if (!p) DeviceRouter::Default()->StreamClosed(engine);      // give the reservation back
    return p;
}

SoundProcessor* SoundProcessor::CreateOn(fe_engine* engine, const std::string& config_file, int samplerate,
                                         int channels) {
    SoundProcessor* p = CreateOnReserved(engine, config_file, samplerate, channels);
    if (p) DeviceRouter::Default()->StreamOpened(engine);
    return p;
}

// The caller has already accounted the stream to `engine` in the router.
SoundProcessor* SoundProcessor::CreateOnReserved(fe_engine* engine, const std::string& config_file, int samplerate,
                                                 int channels) {
    if (!engine) return NULL;
    ZitaConfig zita;
    memset(&zita, 0, sizeof(zita));
    // Parsing and the filter's FFTs happen once per (config, mtime, GPU); every
    // processor of that configuration shares the committed spectra.
    fe_filter* filter = DeviceRouter::Default()->GetFilter(engine, config_file, GetModificationTime(config_file),
                                                           samplerate, channels, &zita);
    if (!filter) return NULL;
    fe_stream* stream = NULL;
    // One block per call is the reference's contract; the ring is sized for run-ahead batches too.
    static const int kMaxBlocksPerCall = 32;
    const int rc = fe_stream_open(filter, kMaxBlocksPerCall, &stream);
    fe_filter_release(filter);           // the stream holds its own reference
    if (rc != 0) {
        Logf("Cannot open a convolver stream for %s: %s", config_file.c_str(), fe_last_error());
        return NULL;
    }
    zita.engine = engine;
    return new SoundProcessor(zita, config_file, stream);
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

SoundProcessor::SoundProcessor(const ZitaConfig& config, const std::string& cfg, fe_stream* stream)
    : zita_config_(config), config_file_(cfg), config_file_timestamp_(GetModificationTime(cfg)), stream_(stream),
      buffer_floats_(static_cast<size_t>(config.fragm) * std::max(config.ninp, config.nout)),
      buffer_(AllocBlockBuffer(buffer_floats_, &buffer_pinned_)),
      input_pos_(0), output_pos_(0), max_out_value_observed_(0.0), max_abs_value_observed_(0.0), ok_(true) {
    if (buffer_pinned_ && fe_stream_bind_host_buffer(stream_, buffer_, buffer_floats_ * sizeof(float)) != 0) {
        Logf("Processor %p: block buffer not bound (%s): blocks will be staged", static_cast<void*>(this), fe_last_error());
    }
    Reset();
}

SoundProcessor::~SoundProcessor() {
    fe_stream_close(stream_);
    DeviceRouter::Default()->StreamClosed(zita_config_.engine);
    if (buffer_pinned_) fe_host_free(buffer_);
    else delete[] buffer_;
}

int SoundProcessor::device() const { return fe_engine_device(zita_config_.engine); }

int SoundProcessor::FillBuffer(FrameSource* in) {
    const int samples_needed = zita_config_.fragm - input_pos_;
    assert(samples_needed);   // Otherwise, call WriteProcessed() first.
    output_pos_ = -1;
    const int r = in->ReadFrames(buffer_ + static_cast<size_t>(input_pos_) * input_channels(), samples_needed);
    input_pos_ += r;
    return r;
}

void SoundProcessor::WriteProcessed(FrameSink* out, int sample_count) {
    if (output_pos_ < 0) {
        Process();
    }
    assert(sample_count <= zita_config_.fragm - output_pos_);
    out->WriteFrames(buffer_ + static_cast<size_t>(output_pos_) * output_channels(), sample_count);
    output_pos_ += sample_count;
    if (output_pos_ == zita_config_.fragm) {
        input_pos_ = 0;
    }
}

// One block through the GPU.  The reference zero-fills the unread tail, splits
// the channels, runs Convproc::process(), re-interleaves input_pos_ frames and
// tracks the maximum (sound-processor.cc:98-127); K1/K2/K3 do the first four on
// the device — frames >= input_pos_ count as zero, input_pos_ frames come back —
// and the maximum is taken here over the returned frames, signed as cc:120-123 does.
void SoundProcessor::Process() {
    if (input_pos_ > 0) {
        // The call goes through the device's combiner: alone it runs at once; while another file's
        // block is in flight on this GPU it is parked and leaves with the next batch.
        std::string error;
        int rc;
        if (BatchScheduler::Enabled()) {
            rc = BatchScheduler::ForEngine(zita_config_.engine)->Process(stream_, buffer_, input_pos_, buffer_, &error);
        } else {
            rc = fe_stream_process(stream_, buffer_, input_pos_, buffer_, NULL, NULL);
            if (rc != 0) error = fe_last_error();
        }
        const size_t n = static_cast<size_t>(input_pos_) * output_channels();
        if (rc != 0) {
            Logf("GPU convolution failed (%d): %s", rc, error.c_str());
            memset(buffer_, 0, sizeof(float) * n);
            ok_ = false;                 // ProcessorPool::Return will not pool this processor
        } else {
            float hi = max_out_value_observed_, mag = max_abs_value_observed_;
            for (size_t j = 0; j < n; ++j) {
                const float v = buffer_[j];
                hi = v > hi ? v : hi;
                const float a = v < 0 ? -v : v;
                mag = a > mag ? a : mag;
            }
            max_out_value_observed_ = hi;
            max_abs_value_observed_ = mag;
        }
    }
    output_pos_ = 0;
}

bool SoundProcessor::ConfigStillUpToDate() const {
    return config_file_timestamp_ == GetModificationTime(config_file_);
}

void SoundProcessor::ResetMaxValues() {
    max_out_value_observed_ = 0.0;
    max_abs_value_observed_ = 0.0;
    fe_stream_reset_peaks(stream_);
}

void SoundProcessor::Reset() {
    fe_stream_reset(stream_);
    input_pos_ = 0;
    output_pos_ = -1;
    ResetMaxValues();
}

}  // namespace folve
