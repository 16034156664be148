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
    return CreateOn(engine, config_file, samplerate, channels);
}

SoundProcessor* SoundProcessor::CreateOn(fe_engine* engine, const std::string& config_file, int samplerate,
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
    DeviceRouter::Default()->StreamOpened(engine);
    zita.engine = engine;
    return new SoundProcessor(zita, config_file, stream);
}

SoundProcessor::SoundProcessor(const ZitaConfig& config, const std::string& cfg, fe_stream* stream)
    : zita_config_(config), config_file_(cfg), config_file_timestamp_(GetModificationTime(cfg)), stream_(stream),
      buffer_(new float[static_cast<size_t>(config.fragm) * std::max(config.ninp, config.nout)]),
      input_pos_(0), output_pos_(0), max_out_value_observed_(0.0), max_abs_value_observed_(0.0), ok_(true) {
    Reset();
}

SoundProcessor::~SoundProcessor() {
    fe_stream_close(stream_);
    DeviceRouter::Default()->StreamClosed(zita_config_.engine);
    delete[] buffer_;
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
// tracks the maximum (sound-processor.cc:98-127); K1/K2/K3 do exactly that on
// the device: frames >= input_pos_ count as zero, input_pos_ frames come back.
void SoundProcessor::Process() {
    float peak_signed = 0.0f, peak_abs = 0.0f;
    if (input_pos_ > 0) {
        // With batching on, the block joins whatever other files' threads submit within the
        // collection window and runs as part of one launch; otherwise it is launched alone.
        const int rc = BatchScheduler::Enabled()
            ? BatchScheduler::ForEngine(zita_config_.engine)->Process(stream_, buffer_, input_pos_, buffer_,
                                                                      &peak_signed, &peak_abs)
            : fe_stream_process(stream_, buffer_, input_pos_, buffer_, &peak_signed, &peak_abs);
        if (rc != 0) {
            Logf("GPU convolution failed (%d): %s", rc, fe_last_error());
            memset(buffer_, 0, sizeof(float) * static_cast<size_t>(input_pos_) * output_channels());
            ok_ = false;
        } else {
            if (peak_signed > max_out_value_observed_) max_out_value_observed_ = peak_signed;
            if (peak_abs > max_abs_value_observed_) max_abs_value_observed_ = peak_abs;
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
