// sound_processor.h — drop-in for folve's SoundProcessor on top of the GPU engine.
//
// Public surface and behaviour follow /root/reference/sound-processor.h:28-85 and
// sound-processor.cc:34-145 method for method.  The reference's two libsndfile members
//     int  FillBuffer(SNDFILE *in);                          (sound-processor.h:35)
//     void WriteProcessed(SNDFILE *out, int sample_count);   (sound-processor.h:55)
// are declared here against an opaque SNDFILE and defined in sndfile_adapter.cpp, which is
// compiled only where <sndfile.h> exists (the folve build); everything else — the block
// machine itself — works on FrameSource / FrameSink (libsndfile's sf_readf_float /
// sf_writef_float contract), so the library builds and is tested without libsndfile.
#pragma once

#include <time.h>

#include <string>

#include "zita_config.h"

// <sndfile.h>'s own declaration of the handle type (a typedef may be repeated)
typedef struct SNDFILE_tag SNDFILE;

namespace folve {

// sf_readf_float(in, dst, frames): fills up to `frames` interleaved frames, returns frames read.
class FrameSource {
public:
    virtual ~FrameSource() {}
    virtual int ReadFrames(float* dst, int frames) = 0;
};

// sf_writef_float(out, src, frames): consumes `frames` interleaved frames.
class FrameSink {
public:
    virtual ~FrameSink() {}
    virtual int WriteFrames(const float* src, int frames) = 0;
};

class FilterCache;

// The workhorse of processing data from soundfiles.
class SoundProcessor {
public:
    // As the reference: NULL if the configuration cannot be parsed or does not
    // define a convolver.  The GPU is chosen by the process-wide DeviceRouter.
    static SoundProcessor* Create(const std::string& config_file, int samplerate, int channels);
    // Same, on a given engine (used by ProcessorPool's sharder).
    static SoundProcessor* CreateOn(fe_engine* engine, const std::string& config_file, int samplerate, int channels);
private:
    static SoundProcessor* CreateOnReserved(fe_engine* engine, const std::string& config_file, int samplerate, int channels);
public:
    ~SoundProcessor();

    // Fill buffer from given source.  Returns number of frames read.
    int FillBuffer(FrameSource* in);
    // The reference's signature (sound-processor.h:35); defined in sndfile_adapter.cpp.
    int FillBuffer(SNDFILE* in);

    inline int input_channels() const { return zita_config_.ninp; }
    inline int output_channels() const { return zita_config_.nout; }

    // True if the input buffer holds a whole block for the FIR filter.
    bool is_input_buffer_complete() const { return zita_config_.fragm == input_pos_; }

    // Frames processed but not yet written (gapless hand-over, see
    // convolve-file-handler.cc:373-376).
    int pending_writes() const { return output_pos_ >= 0 ? zita_config_.fragm - output_pos_ : 0; }

    // Write `sample_count` processed frames to `out`; processes first if necessary.
    void WriteProcessed(FrameSink* out, int sample_count);
    // The reference's signature (sound-processor.h:55); defined in sndfile_adapter.cpp.
    void WriteProcessed(SNDFILE* out, int sample_count);

    // Reset processor for re-use.
    void Reset();

    // Maximum output value observed.  As in the reference (sound-processor.cc:120-123)
    // the comparison is on the signed sample; max_abs_output_value() is the magnitude.
    float max_output_value() const { return max_out_value_observed_; }
    float max_abs_output_value() const { return max_abs_value_observed_; }
    void ResetMaxValues();

    const std::string& config_file() const { return config_file_; }
    time_t config_file_timestamp() const { return config_file_timestamp_; }
    bool ConfigStillUpToDate() const;

    int block_size() const { return zita_config_.fragm; }
    fe_stream* stream() const { return stream_; }      // for batched submission
    fe_engine* engine() const { return zita_config_.engine; }
    int device() const;
    bool ok() const { return ok_; }                     // false after an engine failure

private:
    SoundProcessor(const ZitaConfig& config, const std::string& cfg_file, fe_stream* stream);
    void Process();

    const ZitaConfig zita_config_;
    const std::string config_file_;
    const time_t config_file_timestamp_;
    fe_stream* const stream_;

    const size_t buffer_floats_;
    bool buffer_pinned_;
    float* const buffer_;
    int input_pos_;
    int output_pos_;   // written position. -1, if not processed yet.
    float max_out_value_observed_;
    float max_abs_value_observed_;
    bool ok_;
};

}  // namespace folve
