// Compile-only check of the drop-in claim: the reference's call sites of the
// convolver seam type-check, unchanged, against this repo's classes.
//
// What the reference does at the seam (all in /root/reference/convolve-file-handler.cc):
//   :78-80    fs->processor_pool()->GetOrCreate(dir, rate, channels, bits, &errmsg)
//   :156-158  processor_->max_output_value()
//   :177-178  processor_->config_file().c_str()
//   :335-347  passover_processor->config_file() / config_file_timestamp(); pool->Return(processor_);
//             processor_->is_input_buffer_complete(); processor_->FillBuffer(snd_in_)
//   :373-377  processor_->pending_writes(); processor_->WriteProcessed(snd_out_, n); FillBuffer(snd_in_)
//   :390      processor_->is_input_buffer_complete()
//   :408,418  processor_->WriteProcessed(snd_out_, r)
//   :498-500  max_output_value(), ResetMaxValues()
//   :515      pool->Return(processor_)
// The image has no <sndfile.h>; this file declares the handle type and the two libsndfile
// functions the members need (their published prototypes), then compiles the member definitions
// (sndfile_adapter.cpp) and the call forms.  Nothing here is linked or run.
#include <stdint.h>

#include <string>

#include <stdio.h>

typedef struct SNDFILE_tag SNDFILE;
typedef int64_t sf_count_t;
struct SF_INFO { sf_count_t frames; int samplerate, channels, format, sections, seekable; };
enum { SFM_READ = 0x10 };
extern "C" {
sf_count_t sf_readf_float(SNDFILE* sndfile, float* ptr, sf_count_t frames);
sf_count_t sf_writef_float(SNDFILE* sndfile, const float* ptr, sf_count_t frames);
// ... and the four the impulse-file fallback needs (zita-audiofile.cc:56,179-182)
SNDFILE* sf_open(const char* path, int mode, SF_INFO* sfinfo);
sf_count_t sf_seek(SNDFILE* sndfile, sf_count_t frames, int whence);
int sf_close(SNDFILE* sndfile);
}
#define FOLVE_AMD_SNDFILE_PROTOTYPES 1
#include "../../folve_amd/csrc/host/sndfile_adapter.cpp"

// the reference's own include lines resolve to these two headers (global-namespace classes)
#include "../../include/dropin/processor-pool.h"
#include "../../include/dropin/sound-processor.h"

namespace {

struct FakeFilesystem {
    ProcessorPool* processor_pool() { return pool; }
    ProcessorPool* pool;
};

struct CallSites {
    SNDFILE* snd_in_;
    SNDFILE* snd_out_;
    SoundProcessor* processor_;
    FakeFilesystem* fs_;
    int input_frames_left_;
    float max_seen;

    static SoundProcessor* Open(FakeFilesystem* fs, const std::string& filter_dir, int samplerate, int channels,
                                int bits, std::string* errmsg) {
        SoundProcessor* processor = fs->processor_pool()->GetOrCreate(filter_dir, samplerate, channels, bits, errmsg);
        return processor;
    }
    const char* Stats() {
        if (processor_ != NULL) max_seen = processor_->max_output_value();
        return processor_ != NULL ? processor_->config_file().c_str() : "filter";
    }
    bool Passover(SoundProcessor* passover_processor) {
        if (passover_processor->config_file() != processor_->config_file() ||
            passover_processor->config_file_timestamp() != processor_->config_file_timestamp()) {
            return false;
        }
        fs_->processor_pool()->Return(processor_);
        processor_ = passover_processor;
        if (!processor_->is_input_buffer_complete()) {
            input_frames_left_ -= processor_->FillBuffer(snd_in_);
        }
        return true;
    }
    bool AddMore() {
        if (processor_->pending_writes() > 0) {
            processor_->WriteProcessed(snd_out_, processor_->pending_writes());
            return input_frames_left_;
        }
        const int r = processor_->FillBuffer(snd_in_);
        if (r == 0) return false;
        input_frames_left_ -= r;
        if (!input_frames_left_ && !processor_->is_input_buffer_complete()) {
            processor_->WriteProcessed(snd_out_, r);
            processor_ = NULL;
            return false;
        }
        processor_->WriteProcessed(snd_out_, r);
        return input_frames_left_;
    }
    void Close() {
        if (processor_) {
            max_seen = processor_->max_output_value();
            processor_->ResetMaxValues();
        }
        fs_->processor_pool()->Return(processor_);
        processor_ = NULL;
    }
    int Channels() const { return processor_->output_channels() + processor_->input_channels(); }
};

}  // namespace

int folve_call_sites_compile_check() {
    ProcessorPool pool(3);
    FakeFilesystem fs{&pool};
    std::string err;
    CallSites c{NULL, NULL, CallSites::Open(&fs, "/nonexistent", 44100, 2, 16, &err), &fs, 0, 0.f};
    (void)c;
    return 0;
}
