// Links and RUNS the libsndfile adapter (folve_amd/csrc/host/sndfile_adapter.cpp) — the two members with the
// reference's exact signatures, /root/reference/sound-processor.h:35,55 — against libfolve_amd.so, with
// sf_readf_float / sf_writef_float supplied by this binary over in-memory float "files" (the image has no
// libsndfile).  The driver below follows the call pattern of ConvolveFileHandler
// (/root/reference/convolve-file-handler.cc): open = ProcessorPool::GetOrCreate (:78-80), AddMoreSoundData
// (:370-424) including its gapless branch, PassoverProcessor (:328-351), Close = Return (:515), through the
// reference's own header names (include/dropin/*.h).  tests/test_adapter_gpu.py builds it, runs it on the GPU and
// compares what the "output files" received with the oracle.  This exercises the adapter; it pins nothing about zita.
//
//   usage: adapter_run <filter dir> <rate> <channels> <gapless 0|1> <run_ahead> <in A.f32> <in B.f32> <out A.f32> <out B.f32>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

// ---- a libsndfile stand-in for exactly the two calls the members make --------------------------------------
struct SNDFILE_tag {
    std::vector<float> data;     // interleaved frames
    int channels = 0;
    size_t pos = 0;              // read position, frames
    long long reads = 0, writes = 0;
};
typedef struct SNDFILE_tag SNDFILE;
typedef int64_t sf_count_t;
extern "C" {
sf_count_t sf_readf_float(SNDFILE* f, float* ptr, sf_count_t frames) {
    const size_t have = f->data.size() / f->channels - f->pos;
    const size_t n = (size_t)frames < have ? (size_t)frames : have;
    memcpy(ptr, f->data.data() + f->pos * f->channels, n * f->channels * sizeof(float));
    f->pos += n;
    f->reads++;
    return (sf_count_t)n;
}
sf_count_t sf_writef_float(SNDFILE* f, const float* ptr, sf_count_t frames) {
    f->data.insert(f->data.end(), ptr, ptr + (size_t)frames * f->channels);
    f->writes++;
    return frames;
}
// (the impulse-file fallback of the adapter wants these too; this binary's filters are WAVE files, read in-house:
// tests/compile/impulse_fallback.cpp is where the fallback runs)
struct SF_INFO { sf_count_t frames; int samplerate, channels, format, sections, seekable; };
enum { SFM_READ = 0x10 };
SNDFILE* sf_open(const char*, int, SF_INFO*) { return NULL; }
sf_count_t sf_seek(SNDFILE*, sf_count_t, int) { return -1; }
int sf_close(SNDFILE*) { return 0; }
}
#define FOLVE_AMD_SNDFILE_PROTOTYPES 1
#include "../../folve_amd/csrc/host/sndfile_adapter.cpp"

#include "../../include/dropin/processor-pool.h"
#include "../../include/dropin/sound-processor.h"
#include "../../include/folve_host.h"

namespace {

struct Filesystem {
    ProcessorPool* processor_pool() { return &pool; }
    bool gapless_processing() const { return gapless; }
    ProcessorPool pool{3};
    bool gapless = false;
};

// The part of ConvolveFileHandler that touches the seam.
class Handler {
public:
    Handler(Filesystem* fs, SNDFILE* in, SNDFILE* out, long long frames, const std::string& dir, int rate, int channels)
        : fs_(fs), snd_in_(in), snd_out_(out), frames_(frames), input_frames_left_(frames), next_(NULL) {
        std::string err;
        processor_ = fs_->processor_pool()->GetOrCreate(dir, rate, channels, 16, &err);
        if (!processor_) { fprintf(stderr, "GetOrCreate: %s\n", err.c_str()); exit(3); }
    }
    void set_next(Handler* n) { next_ = n; }
    bool HasStarted() const { return frames_ != input_frames_left_; }

    bool PassoverProcessor(SoundProcessor* passover_processor) {
        if (HasStarted()) return false;
        if (passover_processor->config_file() != processor_->config_file() ||
            passover_processor->config_file_timestamp() != processor_->config_file_timestamp()) {
            return false;
        }
        fs_->processor_pool()->Return(processor_);      // ours goes back; the donor's carries on
        processor_ = passover_processor;
        if (!processor_->is_input_buffer_complete()) {
            input_frames_left_ -= processor_->FillBuffer(snd_in_);     // our beginning completes the donor's last block
        }
        return true;
    }

    bool AddMoreSoundData() {
        if (!input_frames_left_) return false;
        if (processor_->pending_writes() > 0) {
            processor_->WriteProcessed(snd_out_, processor_->pending_writes());
            return input_frames_left_ != 0;
        }
        const int r = processor_->FillBuffer(snd_in_);
        if (r == 0) { input_frames_left_ = 0; Close(); return false; }
        input_frames_left_ -= r;
        if (!input_frames_left_ && !processor_->is_input_buffer_complete() && fs_->gapless_processing()) {
            const bool passed_processor = next_ != NULL && next_->PassoverProcessor(processor_);
            processor_->WriteProcessed(snd_out_, r);
            if (passed_processor) {
                max_out_ = processor_->max_output_value();
                processor_ = NULL;                          // ownership went to the next file
                Close();
            }
        } else {
            processor_->WriteProcessed(snd_out_, r);
        }
        if (input_frames_left_ == 0) Close();
        return input_frames_left_ != 0;
    }

    void Close() {
        if (processor_) {
            max_out_ = processor_->max_output_value();
            processor_->ResetMaxValues();
        }
        fs_->processor_pool()->Return(processor_);          // Return(NULL) is a no-op
        processor_ = NULL;
    }
    float max_out() const { return max_out_; }

private:
    Filesystem* fs_;
    SNDFILE* snd_in_;
    SNDFILE* snd_out_;
    const long long frames_;
    long long input_frames_left_;
    SoundProcessor* processor_;
    Handler* next_;
    float max_out_ = 0.f;
};

bool load(const char* path, int channels, SNDFILE* f) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    const long bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    f->data.resize((size_t)bytes / sizeof(float));
    f->channels = channels;
    const size_t got = fread(f->data.data(), sizeof(float), f->data.size(), fp);
    fclose(fp);
    return got == f->data.size();
}
bool save(const char* path, const SNDFILE& f) {
    FILE* fp = fopen(path, "wb");
    if (!fp) return false;
    const size_t put = fwrite(f.data.data(), sizeof(float), f.data.size(), fp);
    fclose(fp);
    return put == f.data.size();
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 10) { fprintf(stderr, "usage: %s dir rate channels gapless run_ahead inA inB outA outB\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const int rate = atoi(argv[2]), channels = atoi(argv[3]);
    Filesystem fs;
    fs.gapless = atoi(argv[4]) != 0;
    fh_run_ahead_set(atoi(argv[5]));
    SNDFILE in_a, in_b, out_a, out_b;
    if (!load(argv[6], channels, &in_a) || !load(argv[7], channels, &in_b)) { fprintf(stderr, "cannot read inputs\n"); return 2; }
    out_a.channels = out_b.channels = channels;
    {
        Handler a(&fs, &in_a, &out_a, (long long)(in_a.data.size() / channels), dir, rate, channels);
        Handler b(&fs, &in_b, &out_b, (long long)(in_b.data.size() / channels), dir, rate, channels);
        a.set_next(&b);
        while (a.AddMoreSoundData()) {}
        while (b.AddMoreSoundData()) {}
        printf("{\"max_a\": %.9g, \"max_b\": %.9g, \"reads_a\": %lld, \"reads_b\": %lld, \"writes_a\": %lld, \"writes_b\": %lld}\n",
               a.max_out(), b.max_out(), in_a.reads, in_b.reads, out_a.writes, out_b.writes);
    }
    return save(argv[8], out_a) && save(argv[9], out_b) ? 0 : 1;
}
