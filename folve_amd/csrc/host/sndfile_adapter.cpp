// sndfile_adapter.cpp — the two libsndfile members of folve::SoundProcessor, with the
// reference's exact signatures (/root/reference/sound-processor.h:35,55; bodies follow
// sound-processor.cc:76-96: sf_readf_float into the block buffer, sf_writef_float out of it).
//
// Compiled where <sndfile.h> exists, i.e. by the folve build; on an image
// without libsndfile this translation unit is empty and libfolve_amd.so has no dependency on
// it.  tests/compile/ type-checks it (and the reference's literal call sites) against the two
// prototypes it needs, declared by the test.
#if __has_include(<sndfile.h>)
#include <sndfile.h>
#define FOLVE_AMD_HAVE_SNDFILE 1
#elif defined(FOLVE_AMD_SNDFILE_PROTOTYPES)      // the compile-only test declares the prototypes itself
#define FOLVE_AMD_HAVE_SNDFILE 1
#endif

#ifdef FOLVE_AMD_HAVE_SNDFILE
#include "sound_processor.h"

namespace folve {

namespace {
class SndfileSource : public FrameSource {
public:
    explicit SndfileSource(SNDFILE* f) : f_(f) {}
    int ReadFrames(float* dst, int frames) override { return static_cast<int>(sf_readf_float(f_, dst, frames)); }
private:
    SNDFILE* f_;
};
class SndfileSink : public FrameSink {
public:
    explicit SndfileSink(SNDFILE* f) : f_(f) {}
    int WriteFrames(const float* src, int frames) override { return static_cast<int>(sf_writef_float(f_, src, frames)); }
private:
    SNDFILE* f_;
};
}  // namespace

int SoundProcessor::FillBuffer(SNDFILE* in) {
    SndfileSource s(in);
    return FillBuffer(&s);
}

void SoundProcessor::WriteProcessed(SNDFILE* out, int sample_count) {
    SndfileSink s(out);
    WriteProcessed(&s, sample_count);
}

}  // namespace folve
#endif  // FOLVE_AMD_HAVE_SNDFILE
