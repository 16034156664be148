// sndfile_adapter.h — restores the reference's exact SNDFILE* call forms
// (sound-processor.h:35,55: `int FillBuffer(SNDFILE *in)`,
// `void WriteProcessed(SNDFILE *out, int sample_count)`) on top of
// folve::SoundProcessor for builds that have libsndfile.  Untestable on this
// image (no <sndfile.h>): header-only, compiled by the folve build.
#pragma once

#if __has_include(<sndfile.h>)
#include <sndfile.h>

#include "sound_processor.h"

namespace folve {

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

inline int FillBuffer(SoundProcessor* p, SNDFILE* in) {
    SndfileSource s(in);
    return p->FillBuffer(&s);
}

inline void WriteProcessed(SoundProcessor* p, SNDFILE* out, int sample_count) {
    SndfileSink s(out);
    p->WriteProcessed(&s, sample_count);
}

}  // namespace folve
#endif
