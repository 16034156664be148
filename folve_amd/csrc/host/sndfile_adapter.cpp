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
#include <stdio.h>                               // SEEK_SET
#define FOLVE_AMD_HAVE_SNDFILE 1
#elif defined(FOLVE_AMD_SNDFILE_PROTOTYPES)      // the compile-only test declares the prototypes itself
#define FOLVE_AMD_HAVE_SNDFILE 1
#endif

#ifdef FOLVE_AMD_HAVE_SNDFILE
#include "impulse_file.h"
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

// Impulse files in containers the engine's own reader does not decode (FLAC, Ogg, u-law .au, compressed AIFF-C ..): the
// reference's Audiofile is a thin wrapper over sf_open / sf_seek / sf_readf_float / sf_close and takes whatever
// libsndfile opens (/root/reference/zita-audiofile.cc:51-99,170-182), so where libsndfile exists the loader gets it as
// its fallback decoder and accepts every impulse file the reference accepts.  Registered when this translation unit is
// loaded.
namespace {
void* SndfileOpen(const char* name, int* rate, int* chan, uint32_t* frames) {
    SF_INFO info;
    info.format = 0;                                     // (sf_open wants format = 0 for reading, zita-audiofile.cc:56)
    info.frames = 0; info.samplerate = 0; info.channels = 0; info.sections = 0; info.seekable = 0;
    SNDFILE* f = sf_open(name, SFM_READ, &info);
    if (!f) return NULL;
    *rate = info.samplerate;
    *chan = info.channels;
    *frames = info.frames > 0xffffffffLL ? 0xffffffffu : static_cast<uint32_t>(info.frames < 0 ? 0 : info.frames);
    return f;
}
int SndfileSeek(void* h, uint32_t frame) {
    return sf_seek(static_cast<SNDFILE*>(h), static_cast<sf_count_t>(frame), SEEK_SET) == static_cast<sf_count_t>(frame) ? 0 : -1;
}
int SndfileRead(void* h, float* data, uint32_t frames) {
    return static_cast<int>(sf_readf_float(static_cast<SNDFILE*>(h), data, static_cast<sf_count_t>(frames)));
}
void SndfileClose(void* h) { sf_close(static_cast<SNDFILE*>(h)); }
const ImpulseOpener kSndfileOpener = {SndfileOpen, SndfileSeek, SndfileRead, SndfileClose};
const bool kSndfileOpenerRegistered = (ImpulseFile::SetFallbackOpener(&kSndfileOpener), true);
}  // namespace

bool SndfileImpulseOpenerRegistered() { return kSndfileOpenerRegistered; }

}  // namespace folve
#endif  // FOLVE_AMD_HAVE_SNDFILE
