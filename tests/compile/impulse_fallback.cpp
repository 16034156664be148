// The loader's fallback decoder, linked and run (no GPU needed: a filter is assembled on the host until it is committed).
// folve_amd/csrc/host/sndfile_adapter.cpp registers sf_open / sf_seek / sf_readf_float / sf_close as the decoder of
// impulse files the engine's own reader does not read — the reference's Audiofile takes whatever libsndfile opens
// (/root/reference/zita-audiofile.cc:51-99,170-182).  The image has no libsndfile: this binary supplies the four calls over
// a made-up container ("TOY1": rate, channels, frames, float32 samples) that stands for FLAC / Ogg / anything the
// in-house reader refuses, compiles the adapter against them, and loads a configuration whose /impulse/read lines name
// such files through the library's real loader (fh_config_load).  tests/test_host_cpu.py builds and runs it.
//
//   usage: impulse_fallback <config file> <taps to print>      prints one JSON line
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

struct SNDFILE_tag {
    FILE* f;
    int channels;
    long long frames, pos;
};
typedef struct SNDFILE_tag SNDFILE;
typedef int64_t sf_count_t;
struct SF_INFO { sf_count_t frames; int samplerate, channels, format, sections, seekable; };   // <sndfile.h>'s layout
enum { SFM_READ = 0x10 };
static long long g_opens = 0, g_closes = 0, g_seeks = 0;
extern "C" {
SNDFILE* sf_open(const char* path, int mode, SF_INFO* info) {
    if (mode != SFM_READ || info->format != 0) return NULL;
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    char magic[4];
    int32_t h[3];
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "TOY1", 4) != 0 || fread(h, 4, 3, f) != 3) { fclose(f); return NULL; }
    info->samplerate = h[0]; info->channels = h[1]; info->frames = h[2]; info->seekable = 1; info->sections = 1;
    ++g_opens;
    return new SNDFILE_tag{f, h[1], h[2], 0};
}
sf_count_t sf_seek(SNDFILE* s, sf_count_t frames, int whence) {
    if (whence != SEEK_SET || frames < 0 || frames > s->frames) return -1;
    fseek(s->f, 16 + (long)frames * s->channels * 4, SEEK_SET);
    s->pos = frames;
    ++g_seeks;
    return frames;
}
sf_count_t sf_readf_float(SNDFILE* s, float* ptr, sf_count_t frames) {
    const long long n = frames < s->frames - s->pos ? frames : s->frames - s->pos;
    const size_t got = fread(ptr, sizeof(float) * (size_t)s->channels, (size_t)n, s->f);
    s->pos += (long long)got;
    return (sf_count_t)got;
}
sf_count_t sf_writef_float(SNDFILE*, const float*, sf_count_t frames) { return frames; }
int sf_close(SNDFILE* s) { fclose(s->f); delete s; ++g_closes; return 0; }
}
#define FOLVE_AMD_SNDFILE_PROTOTYPES 1
#include "../../folve_amd/csrc/host/sndfile_adapter.cpp"

#include "../../include/folve_host.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int n = atoi(argv[2]);
    if (!folve::SndfileImpulseOpenerRegistered()) return 3;
    fe_filter* flt = NULL;
    int fragm = 0, ninp = 0, nout = 0, size = 0;
    const int st = fh_config_load(NULL, argv[1], 44100, 1, &flt, &fragm, &ninp, &nout, &size);
    printf("{\"status\": %d, \"nout\": %d, \"opens\": %lld, \"closes\": %lld, \"seeks\": %lld, \"taps\": [", st, nout, g_opens, g_closes, g_seeks);
    for (int o = 0; flt && o < nout; ++o) {
        std::vector<float> h((size_t)n, 0.f);
        const int parts = fe_filter_path_partitions(flt, 0, o);
        if (parts > 0) fe_filter_get_taps(flt, 0, o, h.data(), n);
        printf("%s[", o ? ", " : "");
        for (int i = 0; i < n; ++i) printf("%s%.9g", i ? ", " : "", h[(size_t)i]);
        printf("]");
    }
    printf("]}\n");
    if (flt) fe_filter_release(flt);
    return 0;
}
