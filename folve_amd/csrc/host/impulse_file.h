// impulse_file.h — reads impulse-response sound files for /impulse/read.
//
// Stands where the reference's `Audiofile` (zita-audiofile.h:29-103, a thin
// libsndfile wrapper; only open_read/seek/read/close/rate/chan/size are used
// by zita-config.cc:101-175) stands.  libsndfile is not a dependency of the
// engine: this reads the uncompressed containers impulse responses come in —
// RIFF/WAVE (PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE which is
// also what Ambisonic .amb files are), its 64-bit forms RF64 / BW64 ('ds64' sizes) and
// Sony Wave64 (GUID chunks), AIFF and AIFF-C ('NONE', 'sowt', 'fl32', 'fl64'), Core
// Audio Format ('lpcm') and Sun/NeXT .au (linear PCM and float encodings) — the types
// zita-audiofile.cc:63-75 names and the other uncompressed ones libsndfile opens — and
// normalises samples the way sf_readf_float does (int16/32768, int24/2^23, int32/2^31,
// 8-bit /128: unsigned in WAVE, signed in AIFF, CAF and AU).
//
// What this reader does not decode (FLAC, Ogg, compressed AIFF-C ..) goes to a FALLBACK
// OPENER when one is registered: the reference's Audiofile takes whatever sf_open opens
// (zita-audiofile.cc:51-99), so a folve build — which has libsndfile — registers one over
// sf_open / sf_seek / sf_readf_float / sf_close (host/sndfile_adapter.cpp) and accepts
// every impulse file the reference accepts.
#pragma once

#include <stdint.h>
#include <stdio.h>

namespace folve {

// What an external decoder supplies (libsndfile in a folve build).  open: NULL if it cannot read the file either.
struct ImpulseOpener {
    void* (*open)(const char* name, int* rate, int* chan, uint32_t* frames);
    int (*seek)(void* handle, uint32_t frame);                     // 0 on success
    int (*read)(void* handle, float* data, uint32_t frames);       // interleaved frames read, 0 at the end, < 0 on error
    void (*close)(void* handle);
};

class ImpulseFile {
public:
    // Process-wide; NULL removes it.  The table must outlive every ImpulseFile.
    static void SetFallbackOpener(const ImpulseOpener* opener);
    static const ImpulseOpener* FallbackOpener();

    enum { ERR_NONE = 0, ERR_MODE = -1, ERR_TYPE = -2, ERR_FORM = -3, ERR_OPEN = -4, ERR_SEEK = -5,
           ERR_DATA = -6, ERR_READ = -7 };
    enum { FORM_OTHER, FORM_8BIT, FORM_16BIT, FORM_24BIT, FORM_32BIT, FORM_FLOAT, FORM_DOUBLE };

    ImpulseFile();
    ~ImpulseFile();
    ImpulseFile(const ImpulseFile&) = delete;
    ImpulseFile& operator=(const ImpulseFile&) = delete;

    int open_read(const char* name);
    int close();
    int seek(uint32_t frame);
    // Reads up to `frames` interleaved frames as float; returns frames read (0 at EOF) or < 0.
    int read(float* data, uint32_t frames);

    int rate() const { return rate_; }
    int chan() const { return chan_; }
    int form() const { return form_; }
    uint32_t size() const { return size_; }

private:
    void reset();
    int open_wave(FILE* f);
    int open_aiff(FILE* f, bool aifc);
    int open_caf(FILE* f);
    int open_w64(FILE* f);
    int open_au(FILE* f, const unsigned char* hdr12);
    int finish_open(FILE* f, int bits, bool is_float, uint64_t data_bytes);
    FILE* f_;
    int rate_, chan_, form_, bytes_per_sample_, block_align_;
    bool big_endian_, signed8_;
    uint32_t size_, pos_;
    long data_offset_;
    void* ext_;                         // the fallback opener's handle while it reads this file
};

}  // namespace folve
