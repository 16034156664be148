// impulse_file.h — reads impulse-response sound files for /impulse/read.
//
// Stands where the reference's `Audiofile` (zita-audiofile.h:29-103, a thin
// libsndfile wrapper; only open_read/seek/read/close/rate/chan/size are used
// by zita-config.cc:101-175) stands.  libsndfile is not a dependency of the
// engine: this reads the uncompressed containers impulse responses come in —
// RIFF/WAVE (PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE which is
// also what Ambisonic .amb files are; RF64 is not handled), AIFF and AIFF-C
// ('NONE', 'sowt', 'fl32', 'fl64') and Core Audio Format ('lpcm'), the types
// zita-audiofile.cc:63-75 names — and normalises samples the way sf_readf_float
// does (int16/32768, int24/2^23, int32/2^31, 8-bit /128: unsigned in WAVE, signed
// in AIFF and CAF).
#pragma once

#include <stdint.h>
#include <stdio.h>

namespace folve {

class ImpulseFile {
public:
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
    int finish_open(FILE* f, int bits, bool is_float, uint64_t data_bytes);
    FILE* f_;
    int rate_, chan_, form_, bytes_per_sample_, block_align_;
    bool big_endian_, signed8_;
    uint32_t size_, pos_;
    long data_offset_;
};

}  // namespace folve
