#include "impulse_file.h"

#include <math.h>
#include <string.h>

#include <atomic>
#include <vector>

namespace folve {

namespace {
std::atomic<const ImpulseOpener*> g_fallback{nullptr};
}
void ImpulseFile::SetFallbackOpener(const ImpulseOpener* opener) { g_fallback.store(opener); }
const ImpulseOpener* ImpulseFile::FallbackOpener() { return g_fallback.load(); }

namespace {
uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t le16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
uint32_t be32(const unsigned char* p) { return p[3] | (p[2] << 8) | (p[1] << 16) | (static_cast<uint32_t>(p[0]) << 24); }
uint16_t be16(const unsigned char* p) { return static_cast<uint16_t>(p[1] | (p[0] << 8)); }
uint64_t be64(const unsigned char* p) { return (static_cast<uint64_t>(be32(p)) << 32) | be32(p + 4); }
uint64_t le64(const unsigned char* p) { return (static_cast<uint64_t>(le32(p + 4)) << 32) | le32(p); }
// Sony Wave64 chunk ids: a four-character code followed by this tail
const unsigned char kW64Tail[12] = {0xf3, 0xac, 0xd3, 0x11, 0x8c, 0xd1, 0x00, 0xc0, 0x4f, 0x8e, 0xdb, 0x8a};
const unsigned char kW64RiffTail[12] = {0x2e, 0x91, 0xcf, 0x11, 0xa5, 0xd6, 0x28, 0xdb, 0x04, 0xc1, 0x00, 0x00};
// 80-bit IEEE 754 extended (AIFF sample rate)
double ext80(const unsigned char* p) {
    const int sign = p[0] >> 7;
    const int exp = ((p[0] & 0x7f) << 8) | p[1];
    const uint64_t mant = be64(p + 2);
    if (exp == 0 && mant == 0) return 0.0;
    const double v = ldexp(static_cast<double>(mant), exp - 16383 - 63);
    return sign ? -v : v;
}
}  // namespace

ImpulseFile::ImpulseFile() : f_(nullptr), ext_(nullptr) { reset(); }
ImpulseFile::~ImpulseFile() { close(); }

void ImpulseFile::reset() {
    f_ = nullptr;
    rate_ = chan_ = 0;
    form_ = FORM_OTHER;
    bytes_per_sample_ = block_align_ = 0;
    big_endian_ = signed8_ = false;
    size_ = pos_ = 0;
    data_offset_ = 0;
    ext_ = nullptr;
}

int ImpulseFile::close() {
    if (f_) fclose(f_);
    if (ext_) {
        const ImpulseOpener* op = g_fallback.load();
        if (op && op->close) op->close(ext_);
    }
    reset();
    return 0;
}

int ImpulseFile::open_read(const char* name) {
    if (f_ || ext_) return ERR_MODE;
    reset();
    FILE* f = fopen(name, "rb");
    if (!f) return ERR_OPEN;
    unsigned char hdr[12];
    int rc = ERR_TYPE;
    if (fread(hdr, 1, 12, f) == 12) {
        if ((memcmp(hdr, "RIFF", 4) == 0 || memcmp(hdr, "RF64", 4) == 0 || memcmp(hdr, "BW64", 4) == 0) &&
            memcmp(hdr + 8, "WAVE", 4) == 0) rc = open_wave(f);
        else if (memcmp(hdr, "riff", 4) == 0 && memcmp(hdr + 4, kW64RiffTail, 8) == 0) rc = (fseek(f, 0, SEEK_SET) == 0) ? open_w64(f) : ERR_DATA;
        else if (memcmp(hdr, ".snd", 4) == 0) rc = open_au(f, hdr);
        else if (memcmp(hdr, "FORM", 4) == 0 && memcmp(hdr + 8, "AIFF", 4) == 0) rc = open_aiff(f, false);
        else if (memcmp(hdr, "FORM", 4) == 0 && memcmp(hdr + 8, "AIFC", 4) == 0) rc = open_aiff(f, true);
        else if (memcmp(hdr, "caff", 4) == 0) rc = (fseek(f, 8, SEEK_SET) == 0) ? open_caf(f) : ERR_DATA;
    }
    if (rc != ERR_NONE) {
        fclose(f);
        reset();
        // not a container (or not an encoding) this reader decodes: whatever libsndfile opens is an impulse file to the
        // reference (zita-audiofile.cc:51-99) — ask the registered decoder, if any
        const ImpulseOpener* op = g_fallback.load();
        if ((rc == ERR_TYPE || rc == ERR_FORM) && op && op->open) {
            int r = 0, c = 0;
            uint32_t n = 0;
            void* h = op->open(name, &r, &c, &n);
            if (h) {
                if (c < 1) { if (op->close) op->close(h); return ERR_DATA; }
                ext_ = h; rate_ = r; chan_ = c; size_ = n; pos_ = 0;
                return ERR_NONE;
            }
        }
    }
    return rc;
}

// The stream is positioned at the first sample.  data_bytes: what the header claims.
int ImpulseFile::finish_open(FILE* f, int bits, bool is_float, uint64_t data_bytes) {
    if (chan_ < 1 || rate_ < 0) return ERR_DATA;
    bytes_per_sample_ = bits / 8;
    if (!is_float) {
        switch (bytes_per_sample_) {
            case 1: form_ = FORM_8BIT; break;
            case 2: form_ = FORM_16BIT; break;
            case 3: form_ = FORM_24BIT; break;
            case 4: form_ = FORM_32BIT; break;
            default: return ERR_FORM;
        }
    } else if (bytes_per_sample_ == 4) {
        form_ = FORM_FLOAT;
    } else if (bytes_per_sample_ == 8) {
        form_ = FORM_DOUBLE;
    } else {
        return ERR_FORM;
    }
    if (bits != bytes_per_sample_ * 8) return ERR_FORM;
    if (block_align_ == 0) block_align_ = bytes_per_sample_ * chan_;
    if (block_align_ != bytes_per_sample_ * chan_) return ERR_FORM;
    data_offset_ = ftell(f);
    fseek(f, 0, SEEK_END);
    const long avail = ftell(f) - data_offset_;
    fseek(f, data_offset_, SEEK_SET);
    uint64_t bytes = data_bytes;
    if (avail >= 0 && bytes > static_cast<uint64_t>(avail)) bytes = static_cast<uint64_t>(avail);
    const uint64_t frames = bytes / static_cast<uint64_t>(block_align_);
    size_ = frames > 0xffffffffu ? 0xffffffffu : static_cast<uint32_t>(frames);
    pos_ = 0;
    f_ = f;
    return ERR_NONE;
}

int ImpulseFile::open_wave(FILE* f) {
    int tag = 0, bits = 0;
    bool have_fmt = false;
    uint64_t data64 = 0;                  // RF64 / BW64: the 'data' chunk's real size, from 'ds64' (its 32-bit field says 0xffffffff)
    bool have_ds64 = false;
    for (;;) {
        unsigned char ck[8];
        if (fread(ck, 1, 8, f) != 8) return ERR_DATA;
        const uint32_t len = le32(ck + 4);
        if (memcmp(ck, "ds64", 4) == 0) {
            unsigned char b[24];                                    // riff size, data size, sample count (8 bytes each) [, table]
            if (len < 24 || fread(b, 1, 24, f) != 24) return ERR_DATA;
            data64 = le64(b + 8);
            have_ds64 = true;
            const long skip = static_cast<long>(len - 24) + (len & 1);
            if (skip && fseek(f, skip, SEEK_CUR) != 0) return ERR_DATA;
        } else if (memcmp(ck, "fmt ", 4) == 0) {
            unsigned char b[40];
            const uint32_t n = len < sizeof(b) ? len : static_cast<uint32_t>(sizeof(b));
            if (len < 16 || fread(b, 1, n, f) != n) return ERR_DATA;
            tag = le16(b);
            chan_ = le16(b + 2);
            rate_ = static_cast<int>(le32(b + 4));
            block_align_ = le16(b + 12);
            bits = le16(b + 14);
            if (tag == 0xFFFE && n >= 26) tag = le16(b + 24);       // extensible: sub-format GUID's first word
            const long skip = static_cast<long>(len - n) + (len & 1);
            if (skip) fseek(f, skip, SEEK_CUR);
            have_fmt = true;
        } else if (memcmp(ck, "data", 4) == 0) {
            if (!have_fmt || block_align_ < 1) return ERR_DATA;
            if (tag != 1 && tag != 3) return ERR_FORM;
            return finish_open(f, bits, tag == 3, (len == 0xffffffffu && have_ds64) ? data64 : len);
        } else {
            if (fseek(f, static_cast<long>(len) + (len & 1), SEEK_CUR) != 0) return ERR_DATA;
        }
    }
}

// Sony Wave64: the RIFF/WAVE chunks with 16-byte GUID ids and 64-bit sizes that INCLUDE the 24-byte chunk header;
// chunks are padded to 8 bytes.  The file starts with the 'riff' GUID, its size and the 'wave' GUID.
int ImpulseFile::open_w64(FILE* f) {
    unsigned char head[40];
    if (fread(head, 1, 40, f) != 40 || memcmp(head + 24, "wave", 4) != 0 || memcmp(head + 28, kW64Tail, 12) != 0) return ERR_TYPE;
    int tag = 0, bits = 0;
    bool have_fmt = false;
    for (;;) {
        unsigned char ck[24];
        if (fread(ck, 1, 24, f) != 24) return ERR_DATA;
        const uint64_t len = le64(ck + 16);
        if (len < 24 || memcmp(ck + 4, kW64Tail, 12) != 0) return ERR_DATA;
        const uint64_t body = len - 24, pad = (8 - (len & 7)) & 7;
        if (memcmp(ck, "fmt ", 4) == 0) {
            unsigned char b[40];
            const size_t n = body < sizeof(b) ? static_cast<size_t>(body) : sizeof(b);
            if (body < 16 || fread(b, 1, n, f) != n) return ERR_DATA;
            tag = le16(b);
            chan_ = le16(b + 2);
            rate_ = static_cast<int>(le32(b + 4));
            block_align_ = le16(b + 12);
            bits = le16(b + 14);
            if (tag == 0xFFFE && n >= 26) tag = le16(b + 24);
            if (fseek(f, static_cast<long>(body - n + pad), SEEK_CUR) != 0) return ERR_DATA;
            have_fmt = true;
        } else if (memcmp(ck, "data", 4) == 0) {
            if (!have_fmt || block_align_ < 1) return ERR_DATA;
            if (tag != 1 && tag != 3) return ERR_FORM;
            return finish_open(f, bits, tag == 3, body);
        } else {
            if (fseek(f, static_cast<long>(body + pad), SEEK_CUR) != 0) return ERR_DATA;
        }
    }
}

// Sun / NeXT .au: big-endian header — magic, data offset, data size (0xffffffff: to the end), encoding, rate, channels.
// Encodings 2 .. 7 are linear PCM of 8 / 16 / 24 / 32 bits and IEEE float / double; the companded ones (1 u-law, 27 A-law)
// and ADPCM are left to the fallback opener.
int ImpulseFile::open_au(FILE* f, const unsigned char* hdr12) {
    unsigned char b[12];
    if (fread(b, 1, 12, f) != 12) return ERR_DATA;
    const uint32_t offset = be32(hdr12 + 4), bytes = be32(hdr12 + 8), enc = be32(b);
    rate_ = static_cast<int>(be32(b + 4));
    chan_ = static_cast<int>(be32(b + 8));
    big_endian_ = true;
    signed8_ = true;
    int bits = 0;
    bool is_float = false;
    switch (enc) {
        case 2: bits = 8; break;
        case 3: bits = 16; break;
        case 4: bits = 24; break;
        case 5: bits = 32; break;
        case 6: bits = 32; is_float = true; break;
        case 7: bits = 64; is_float = true; break;
        default: return ERR_FORM;
    }
    if (offset < 24 || fseek(f, static_cast<long>(offset), SEEK_SET) != 0) return ERR_DATA;
    return finish_open(f, bits, is_float, bytes == 0xffffffffu ? ~static_cast<uint64_t>(0) : bytes);
}

// AIFF / AIFF-C: big-endian chunks; COMM = channels, frames, bits, 80-bit rate [, compression id]
int ImpulseFile::open_aiff(FILE* f, bool aifc) {
    int bits = 0;
    uint32_t frames = 0;
    bool have_comm = false, is_float = false;
    big_endian_ = true;
    signed8_ = true;
    for (;;) {
        unsigned char ck[8];
        if (fread(ck, 1, 8, f) != 8) return ERR_DATA;
        const uint32_t len = be32(ck + 4);
        if (memcmp(ck, "COMM", 4) == 0) {
            unsigned char b[22];
            const uint32_t need = aifc ? 22 : 18;
            if (len < need || fread(b, 1, need, f) != need) return ERR_DATA;
            chan_ = be16(b);
            frames = be32(b + 2);
            bits = be16(b + 6);
            rate_ = static_cast<int>(ext80(b + 8) + 0.5);
            if (aifc) {
                const unsigned char* id = b + 18;
                if (memcmp(id, "NONE", 4) == 0 || memcmp(id, "twos", 4) == 0) {
                } else if (memcmp(id, "sowt", 4) == 0) {
                    big_endian_ = false;
                } else if (memcmp(id, "fl32", 4) == 0 || memcmp(id, "FL32", 4) == 0) {
                    is_float = true; bits = 32;
                } else if (memcmp(id, "fl64", 4) == 0 || memcmp(id, "FL64", 4) == 0) {
                    is_float = true; bits = 64;
                } else {
                    return ERR_FORM;                                // a compressed AIFF-C
                }
            }
            bits = (bits + 7) / 8 * 8;                              // AIFF stores e.g. 20-bit samples in 3 bytes
            const long skip = static_cast<long>(len - need) + (len & 1);
            if (skip) fseek(f, skip, SEEK_CUR);
            have_comm = true;
        } else if (memcmp(ck, "SSND", 4) == 0) {
            unsigned char b[8];
            if (!have_comm || len < 8 || fread(b, 1, 8, f) != 8) return ERR_DATA;   // COMM after SSND: not handled
            const uint32_t offset = be32(b);
            if (offset && fseek(f, static_cast<long>(offset), SEEK_CUR) != 0) return ERR_DATA;
            const uint64_t claimed = static_cast<uint64_t>(frames) * static_cast<uint64_t>(chan_) * (bits / 8);
            const uint64_t in_chunk = static_cast<uint64_t>(len) - 8 - offset;
            return finish_open(f, bits, is_float, claimed < in_chunk ? claimed : in_chunk);
        } else {
            if (fseek(f, static_cast<long>(len) + (len & 1), SEEK_CUR) != 0) return ERR_DATA;
        }
    }
}

// Core Audio Format: chunks of (type, int64 size); 'desc' = f64 rate, format id, flags, bytes/packet,
// frames/packet, channels, bits; 'data' = uint32 edit count + samples (size -1: to the end of the file)
int ImpulseFile::open_caf(FILE* f) {
    int bits = 0;
    bool have_desc = false, is_float = false;
    signed8_ = true;
    for (;;) {
        unsigned char ck[12];
        if (fread(ck, 1, 12, f) != 12) return ERR_DATA;
        const int64_t len = static_cast<int64_t>(be64(ck + 4));
        if (memcmp(ck, "desc", 4) == 0) {
            unsigned char b[32];
            if (len < 32 || fread(b, 1, 32, f) != 32) return ERR_DATA;
            const uint64_t rb = be64(b);
            double r; memcpy(&r, &rb, 8);
            rate_ = static_cast<int>(r + 0.5);
            if (memcmp(b + 8, "lpcm", 4) != 0) return ERR_FORM;
            const uint32_t flags = be32(b + 12);
            is_float = (flags & 1) != 0;
            big_endian_ = (flags & 2) == 0;
            block_align_ = static_cast<int>(be32(b + 16));
            if (be32(b + 20) != 1) return ERR_FORM;                 // frames per packet
            chan_ = static_cast<int>(be32(b + 24));
            bits = static_cast<int>(be32(b + 28));
            if (len > 32) fseek(f, static_cast<long>(len - 32), SEEK_CUR);
            have_desc = true;
        } else if (memcmp(ck, "data", 4) == 0) {
            unsigned char edit[4];
            if (!have_desc || fread(edit, 1, 4, f) != 4) return ERR_DATA;
            const uint64_t bytes = len < 0 ? ~static_cast<uint64_t>(0) : static_cast<uint64_t>(len - 4);
            return finish_open(f, bits, is_float, bytes);
        } else {
            if (len < 0 || fseek(f, static_cast<long>(len), SEEK_CUR) != 0) return ERR_DATA;
        }
    }
}

int ImpulseFile::seek(uint32_t frame) {
    if (ext_) {
        const ImpulseOpener* op = g_fallback.load();
        if (frame > size_ || !op || !op->seek || op->seek(ext_, frame) != 0) return ERR_SEEK;
        pos_ = frame;
        return 0;
    }
    if (!f_) return ERR_MODE;
    if (frame > size_) return ERR_SEEK;
    if (fseek(f_, data_offset_ + static_cast<long>(frame) * block_align_, SEEK_SET) != 0) return ERR_SEEK;
    pos_ = frame;
    return 0;
}

int ImpulseFile::read(float* data, uint32_t frames) {
    if (ext_) {
        const ImpulseOpener* op = g_fallback.load();
        if (!op || !op->read) return ERR_READ;
        if (frames > size_ - pos_) frames = size_ - pos_;
        if (!frames) return 0;
        const int got = op->read(ext_, data, frames);
        if (got < 0) return ERR_READ;
        pos_ += static_cast<uint32_t>(got);
        return got;
    }
    if (!f_) return ERR_MODE;
    if (frames > size_ - pos_) frames = size_ - pos_;
    if (!frames) return 0;
    std::vector<unsigned char> raw(static_cast<size_t>(frames) * block_align_);
    const size_t got = fread(raw.data(), static_cast<size_t>(block_align_), frames, f_);
    pos_ += static_cast<uint32_t>(got);
    const size_t n = got * static_cast<size_t>(chan_);
    unsigned char* p = raw.data();
    for (size_t i = 0; i < n; ++i, p += bytes_per_sample_) {
        if (big_endian_) {                                          // to little-endian, in place
            for (int a = 0, b = bytes_per_sample_ - 1; a < b; ++a, --b) { const unsigned char t = p[a]; p[a] = p[b]; p[b] = t; }
        }
        float v = 0.0f;
        switch (form_) {
            case FORM_8BIT:
                v = signed8_ ? static_cast<float>(static_cast<signed char>(p[0])) / 128.0f
                             : static_cast<float>(static_cast<int>(p[0]) - 128) / 128.0f;
                break;
            case FORM_16BIT: v = static_cast<float>(static_cast<int16_t>(le16(p))) / 32768.0f; break;
            case FORM_24BIT: {
                const int32_t s = static_cast<int32_t>((static_cast<uint32_t>(p[0]) << 8) | (static_cast<uint32_t>(p[1]) << 16) |
                                                       (static_cast<uint32_t>(p[2]) << 24));
                v = static_cast<float>(s) / 2147483648.0f;
                break;
            }
            case FORM_32BIT: v = static_cast<float>(static_cast<int32_t>(le32(p))) / 2147483648.0f; break;
            case FORM_FLOAT: memcpy(&v, p, 4); break;
            case FORM_DOUBLE: { double d; memcpy(&d, p, 8); v = static_cast<float>(d); break; }
            default: break;
        }
        data[i] = v;
    }
    return static_cast<int>(got);
}

}  // namespace folve
