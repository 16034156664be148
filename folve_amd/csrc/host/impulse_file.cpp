#include "impulse_file.h"

#include <string.h>

#include <vector>

namespace folve {

namespace {
uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t le16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
}  // namespace

ImpulseFile::ImpulseFile() : f_(nullptr) { reset(); }
ImpulseFile::~ImpulseFile() { close(); }

void ImpulseFile::reset() {
    f_ = nullptr;
    rate_ = chan_ = 0;
    form_ = FORM_OTHER;
    bytes_per_sample_ = block_align_ = 0;
    size_ = pos_ = 0;
    data_offset_ = 0;
}

int ImpulseFile::close() {
    if (f_) fclose(f_);
    reset();
    return 0;
}

int ImpulseFile::open_read(const char* name) {
    if (f_) return ERR_MODE;
    reset();
    FILE* f = fopen(name, "rb");
    if (!f) return ERR_OPEN;
    unsigned char hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) != 0 || memcmp(hdr + 8, "WAVE", 4) != 0) {
        fclose(f);
        return ERR_TYPE;
    }
    int tag = 0, bits = 0;
    bool have_fmt = false;
    for (;;) {
        unsigned char ck[8];
        if (fread(ck, 1, 8, f) != 8) { fclose(f); return ERR_DATA; }
        const uint32_t len = le32(ck + 4);
        if (memcmp(ck, "fmt ", 4) == 0) {
            unsigned char b[40];
            const uint32_t n = len < sizeof(b) ? len : static_cast<uint32_t>(sizeof(b));
            if (len < 16 || fread(b, 1, n, f) != n) { fclose(f); return ERR_DATA; }
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
            if (!have_fmt || chan_ < 1 || block_align_ < 1) { fclose(f); return ERR_DATA; }
            bytes_per_sample_ = bits / 8;
            if (tag == 1) {
                switch (bytes_per_sample_) {
                    case 1: form_ = FORM_8BIT; break;
                    case 2: form_ = FORM_16BIT; break;
                    case 3: form_ = FORM_24BIT; break;
                    case 4: form_ = FORM_32BIT; break;
                    default: fclose(f); return ERR_FORM;
                }
            } else if (tag == 3 && bytes_per_sample_ == 4) {
                form_ = FORM_FLOAT;
            } else if (tag == 3 && bytes_per_sample_ == 8) {
                form_ = FORM_DOUBLE;
            } else {
                fclose(f);
                return ERR_FORM;
            }
            if (block_align_ != bytes_per_sample_ * chan_) { fclose(f); return ERR_FORM; }
            data_offset_ = ftell(f);
            fseek(f, 0, SEEK_END);
            const long avail = ftell(f) - data_offset_;
            fseek(f, data_offset_, SEEK_SET);
            uint32_t bytes = len;
            if (avail >= 0 && static_cast<long>(bytes) > avail) bytes = static_cast<uint32_t>(avail);
            size_ = bytes / static_cast<uint32_t>(block_align_);
            pos_ = 0;
            f_ = f;
            return ERR_NONE;
        } else {
            if (fseek(f, static_cast<long>(len) + (len & 1), SEEK_CUR) != 0) { fclose(f); return ERR_DATA; }
        }
    }
}

int ImpulseFile::seek(uint32_t frame) {
    if (!f_) return ERR_MODE;
    if (frame > size_) return ERR_SEEK;
    if (fseek(f_, data_offset_ + static_cast<long>(frame) * block_align_, SEEK_SET) != 0) return ERR_SEEK;
    pos_ = frame;
    return 0;
}

int ImpulseFile::read(float* data, uint32_t frames) {
    if (!f_) return ERR_MODE;
    if (frames > size_ - pos_) frames = size_ - pos_;
    if (!frames) return 0;
    std::vector<unsigned char> raw(static_cast<size_t>(frames) * block_align_);
    const size_t got = fread(raw.data(), static_cast<size_t>(block_align_), frames, f_);
    pos_ += static_cast<uint32_t>(got);
    const size_t n = got * static_cast<size_t>(chan_);
    const unsigned char* p = raw.data();
    for (size_t i = 0; i < n; ++i, p += bytes_per_sample_) {
        float v = 0.0f;
        switch (form_) {
            case FORM_8BIT: v = static_cast<float>(static_cast<int>(p[0]) - 128) / 128.0f; break;
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
