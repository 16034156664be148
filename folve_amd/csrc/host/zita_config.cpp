// zita_config.cpp — the jconvolver configuration loader folve uses, on top of the GPU engine's filter ABI.
//
// This file restates the grammar, the error paths and the log messages of folve's zita-config.cc / zita-fconfig.cc,
// which derive from config.cc of jconvolver 0.9.2: Copyright (C) 2006-2011 Fons Adriaensen <fons@linuxaudio.org>,
// modifications for folve Copyright (C) 2012 Henner Zeller <h.zeller@acm.org>; free software under the GNU General
// Public License, version 2 or (at your option) any later version.  This restatement is distributed under the same
// terms, WITHOUT ANY WARRANTY.  The impulse-file reader (impulse_file.cpp) and everything below fe_filter_* are original.
#include "zita_config.h"

#include <ctype.h>
#include <libgen.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "impulse_file.h"
#include "sstring.h"

namespace folve {

namespace {

LogFn g_log = nullptr;
const unsigned kChunkFrames = 0x4000;   // read granularity of /impulse/read (zita-config.cc:43)

int check_inout(const ZitaConfig* cfg, int ip, int op) {
    if (!cfg->size) return ERR_NOCONV;
    if (ip < 1 || ip > cfg->ninp) return ERR_IONUM;
    if (op < 1 || op > cfg->nout) return ERR_IONUM;
    return 0;
}

// /impulse/read <in> <out> <gain> <delay> <offset> <length> <chan> <file>
int readfile(ZitaConfig* cfg, const char* line, int lnum, const std::string& cdir) {
    unsigned int ip1, op1, delay, offset, length, ichan;
    float gain;
    int used = 0;
    char file[1024];
    if (sscanf(line, "%u %u %f %u %u %u %u %n", &ip1, &op1, &gain, &delay, &offset, &length, &ichan, &used) != 7)
        return ERR_PARAM;
    if (!sstring(line + used, file, 1024)) return ERR_PARAM;

    unsigned int k = static_cast<unsigned>(cfg->latency);
    if (k) {                                    // latency compensation (always 0 under folve)
        if (delay >= k) {
            delay -= k;
        } else {
            k -= delay;
            delay = 0;
            offset += k;
            Logf("%s:%d: First %d frames removed by latency compensation.", cfg->config_file, lnum, k);
        }
    }
    const int err = check_inout(cfg, static_cast<int>(ip1), static_cast<int>(op1));
    if (err) return err;

    const std::string path = (file[0] == '/') ? std::string(file) : cdir + "/" + file;
    ImpulseFile audio;
    if (cfg->on_impulse_file) cfg->on_impulse_file(cfg->on_impulse_user, path.c_str());
    if (audio.open_read(path.c_str())) {
        Logf("%s:%d: Unable to open '%s' >%s<.", cfg->config_file, lnum, path.c_str(), cdir.c_str());
        return ERR_OTHER;
    }
    if (audio.rate() != cfg->fsamp)
        Logf("%s:%d: Sample rate (%d) of '%s' does not match.", cfg->config_file, lnum, audio.rate(), path.c_str());

    const unsigned int nchan = static_cast<unsigned>(audio.chan());
    if (ichan < 1 || ichan > nchan) {
        Logf("%s:%d: Channel not available.", cfg->config_file, lnum);
        return ERR_OTHER;
    }
    if (offset && audio.seek(offset)) {
        Logf("%s:%d: Can't seek to offset.", cfg->config_file, lnum);
        return ERR_OTHER;
    }
    if (!length) length = audio.size() - offset;
    if (length > static_cast<unsigned>(cfg->size) - delay) {
        length = static_cast<unsigned>(cfg->size) - delay;
        Logf("%s:%d: Data truncated.", cfg->config_file, lnum);
    }

    std::vector<float> buff;
    try {
        buff.resize(static_cast<size_t>(kChunkFrames) * nchan);
    } catch (...) {
        return ERR_ALLOC;
    }
    while (length) {
        int nfram = static_cast<int>(length > kChunkFrames ? kChunkFrames : length);
        nfram = audio.read(buff.data(), static_cast<uint32_t>(nfram));
        if (nfram < 0) {
            Logf("%s:%d: Error reading file.", cfg->config_file, lnum);
            return ERR_OTHER;
        }
        if (nfram == 0) {
            // The reference keeps asking a file that has no more frames (it never
            // leaves this loop); a short file simply ends the impulse here.
            Logf("%s:%d: File shorter than requested length.", cfg->config_file, lnum);
            break;
        }
        float* p = buff.data() + ichan - 1;
        for (int i = 0; i < nfram; ++i) p[static_cast<size_t>(i) * nchan] *= gain;     // float32 gain
        if (fe_filter_add(cfg->filter, static_cast<int>(ip1) - 1, static_cast<int>(op1) - 1, static_cast<int>(nchan), p,
                          static_cast<int>(delay), static_cast<int>(delay) + nfram))
            return ERR_ALLOC;
        delay += static_cast<unsigned>(nfram);
        length -= static_cast<unsigned>(nfram);
    }
    return 0;
}

// /impulse/dirac <in> <out> <gain> <delay>
int impdirac(ZitaConfig* cfg, const char* line, int lnum) {
    unsigned int uip, uop, udelay;
    float gain;
    if (sscanf(line, "%u %u %f %u", &uip, &uop, &gain, &udelay) != 4) return ERR_PARAM;
    const int ip1 = static_cast<int>(uip), op1 = static_cast<int>(uop);
    int delay = static_cast<int>(udelay);
    const int stat = check_inout(cfg, ip1, op1);
    if (stat) return stat;
    if (delay < cfg->latency) {
        Logf("%s:%d: Dirac pulse removed: delay < latency.", cfg->config_file, lnum);
        return 0;
    }
    delay -= cfg->latency;
    if (delay < cfg->size) {
        if (fe_filter_add(cfg->filter, ip1 - 1, op1 - 1, 1, &gain, delay, delay + 1)) return ERR_ALLOC;
    }
    return 0;
}

// /impulse/hilbert <in> <out> <gain> <delay> <length>
int imphilbert(ZitaConfig* cfg, const char* line, int lnum) {
    unsigned int ip1, op1, delay, length;
    float gain;
    if (sscanf(line, "%u %u %f %u %u", &ip1, &op1, &gain, &delay, &length) != 5) return ERR_PARAM;
    const int stat = check_inout(cfg, static_cast<int>(ip1), static_cast<int>(op1));
    if (stat) return stat;
    if (length < 64 || length > 65536) return ERR_PARAM;
    const unsigned int lat = static_cast<unsigned>(cfg->latency);
    if (delay < lat + length / 2) {
        Logf("%s:%d: Hilbert impulse removed: delay < latency + length / 2.", cfg->config_file, lnum);
        return 0;
    }
    delay -= lat + length / 2;
    std::vector<float> taps(length, 0.0f);
    gain *= 2 / M_PI;                                   // double product, rounded back to float
    const unsigned int half = length / 2;
    for (unsigned int i = 1; i < half; i += 2) {        // odd taps only, antisymmetric about `half`
        float v = gain / i;
        const float w = 0.43f + 0.57f * cosf(i * M_PI / half);
        v *= w;
        taps[half + i] = -v;
        taps[half - i] = v;
    }
    if (fe_filter_add(cfg->filter, static_cast<int>(ip1) - 1, static_cast<int>(op1) - 1, 1, taps.data(),
                      static_cast<int>(delay), static_cast<int>(delay + length)))
        return ERR_ALLOC;
    return 0;
}

// /impulse/copy <in> <out> <from in> <from out>
int impcopy(ZitaConfig* cfg, const char* line, int) {
    unsigned int ip1, op1, ip2, op2;
    if (sscanf(line, "%u %u %u %u", &ip1, &op1, &ip2, &op2) != 4) return ERR_PARAM;
    const int stat = check_inout(cfg, static_cast<int>(ip1), static_cast<int>(op1)) |
                     check_inout(cfg, static_cast<int>(ip2), static_cast<int>(op2));
    if (stat) return stat;
    if (ip1 == ip2 && op1 == op2) return ERR_PARAM;
    if (fe_filter_link(cfg->filter, static_cast<int>(ip2) - 1, static_cast<int>(op2) - 1,
                       static_cast<int>(ip1) - 1, static_cast<int>(op1) - 1))
        return ERR_ALLOC;
    return 0;
}

}  // namespace

void SetLogHandler(LogFn fn) { g_log = fn; }

void Logf(const char* fmt, ...) {
    char buf[1400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g_log) {
        g_log(buf);
    } else {
        static const bool to_stderr = getenv("FOLVE_AMD_LOG") != nullptr;
        if (to_stderr) fprintf(stderr, "folve-amd: %s\n", buf);
    }
}

// /convolver/new <inputs> <outputs> <partition> <max size> [density]
int convnew(ZitaConfig* cfg, const char* line, int lnum) {
    unsigned int ninp = static_cast<unsigned>(cfg->ninp), nout = static_cast<unsigned>(cfg->nout), part = 0,
                 size = static_cast<unsigned>(cfg->size);
    float dens = 0.0f;
    const int r = sscanf(line, "%u %u %u %u %f", &ninp, &nout, &part, &size, &dens);
    cfg->ninp = static_cast<int>(ninp);       // the config's channel counts replace the file's
    cfg->nout = static_cast<int>(nout);
    cfg->size = static_cast<int>(size);
    if (r < 4) return ERR_PARAM;
    if (r < 5) dens = 0;
    if (cfg->ninp == 0 || cfg->ninp > FE_MAXINP) {
        Logf("%s:%d: Number of inputs (%d) is out of range.", cfg->config_file, lnum, cfg->ninp);
        return ERR_OTHER;
    }
    if (cfg->nout == 0 || cfg->nout > FE_MAXOUT) {
        Logf("%s:%d: Number of outputs (%d) is out of range.", cfg->config_file, lnum, cfg->nout);
        return ERR_OTHER;
    }
    if (cfg->size > FOLVE_MAXSIZE) {
        Logf("%s:%d: Convolver size (%d) is out of range.", cfg->config_file, lnum, cfg->size);
        return ERR_OTHER;
    }
    if (dens < 0.0f || dens > 1.0f) {
        Logf("%s:%d: Density parameter is out of range.", cfg->config_file, lnum);
        return ERR_OTHER;
    }
    // `part` is read and ignored; the block size follows from `size` alone.
    cfg->fragm = fe_fragm_for_size(static_cast<unsigned>(cfg->size));
    if (cfg->filter) {                       // a second /convolver/new: the engine is already configured
        Logf("Can't initialise convolution engine");
        return ERR_OTHER;
    }
    if (fe_filter_create(cfg->engine, cfg->ninp, cfg->nout, cfg->size, dens, &cfg->filter)) {
        Logf("Can't initialise convolution engine");
        return ERR_OTHER;
    }
    return 0;
}

int inpname(ZitaConfig*, const char*) { return 0; }
int outname(ZitaConfig*, const char*) { return 0; }

int config(ZitaConfig* cfg, const char* config_file) {
    FILE* F = fopen(config_file, "r");
    if (!F) {
        Logf("Can't open '%s' for reading", config_file);
        return -1;
    }
    std::string cdir;
    {
        char* copy = strdup(config_file);     // dirname() may modify its argument
        cdir = dirname(copy);
        free(copy);
    }
    cfg->config_file = config_file;
    int stat = 0, lnum = 0;
    char line[1024];
    while (!stat && fgets(line, 1024, F)) {
        lnum++;
        char* p = line;
        if (*p != '/') {
            while (isspace(static_cast<unsigned char>(*p))) p++;
            if (*p > ' ' && *p != '#') {       // plain char compare: bytes >= 0x80 count as blank
                stat = ERR_SYNTAX;
                break;
            }
            continue;
        }
        char* q = p;
        while (*q >= ' ' && !isspace(static_cast<unsigned char>(*q))) q++;
        if (*q) {
            *q++ = 0;
            while (*q >= ' ' && isspace(static_cast<unsigned char>(*q))) q++;
        }
        if (!strcmp(p, "/cd")) {
            char tmp[1024];
            if (sstring(q, tmp, 1024) == 0) stat = ERR_PARAM;
            if (tmp[0] == '/') cdir = tmp;
            else { cdir += "/"; cdir += tmp; }
        }
        else if (!strcmp(p, "/convolver/new"))   stat = convnew(cfg, q, lnum);
        else if (!strcmp(p, "/impulse/read"))    stat = readfile(cfg, q, lnum, cdir);
        else if (!strcmp(p, "/impulse/dirac"))   stat = impdirac(cfg, q, lnum);
        else if (!strcmp(p, "/impulse/hilbert")) stat = imphilbert(cfg, q, lnum);
        else if (!strcmp(p, "/impulse/copy"))    stat = impcopy(cfg, q, lnum);
        else if (!strcmp(p, "/input/name"))      stat = inpname(cfg, q);
        else if (!strcmp(p, "/output/name"))     stat = outname(cfg, q);
        else stat = ERR_COMMAND;
    }
    fclose(F);
    if (stat == ERR_OTHER) stat = 0;
    if (stat) {
        static const char* const what[] = {"", "", "Syntax error.", "Bad or missing parameters.", "Out of memory.",
                                           "Can't change directory.", "Unknown command.", "No convolver yet defined.",
                                           "Bad input or output number."};
        Logf("%s:%d: %s", config_file, lnum, (stat > 0 && stat <= ERR_IONUM) ? what[stat] : "Unknown error.");
    }
    return stat;
}

}  // namespace folve
