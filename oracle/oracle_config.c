/* oracle/oracle_config.c — TEST INFRASTRUCTURE ONLY. See oracle_config.h. */
#define _GNU_SOURCE
#include "oracle_config.h"

#include <ctype.h>
#include <libgen.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define OC_BSIZE 0x4000   /* zita-config.cc:43 */

/* ---- sstring (spec: zita-sstring.h:26-43) -------------------------------- */
int oc_sstring(const char *srce, char *dest, int size) {
    enum { Q_NONE = 0, Q_SINGLE = '\'', Q_DOUBLE = '"' };
    int quote = Q_NONE, escaped = 0, in = 0, out = 0;
    if (size < 0) return 0;
    for (;;) {
        if (out == size) break;                       /* no room for the terminator */
        unsigned char ch = (unsigned char)srce[in++];
        if (ch == '\t') ch = ' ';
        if (ch < 0x80 && iscntrl(ch)) {               /* incl. NUL and newline */
            if (quote || escaped) break;              /* unterminated quote / dangling escape */
            dest[out] = 0;
            return in - 1;
        }
        if (escaped) { dest[out++] = (char)ch; escaped = 0; continue; }
        if (ch == '\\') {
            if (quote == Q_SINGLE) dest[out++] = (char)ch; else escaped = 1;
            continue;
        }
        if (ch == Q_SINGLE || ch == Q_DOUBLE) {
            if (quote == ch) { dest[out] = 0; return in; }
            if (quote != Q_NONE || out > 0) break;    /* stray quote */
            quote = ch;
            continue;
        }
        if (ch == ' ') {
            if (quote) { dest[out++] = ' '; continue; }
            if (out > 0) { dest[out] = 0; return in - 1; }
            continue;                                  /* leading blank */
        }
        dest[out++] = (char)ch;
    }
    dest[0] = 0;
    return 0;
}

/* ---- /convolver/new (zita-fconfig.cc:38-97) ------------------------------ */
int oc_fragm_for_size(unsigned int size) {
    unsigned int fragm = OC_MAXQUANT;
    while (fragm > OC_MINPART && fragm >= 2 * size) fragm /= 2;
    return (int)fragm;
}

static int convnew(oc_zita_config *cfg, const char *line) {
    unsigned int ninp = (unsigned)cfg->ninp, nout = (unsigned)cfg->nout, part = 0, size = (unsigned)cfg->size;
    float dens = 0.0f;
    const int r = sscanf(line, "%u %u %u %u %f", &ninp, &nout, &part, &size, &dens);
    cfg->ninp = (int)ninp; cfg->nout = (int)nout; cfg->size = (int)size;
    if (r < 4) return OC_ERR_PARAM;
    if (r < 5) dens = 0.0f;
    if (cfg->ninp == 0 || cfg->ninp > OC_MAXINP) return OC_ERR_OTHER;
    if (cfg->nout == 0 || cfg->nout > OC_MAXOUT) return OC_ERR_OTHER;
    if (cfg->size > OC_MAXSIZE) return OC_ERR_OTHER;
    if (dens < 0.0f || dens > 1.0f) return OC_ERR_OTHER;
    cfg->fragm = oc_fragm_for_size((unsigned)cfg->size);   /* `part` is parsed and ignored */
    if (oc_configure(cfg->convproc, cfg->ninp, cfg->nout, cfg->size,
                     cfg->fragm, cfg->fragm, cfg->fragm, dens)) return OC_ERR_OTHER;
    return 0;
}

static int check_inout(const oc_zita_config *cfg, int ip, int op) {
    if (!cfg->size) return OC_ERR_NOCONV;
    if (ip < 1 || ip > cfg->ninp) return OC_ERR_IONUM;
    if (op < 1 || op > cfg->nout) return OC_ERR_IONUM;
    return 0;
}

/* ---- WAV reader (stands in for libsndfile behind zita-audiofile.cc) ------ */
static uint32_t rd_u32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd_u16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int oc_wav_load(const char *path, oc_wav *w) {
    memset(w, 0, sizeof(*w));
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    unsigned char hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) { fclose(f); return -2; }
    int fmt = 0, bits = 0, chan = 0, rate = 0, align = 0, have_fmt = 0;
    for (;;) {
        unsigned char ck[8];
        if (fread(ck, 1, 8, f) != 8) { fclose(f); return -3; }
        uint32_t len = rd_u32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            unsigned char b[40];
            uint32_t n = len < 40 ? len : 40;
            if (len < 16 || fread(b, 1, n, f) != n) { fclose(f); return -3; }
            fmt = rd_u16(b); chan = rd_u16(b + 2); rate = (int)rd_u32(b + 4);
            align = rd_u16(b + 12); bits = rd_u16(b + 14);
            if (fmt == 0xFFFE && n >= 26) fmt = rd_u16(b + 24);   /* WAVE_FORMAT_EXTENSIBLE sub-format */
            if (len > n) fseek(f, (long)(len - n), SEEK_CUR);
            if (len & 1) fseek(f, 1, SEEK_CUR);
            have_fmt = 1;
        } else if (!memcmp(ck, "data", 4)) {
            if (!have_fmt || chan < 1 || align < 1) { fclose(f); return -3; }
            const int bps = bits / 8;
            if (!((fmt == 1 && (bps >= 1 && bps <= 4)) || (fmt == 3 && (bps == 4 || bps == 8)))) { fclose(f); return -4; }
            long here = ftell(f);
            fseek(f, 0, SEEK_END);
            long avail = ftell(f) - here;
            fseek(f, here, SEEK_SET);
            if ((long)len > avail) len = (uint32_t)avail;
            const unsigned int frames = len / (unsigned)align;
            unsigned char *raw = (unsigned char *)malloc((size_t)frames * (size_t)align + 1);
            float *data = (float *)malloc(sizeof(float) * (size_t)frames * (size_t)chan + sizeof(float));
            if (!raw || !data) { free(raw); free(data); fclose(f); return -5; }
            if (fread(raw, (size_t)align, frames, f) != frames) { free(raw); free(data); fclose(f); return -3; }
            const size_t n = (size_t)frames * (size_t)chan;
            for (size_t i = 0; i < n; ++i) {
                const unsigned char *p = raw + i * (size_t)bps;
                float v;
                if (fmt == 3) {
                    if (bps == 4) { memcpy(&v, p, 4); } else { double d; memcpy(&d, p, 8); v = (float)d; }
                } else if (bps == 1) {
                    v = (float)((int)p[0] - 128) / 128.0f;
                } else if (bps == 2) {
                    v = (float)(int16_t)rd_u16(p) / 32768.0f;
                } else if (bps == 3) {
                    int32_t s = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
                    v = (float)s / 2147483648.0f;
                } else {
                    v = (float)(int32_t)rd_u32(p) / 2147483648.0f;
                }
                data[i] = v;
            }
            free(raw);
            fclose(f);
            w->rate = rate; w->chan = chan; w->frames = frames; w->data = data;
            return 0;
        } else {
            fseek(f, (long)(len + (len & 1)), SEEK_CUR);
        }
    }
}

void oc_wav_free(oc_wav *w) { free(w->data); w->data = NULL; }

/* ---- /impulse/read (zita-config.cc:55-177) ------------------------------- */
static int readfile(oc_zita_config *cfg, const char *line, const char *cdir) {
    unsigned int ip1, op1, delay, offset, length, ichan;
    float gain;
    int n = 0;
    char file[1024], path[3100];
    if (sscanf(line, "%u %u %f %u %u %u %u %n", &ip1, &op1, &gain, &delay, &offset, &length, &ichan, &n) != 7)
        return OC_ERR_PARAM;
    if (!oc_sstring(line + n, file, 1024)) return OC_ERR_PARAM;
    /* cfg->latency is always 0 in folve (sound-processor.cc:37): latency branch dead */
    int err = check_inout(cfg, (int)ip1, (int)op1);
    if (err) return err;
    if (file[0] == '/') snprintf(path, sizeof(path), "%s", file);
    else snprintf(path, sizeof(path), "%s/%s", cdir, file);

    oc_wav wav;
    if (oc_wav_load(path, &wav)) return OC_ERR_OTHER;
    /* rate mismatch is only logged (zita-config.cc:108-112) */
    const unsigned int nchan = (unsigned)wav.chan;
    if (ichan < 1 || ichan > nchan) { oc_wav_free(&wav); return OC_ERR_OTHER; }
    if (offset && offset > wav.frames) { oc_wav_free(&wav); return OC_ERR_OTHER; } /* sf_seek fails */
    if (!length) length = wav.frames - offset;
    if (length > (unsigned)cfg->size - delay) length = (unsigned)cfg->size - delay;  /* "Data truncated" */

    float *buff = (float *)malloc(sizeof(float) * OC_BSIZE * (size_t)nchan);
    if (!buff) { oc_wav_free(&wav); return OC_ERR_ALLOC; }
    unsigned int pos = offset;
    while (length) {
        unsigned int nfram = length > OC_BSIZE ? OC_BSIZE : length;
        if (pos + nfram > wav.frames) nfram = wav.frames > pos ? wav.frames - pos : 0;
        if (!nfram) break;   /* reference would spin forever on a short file; stop instead */
        memcpy(buff, wav.data + (size_t)pos * nchan, sizeof(float) * (size_t)nfram * nchan);
        pos += nfram;
        float *p = buff + ichan - 1;
        for (unsigned int i = 0; i < nfram; ++i) p[(size_t)i * nchan] *= gain;    /* float32 gain, cc:161-162 */
        if (oc_impdata_create(cfg->convproc, (int)ip1 - 1, (int)op1 - 1, (int)nchan, p, (int)delay, (int)(delay + nfram))) {
            free(buff); oc_wav_free(&wav); return OC_ERR_ALLOC;
        }
        delay += nfram;
        length -= nfram;
    }
    free(buff);
    oc_wav_free(&wav);
    return 0;
}

/* ---- /impulse/dirac (zita-config.cc:180-209) ----------------------------- */
static int impdirac(oc_zita_config *cfg, const char *line) {
    int ip1, op1, delay;
    float gain;
    if (sscanf(line, "%u %u %f %u", (unsigned *)&ip1, (unsigned *)&op1, &gain, (unsigned *)&delay) != 4) return OC_ERR_PARAM;
    int stat = check_inout(cfg, ip1, op1);
    if (stat) return stat;
    if (delay < cfg->latency) return 0;
    delay -= cfg->latency;
    if (delay < cfg->size) {
        if (oc_impdata_create(cfg->convproc, ip1 - 1, op1 - 1, 1, &gain, delay, delay + 1)) return OC_ERR_ALLOC;
    }
    return 0;
}

/* ---- /impulse/hilbert (zita-config.cc:212-259) --------------------------- */
static int imphilbert(oc_zita_config *cfg, const char *line) {
    unsigned int ip1, op1, delay, length;
    float gain;
    if (sscanf(line, "%u %u %f %u %u", &ip1, &op1, &gain, &delay, &length) != 5) return OC_ERR_PARAM;
    int stat = check_inout(cfg, (int)ip1, (int)op1);
    if (stat) return stat;
    if (length < 64 || length > 65536) return OC_ERR_PARAM;
    const unsigned int k = (unsigned)cfg->latency;
    if (delay < k + length / 2) return 0;              /* "Hilbert impulse removed" */
    delay -= k + length / 2;
    float *hdata = (float *)calloc(length, sizeof(float));
    if (!hdata) return OC_ERR_ALLOC;
    gain *= 2 / M_PI;                                    /* double product rounded to float */
    const unsigned int h = length / 2;
    for (unsigned int i = 1; i < h; i += 2) {
        float v = gain / i;
        float w = 0.43f + 0.57f * cosf(i * M_PI / h);   /* argument computed in double, as in the reference */
        v *= w;
        hdata[h + i] = -v;
        hdata[h - i] = v;
    }
    stat = oc_impdata_create(cfg->convproc, (int)ip1 - 1, (int)op1 - 1, 1, hdata, (int)delay, (int)(delay + length));
    free(hdata);
    return stat ? OC_ERR_ALLOC : 0;
}

/* ---- /impulse/copy (zita-config.cc:262-279) ------------------------------ */
static int impcopy(oc_zita_config *cfg, const char *line) {
    unsigned int ip1, op1, ip2, op2;
    if (sscanf(line, "%u %u %u %u", &ip1, &op1, &ip2, &op2) != 4) return OC_ERR_PARAM;
    int stat = check_inout(cfg, (int)ip1, (int)op1) | check_inout(cfg, (int)ip2, (int)op2);
    if (stat) return stat;
    if (ip1 == ip2 && op1 == op2) return OC_ERR_PARAM;
    if (oc_impdata_copy(cfg->convproc, (int)ip2 - 1, (int)op2 - 1, (int)ip1 - 1, (int)op1 - 1)) return OC_ERR_ALLOC;
    return 0;
}

/* ---- config() (zita-config.cc:282-378) ----------------------------------- */
int oc_config(oc_zita_config *cfg, const char *config_file) {
    FILE *F = fopen(config_file, "r");
    if (!F) return -1;
    char line[1024], cdir[2048];
    {
        char *copy = strdup(config_file);
        snprintf(cdir, sizeof(cdir), "%s", dirname(copy));
        free(copy);
    }
    cfg->config_file = config_file;
    int stat = 0;
    while (!stat && fgets(line, 1024, F)) {
        char *p = line;
        if (*p != '/') {
            while (isspace((unsigned char)*p)) p++;
            if (*p > ' ' && *p != '#') { stat = OC_ERR_SYNTAX; break; }   /* plain char: bytes >= 0x80 pass, as in the reference */
            continue;
        }
        char *q = p;
        while (*q >= ' ' && !isspace((unsigned char)*q)) q++;
        if (*q) { *q++ = 0; while (*q >= ' ' && isspace((unsigned char)*q)) q++; }

        if (!strcmp(p, "/cd")) {
            char tmp[1024];
            if (oc_sstring(q, tmp, 1024) == 0) stat = OC_ERR_PARAM;
            if (tmp[0] == '/') snprintf(cdir, sizeof(cdir), "%s", tmp);
            else { strncat(cdir, "/", sizeof(cdir) - strlen(cdir) - 1); strncat(cdir, tmp, sizeof(cdir) - strlen(cdir) - 1); }
        }
        else if (!strcmp(p, "/convolver/new"))   stat = convnew(cfg, q);
        else if (!strcmp(p, "/impulse/read"))    stat = readfile(cfg, q, cdir);
        else if (!strcmp(p, "/impulse/dirac"))   stat = impdirac(cfg, q);
        else if (!strcmp(p, "/impulse/hilbert")) stat = imphilbert(cfg, q);
        else if (!strcmp(p, "/impulse/copy"))    stat = impcopy(cfg, q);
        else if (!strcmp(p, "/input/name"))      stat = 0;    /* zita-fconfig.cc:100-109 */
        else if (!strcmp(p, "/output/name"))     stat = 0;
        else stat = OC_ERR_COMMAND;
    }
    fclose(F);
    if (stat == OC_ERR_OTHER) stat = 0;    /* zita-config.cc:345 */
    return stat;
}

/* ---- SoundProcessor (sound-processor.cc:34-145) -------------------------- */
struct oc_sound_processor {
    oc_zita_config cfg;
    char *config_path;
    float *buffer;
    int input_pos, output_pos;
    float max_out;
};

static oc_sound_processor *sp_finish(oc_sound_processor *sp) {
    const int ch = sp->cfg.ninp > sp->cfg.nout ? sp->cfg.ninp : sp->cfg.nout;
    sp->buffer = (float *)calloc((size_t)sp->cfg.fragm * (size_t)ch, sizeof(float));
    oc_sp_reset(sp);
    return sp;
}

oc_sound_processor *oc_sp_create(const char *config_file, int samplerate, int channels) {
    oc_sound_processor *sp = (oc_sound_processor *)calloc(1, sizeof(*sp));
    sp->config_path = strdup(config_file);
    sp->cfg.fsamp = samplerate;
    sp->cfg.ninp = channels;
    sp->cfg.nout = channels;
    sp->cfg.convproc = oc_convproc_new();
    if (oc_config(&sp->cfg, sp->config_path) != 0
        || oc_inpdata(sp->cfg.convproc, sp->cfg.ninp - 1) == NULL
        || oc_outdata(sp->cfg.convproc, sp->cfg.nout - 1) == NULL) {
        oc_convproc_delete(sp->cfg.convproc);   /* the reference leaks it here (q2) */
        free(sp->config_path);
        free(sp);
        return NULL;
    }
    return sp_finish(sp);
}

oc_sound_processor *oc_sp_wrap(oc_convproc *conv, int fragm, int ninp, int nout) {
    oc_sound_processor *sp = (oc_sound_processor *)calloc(1, sizeof(*sp));
    sp->cfg.convproc = conv; sp->cfg.fragm = fragm; sp->cfg.ninp = ninp; sp->cfg.nout = nout;
    return sp_finish(sp);
}

void oc_sp_delete(oc_sound_processor *sp) {
    if (!sp) return;
    oc_stop_process(sp->cfg.convproc);
    oc_cleanup(sp->cfg.convproc);
    oc_convproc_delete(sp->cfg.convproc);
    free(sp->buffer);
    free(sp->config_path);
    free(sp);
}

int oc_sp_fill_buffer(oc_sound_processor *sp, const float *src, int frames_available) {
    const int needed = sp->cfg.fragm - sp->input_pos;
    sp->output_pos = -1;
    const int r = frames_available < needed ? frames_available : needed;
    memcpy(sp->buffer + (size_t)sp->input_pos * (size_t)sp->cfg.ninp, src, sizeof(float) * (size_t)r * (size_t)sp->cfg.ninp);
    sp->input_pos += r;
    return r;
}

static void sp_process(oc_sound_processor *sp) {
    const int P = sp->cfg.fragm, ni = sp->cfg.ninp, no = sp->cfg.nout;
    const int missing = P - sp->input_pos;
    if (missing) memset(sp->buffer + (size_t)sp->input_pos * ni, 0, sizeof(float) * (size_t)missing * ni);
    for (int ch = 0; ch < ni; ++ch) {
        float *dest = oc_inpdata(sp->cfg.convproc, ch);
        /* The reference copies only input_pos_ frames (cc:108); the engine window
         * beyond is whatever it held.  Zero it: same emitted samples (causal). */
        for (int j = 0; j < sp->input_pos; ++j) dest[j] = sp->buffer[(size_t)j * ni + ch];
        for (int j = sp->input_pos; j < P; ++j) dest[j] = 0.0f;
    }
    oc_process(sp->cfg.convproc);
    for (int ch = 0; ch < no; ++ch) {
        const float *source = oc_outdata(sp->cfg.convproc, ch);
        for (int j = 0; j < sp->input_pos; ++j) {
            sp->buffer[(size_t)j * no + ch] = source[j];
            if (source[j] > sp->max_out) sp->max_out = source[j];   /* signed compare, cc:120-123 */
        }
    }
    sp->output_pos = 0;
}

void oc_sp_write_processed(oc_sound_processor *sp, float *dst, int sample_count) {
    if (sp->output_pos < 0) sp_process(sp);
    memcpy(dst, sp->buffer + (size_t)sp->output_pos * (size_t)sp->cfg.nout,
           sizeof(float) * (size_t)sample_count * (size_t)sp->cfg.nout);
    sp->output_pos += sample_count;
    if (sp->output_pos == sp->cfg.fragm) sp->input_pos = 0;
}

int oc_sp_is_input_buffer_complete(const oc_sound_processor *sp) { return sp->cfg.fragm == sp->input_pos; }
int oc_sp_pending_writes(const oc_sound_processor *sp) { return sp->output_pos >= 0 ? sp->cfg.fragm - sp->output_pos : 0; }
float oc_sp_max_output_value(const oc_sound_processor *sp) { return sp->max_out; }
int oc_sp_input_channels(const oc_sound_processor *sp) { return sp->cfg.ninp; }
int oc_sp_output_channels(const oc_sound_processor *sp) { return sp->cfg.nout; }
int oc_sp_fragm(const oc_sound_processor *sp) { return sp->cfg.fragm; }
oc_convproc *oc_sp_convproc(oc_sound_processor *sp) { return sp->cfg.convproc; }

void oc_sp_reset(oc_sound_processor *sp) {
    oc_reset(sp->cfg.convproc);
    sp->input_pos = 0;
    sp->output_pos = -1;
    sp->max_out = 0.0f;
    oc_start_process(sp->cfg.convproc, 0, 0);
}

long oc_sp_run(oc_sound_processor *sp, const float *in, long nframes, float *out) {
    long done = 0;
    const int ni = sp->cfg.ninp, no = sp->cfg.nout;
    while (done < nframes) {
        long left = nframes - done;
        int r = oc_sp_fill_buffer(sp, in + (size_t)done * ni, left > 1 << 30 ? 1 << 30 : (int)left);
        if (r == 0) break;
        oc_sp_write_processed(sp, out + (size_t)done * no, r);
        done += r;
    }
    return done;
}

/* ---- CPU baseline driver -------------------------------------------------- */
typedef struct { int first, stride, nstreams, nblocks; oc_sound_processor **sp; float *in; float *out; } bench_arg;

static void *bench_worker(void *vp) {
    bench_arg *a = (bench_arg *)vp;
    for (int b = 0; b < a->nblocks; ++b) {
        for (int s = a->first; s < a->nstreams; s += a->stride) {
            oc_sound_processor *sp = a->sp[s];
            const int P = sp->cfg.fragm;
            oc_sp_fill_buffer(sp, a->in, P);
            oc_sp_write_processed(sp, a->out + (size_t)a->first * P * sp->cfg.nout, P);
        }
    }
    return NULL;
}

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s; }

double oc_bench_streams(int nstreams, int nblocks, int nthreads, int ninp, int nout, int size, unsigned seed) {
    const int fragm = oc_fragm_for_size((unsigned)size);
    oc_sound_processor **sps = (oc_sound_processor **)calloc((size_t)nstreams, sizeof(*sps));
    float *h = (float *)malloc(sizeof(float) * (size_t)size);
    for (int i = 0; i < size; ++i) h[i] = ((float)(lcg(&seed) >> 8) / 8388608.0f - 1.0f) / sqrtf((float)size);
    const int npaths = ninp < nout ? ninp : nout;
    for (int s = 0; s < nstreams; ++s) {
        oc_convproc *c = oc_convproc_new();
        oc_configure(c, ninp, nout, size, fragm, fragm, fragm, 0.0f);
        for (int p = 0; p < npaths; ++p) oc_impdata_create(c, p, p, 1, h, 0, size);
        sps[s] = oc_sp_wrap(c, fragm, ninp, nout);
    }
    float *in = (float *)malloc(sizeof(float) * (size_t)fragm * (size_t)ninp);
    for (int i = 0; i < fragm * ninp; ++i) in[i] = (float)(lcg(&seed) >> 8) / 8388608.0f - 1.0f;
    float *out = (float *)malloc(sizeof(float) * (size_t)fragm * (size_t)nout * (size_t)nthreads);
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    bench_arg *args = (bench_arg *)calloc((size_t)nthreads, sizeof(bench_arg));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; ++t) {
        args[t] = (bench_arg){ t, nthreads, nstreams, nblocks, sps, in, out };
        pthread_create(&th[t], NULL, bench_worker, &args[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int s = 0; s < nstreams; ++s) oc_sp_delete(sps[s]);
    free(sps); free(h); free(in); free(out); free(th); free(args);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
