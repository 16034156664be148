// zita_ref.cpp — the REAL libzita-convolver, driven exactly as folve drives it, where a box has it.
//
// The convolution arithmetic of folve's hot path lives in libzita-convolver (/root/reference/Makefile:14), which is in
// neither the reference tree nor this image: the oracle restates it and its parity is therefore "unpinned".  This program
// is the conditional pin: built only where <zita-convolver.h> and the library exist (tests/test_zita_gpu.py and bench.py
// try, and say why not otherwise), it configures a Convproc as /root/reference/zita-fconfig.cc:74-94 does — one level,
// quantum = minpart = maxpart = fragm, options 0 — fills it as zita-config.cc:163 does (impdata_create, step 1), and runs
// blocks as /root/reference/sound-processor.cc:98-127 does (planar copy in, process(), planar copy out).  No zita code is
// in this repository; nothing here stands in for its headers.
//
//   zita_ref run   <channels> <size> <taps.f32> <in.f32> <out.f32>   diagonal paths; taps [channels][size]; PCM interleaved
//   zita_ref bench <channels> <size> <streams> <blocks> <threads>     prints the seconds the timed blocks took
// exit code 77: built without zita-convolver.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(__has_include)
#if __has_include(<zita-convolver.h>)
#define HAVE_ZITA 1
#endif
#endif

#ifdef HAVE_ZITA
#include <zita-convolver.h>

#include <chrono>
#include <thread>
#include <vector>

static int fragm_for(unsigned size) {                       // zita-fconfig.cc:74-77
    unsigned fragm = Convproc::MAXQUANT;
    while (fragm > Convproc::MINPART && fragm >= 2 * size) fragm /= 2;
    return (int)fragm;
}

static Convproc* make(int ch, int size, const float* taps, int* fragm) {
    Convproc* c = new Convproc();
    *fragm = fragm_for((unsigned)size);
    c->set_options(0);
#if ZITA_CONVOLVER_MAJOR_VERSION >= 4
    if (c->configure(ch, ch, size, *fragm, *fragm, *fragm, 0.0f)) { delete c; return NULL; }
#else
    c->set_density(0.0f);
    if (c->configure(ch, ch, size, *fragm, *fragm, *fragm)) { delete c; return NULL; }
#endif
    for (int k = 0; k < ch; ++k)
        if (c->impdata_create(k, k, 1, const_cast<float*>(taps + (size_t)k * size), 0, size)) { delete c; return NULL; }
    c->start_process(0, 0);
    return c;
}

static void block(Convproc* c, int ch, int fragm, const float* in, int valid, float* out) {   // sound-processor.cc:98-127
    for (int k = 0; k < ch; ++k) {
        float* dest = c->inpdata(k);
        for (int j = 0; j < valid; ++j) dest[j] = in[(size_t)j * ch + k];
        for (int j = valid; j < fragm; ++j) dest[j] = 0.f;
    }
    c->process();
    for (int k = 0; k < ch; ++k) {
        const float* src = c->outdata(k);
        for (int j = 0; j < valid; ++j) out[(size_t)j * ch + k] = src[j];
    }
}

static std::vector<float> slurp(const char* path) {
    std::vector<float> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(float));
    if (fread(v.data(), sizeof(float), v.size(), f) != v.size()) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc >= 7 && !strcmp(argv[1], "run")) {
        const int ch = atoi(argv[2]), size = atoi(argv[3]);
        const std::vector<float> taps = slurp(argv[4]), x = slurp(argv[5]);
        if ((long long)taps.size() != (long long)ch * size || x.empty()) { fprintf(stderr, "bad input files\n"); return 2; }
        int fragm = 0;
        Convproc* c = make(ch, size, taps.data(), &fragm);
        if (!c) { fprintf(stderr, "Convproc configure / impdata_create failed\n"); return 3; }
        const size_t frames = x.size() / ch;
        std::vector<float> y(x.size());
        for (size_t a = 0; a < frames; a += (size_t)fragm) {
            const int valid = (int)std::min<size_t>((size_t)fragm, frames - a);
            block(c, ch, fragm, x.data() + a * ch, valid, y.data() + a * ch);
        }
        c->stop_process();
        c->cleanup();
        delete c;
        FILE* f = fopen(argv[6], "wb");
        if (!f || fwrite(y.data(), sizeof(float), y.size(), f) != y.size()) return 4;
        fclose(f);
        printf("{\"zita_major\": %d, \"fragm\": %d, \"frames\": %zu}\n", ZITA_CONVOLVER_MAJOR_VERSION, fragm, frames);
        return 0;
    }
    if (argc >= 7 && !strcmp(argv[1], "bench")) {
        const int ch = atoi(argv[2]), size = atoi(argv[3]), streams = atoi(argv[4]), blocks = atoi(argv[5]), threads = atoi(argv[6]);
        std::vector<float> taps((size_t)ch * size);
        unsigned s = 3;
        for (float& v : taps) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.01f; }
        int fragm = 0;
        std::vector<Convproc*> cs;
        for (int i = 0; i < streams; ++i) { Convproc* c = make(ch, size, taps.data(), &fragm); if (!c) return 3; cs.push_back(c); }
        std::vector<float> in((size_t)fragm * ch), outb((size_t)fragm * ch * threads);
        for (float& v : in) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.0f - 1.0f; }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back([&, t] {
            float* out = outb.data() + (size_t)t * fragm * ch;
            for (int b = 0; b < blocks; ++b)
                for (int i = t; i < streams; i += threads) block(cs[(size_t)i], ch, fragm, in.data(), fragm, out);
        });
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"seconds\": %.6f, \"fragm\": %d, \"zita_major\": %d}\n", dt, fragm, ZITA_CONVOLVER_MAJOR_VERSION);
        for (Convproc* c : cs) { c->stop_process(); c->cleanup(); delete c; }
        return 0;
    }
    fprintf(stderr, "usage: zita_ref run <channels> <size> <taps.f32> <in.f32> <out.f32> | bench <channels> <size> <streams> <blocks> <threads>\n");
    return 2;
}
#else
int main() {
    puts("built without <zita-convolver.h>: the real library is not on this box");
    return 77;
}
#endif
