// CPU test of the whole HOST layer — folve::SoundProcessor with run-ahead, ProcessorPool, DeviceRouter, the combiner, the
// jconvolver loader — on a FAKE engine: this file implements the folve_engine.h entry points the host code calls with a
// direct-form FIR on the CPU (tiny filters, block size 64), tickets that complete a little later, and is linked with the
// real host sources.  It exists so that the block machine's ring / tail / ramp logic, the gapless hand-over and the pool
// are exercised without a GPU and under ThreadSanitizer / AddressSanitizer (tests/test_host_cpu.py builds and runs it).
// It is a test double for the C ABI, not a CPU path of the product: nothing in folve_amd/ links it.
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../folve_amd/csrc/host/batch_scheduler.h"
#include "../../folve_amd/csrc/host/device_router.h"
#include "../../folve_amd/csrc/host/processor_pool.h"
#include "../../folve_amd/csrc/host/sound_processor.h"
#include "../../include/folve_engine.h"

// ---------------------------------------------------------------- the fake engine
struct fe_engine { int device = 0; };
struct fe_filter {
    fe_engine* eng = nullptr;
    std::atomic<int> refs{1};
    int ninp = 0, nout = 0, size = 0, P = 0;
    bool committed = false;
    std::vector<std::vector<float>> taps;      // [inp * nout + out][size]
    std::vector<int> link;
};
struct fe_stream {
    fe_filter* f = nullptr;
    int max_blocks = 1;
    long long blocks_done = 0;
    std::vector<std::vector<float>> hist;      // [inp][size]: the last `size` input samples, newest last
    float peak_s = 0.f, peak_a = 0.f;
    const char* bound = nullptr;
    size_t bound_bytes = 0;
};
struct fe_ticket {
    std::chrono::steady_clock::time_point ready;
};

// fault knobs of the fake, per device: what a sick GPU does to the router (the `router` scenario below)
std::atomic<int> g_dead[16];            // every compute call, commit, stream open and probe on the device fails
std::atomic<int> g_no_create[16];       // fe_engine_create fails
std::atomic<int> g_probe_hangs_ms[16];  // fe_engine_probe sleeps this long before it answers

namespace {
thread_local std::string g_err;
std::mutex g_engine_mu;                         // the fake computes at submit time, one call at a time (as e->mu serialises the real one)
int fail(int rc, const char* m) { g_err = m; return rc; }

// `frames` interleaved frames of stream s: y_o[n] = sum_i sum_t h_io[t] x_i[n - t], state carried in s->hist; whole blocks
void fir(fe_stream* s, const float* in, long long frames, float* out, float* block_peaks) {
    const fe_filter* f = s->f;
    const int P = f->P;
    const long long padded = ((frames + P - 1) / P) * P;      // the short last block is zero-padded: time advances by whole blocks
    std::vector<float> y((size_t)frames * f->nout, 0.f);
    std::vector<float> x((size_t)f->size + (size_t)padded);
    for (int i = 0; i < f->ninp; ++i) {
        std::copy(s->hist[(size_t)i].begin(), s->hist[(size_t)i].end(), x.begin());
        for (long long n = 0; n < padded; ++n) x[(size_t)f->size + (size_t)n] = n < frames ? in[(size_t)n * f->ninp + i] : 0.f;
        for (int o = 0; o < f->nout; ++o) {
            int idx = i * f->nout + o;
            for (int g = 0; g < 64 && f->link[(size_t)idx] >= 0; ++g) idx = f->link[(size_t)idx];
            const std::vector<float>& h = f->taps[(size_t)idx];
            if (h.empty()) continue;
            std::vector<std::pair<int, float>> nz;            // (the test filters are a handful of diracs in up to 20 000 taps)
            for (int t = 0; t < f->size; ++t) if (h[(size_t)t] != 0.f) nz.push_back({t, h[(size_t)t]});
            for (long long n = 0; n < frames; ++n) {
                double acc = 0.0;
                for (const auto& th : nz) acc += (double)th.second * x[(size_t)f->size + (size_t)n - (size_t)th.first];
                y[(size_t)n * f->nout + o] += (float)acc;
            }
        }
        std::copy(x.end() - f->size, x.end(), s->hist[(size_t)i].begin());
    }
    const long long nb = padded / P;
    for (long long b = 0; b < nb; ++b) {
        float ps = 0.f, pa = 0.f;
        for (long long n = b * P; n < std::min<long long>((b + 1) * P, frames); ++n)
            for (int o = 0; o < f->nout; ++o) {
                const float v = y[(size_t)n * f->nout + o];
                ps = std::max(ps, v);
                pa = std::max(pa, fabsf(v));
            }
        if (block_peaks) { block_peaks[2 * b] = ps; block_peaks[2 * b + 1] = pa; }
        s->peak_s = std::max(s->peak_s, ps);
        s->peak_a = std::max(s->peak_a, pa);
    }
    memcpy(out, y.data(), y.size() * sizeof(float));          // (in == out is allowed: y is a copy)
    s->blocks_done += nb;
}
}  // namespace

extern "C" {
const char* fe_last_error(void) { return g_err.c_str(); }
int fe_device_count(void) { return 1; }
static bool dead(const fe_engine* e) { return e && e->device >= 0 && e->device < 16 && g_dead[e->device].load() != 0; }
int fe_engine_probe(fe_engine* e) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    const int ms = e->device < 16 ? g_probe_hangs_ms[e->device].load() : 0;
    if (ms) std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    return dead(e) ? fail(FE_ERR_DEVICE, "dead (fake)") : 0;
}
int fe_device_local_cpulist(int, char* buf, size_t size) { if (size) buf[0] = 0; return FE_ERR_UNSUPPORTED; }
int fe_fragm_for_size(unsigned int maxsize) {
    unsigned int fragm = FE_MAXQUANT;
    while (fragm > FE_MINPART && fragm >= 2 * maxsize) fragm /= 2;
    return (int)fragm;
}
int fe_engine_create(int device, void*, fe_engine** out) {
    if (device >= 0 && device < 16 && g_no_create[device].load()) return fail(FE_ERR_DEVICE, "no such device (fake)");
    *out = new fe_engine(); (*out)->device = device; return 0;
}
void fe_engine_destroy(fe_engine* e) { delete e; }
int fe_engine_device(const fe_engine* e) { return e ? e->device : -1; }
int fe_filter_create(fe_engine* e, int ninp, int nout, int maxsize, float, fe_filter** out) {
    if (ninp < 1 || nout < 1 || maxsize < 1 || maxsize > 65536) return fail(FE_ERR_PARAM, "fake engine: small filters only");
    fe_filter* f = new fe_filter();
    f->eng = e; f->ninp = ninp; f->nout = nout; f->size = maxsize; f->P = fe_fragm_for_size((unsigned)maxsize);
    f->taps.resize((size_t)ninp * nout);
    f->link.assign((size_t)ninp * nout, -1);
    *out = f;
    return 0;
}
int fe_filter_add(fe_filter* f, int inp, int out, int step, const float* data, int ind0, int ind1) {
    std::vector<float>& h = f->taps[(size_t)(inp * f->nout + out)];
    if (f->link[(size_t)(inp * f->nout + out)] >= 0) return 0;
    if (h.empty()) h.assign((size_t)f->size, 0.f);
    for (int t = ind0; t < ind1 && t < f->size; ++t) h[(size_t)t] += data[(size_t)(t - ind0) * step];
    return 0;
}
int fe_filter_link(fe_filter* f, int i1, int o1, int i2, int o2) {
    if (f->taps[(size_t)(i1 * f->nout + o1)].empty() || !f->taps[(size_t)(i2 * f->nout + o2)].empty()) return 0;
    f->link[(size_t)(i2 * f->nout + o2)] = i1 * f->nout + o1;
    return 0;
}
int fe_filter_commit(fe_filter* f) {
    if (dead(f->eng)) return fail(FE_ERR_DEVICE, "dead (fake)");
    f->committed = true; return 0;
}
void fe_filter_retain(fe_filter* f) { if (f) f->refs.fetch_add(1); }
void fe_filter_release(fe_filter* f) { if (f && f->refs.fetch_sub(1) == 1) delete f; }
int fe_filter_use_count(const fe_filter* f) { return f ? f->refs.load() : 0; }
int fe_stream_open(fe_filter* f, int max_blocks, fe_stream** out) {
    if (dead(f->eng)) return fail(FE_ERR_DEVICE, "dead (fake)");
    fe_stream* s = new fe_stream();
    s->f = f; s->max_blocks = max_blocks;
    s->hist.assign((size_t)f->ninp, std::vector<float>((size_t)f->size, 0.f));
    fe_filter_retain(f);
    *out = s;
    return 0;
}
int fe_host_alloc(size_t bytes, void** out) { *out = malloc(bytes); return *out ? 0 : FE_ERR_ALLOC; }
void fe_host_free(void* p) { free(p); }
int fe_stream_bind_host_buffer(fe_stream* s, void* buf, size_t bytes) { s->bound = (const char*)buf; s->bound_bytes = bytes; return 0; }
int fe_stream_reset(fe_stream* s) {
    for (auto& h : s->hist) std::fill(h.begin(), h.end(), 0.f);
    s->blocks_done = 0; s->peak_s = s->peak_a = 0.f;
    return 0;
}
int fe_stream_reset_peaks(fe_stream* s) { s->peak_s = s->peak_a = 0.f; return 0; }
void fe_stream_close(fe_stream* s) { if (!s) return; fe_filter_release(s->f); delete s; }
long long fe_stream_blocks_done(const fe_stream* s) { return s ? s->blocks_done : 0; }
int fe_stream_block_size(const fe_stream* s) { return s ? s->f->P : 0; }
int fe_stream_process(fe_stream* s, const float* in, int valid, float* out, float* ps, float* pa) {
    if (!s || valid < 1 || valid > s->f->P) return fail(FE_ERR_PARAM, "bad block");
    if (dead(s->f->eng)) return fail(FE_ERR_DEVICE, "dead (fake)");
    std::lock_guard<std::mutex> lk(g_engine_mu);
    fir(s, in, valid, out, nullptr);
    if (ps) *ps = s->peak_s;
    if (pa) *pa = s->peak_a;
    return 0;
}
int fe_batch_process(fe_stream* const* ss, int n, const float* const* in, const long long* nf, float* const* out, int) {
    if (n > 0 && dead(ss[0]->f->eng)) return fail(FE_ERR_DEVICE, "dead (fake)");
    std::lock_guard<std::mutex> lk(g_engine_mu);
    for (int i = 0; i < n; ++i) fir(ss[i], in[i], nf[i], out[i], nullptr);
    return 0;
}
int fe_stream_process_blocks(fe_stream* s, const float* in, long long nframes, float* out) {
    fe_stream* ss[1] = {s}; const float* ii[1] = {in}; float* oo[1] = {out}; long long nn[1] = {nframes};
    return fe_batch_process(ss, 1, ii, nn, oo, 0);
}
int fe_batch_submit_peaks(fe_stream* const* ss, int n, const float* const* in, const long long* nf, float* const* out,
                          float* const* block_peaks, fe_ticket** ticket) {
    static std::atomic<long long> calls{0};
    const long long k = calls.fetch_add(1);
    for (int i = 0; i < n; ++i) {                             // the real engine refuses buffers outside the bound memory
        const char* lo = ss[i]->bound;
        if (!lo || (const char*)in[i] < lo || (const char*)in[i] + nf[i] * ss[i]->f->ninp * 4 > lo + ss[i]->bound_bytes)
            return fail(FE_ERR_UNSUPPORTED, "buffer not bound");
    }
    if (k % 23 == 7) return fail(FE_ERR_DEVICE, "refused (fake)");     // nothing was enqueued
    if (n > 0 && dead(ss[0]->f->eng)) return fail(FE_ERR_DEVICE, "dead (fake)");
    {
        std::lock_guard<std::mutex> lk(g_engine_mu);
        for (int i = 0; i < n; ++i) fir(ss[i], in[i], nf[i], out[i], block_peaks ? block_peaks[i] : nullptr);
    }
    fe_ticket* t = new fe_ticket();
    t->ready = std::chrono::steady_clock::now() + std::chrono::microseconds(20 + (k * 13) % 150);
    *ticket = t;
    return 0;
}
int fe_ticket_done(fe_ticket* t) { return std::chrono::steady_clock::now() >= t->ready ? 1 : 0; }
int fe_ticket_wait(fe_ticket* t) { std::this_thread::sleep_until(t->ready); delete t; return 0; }
}

// ---------------------------------------------------------------- the test
namespace {
struct MemSource : folve::FrameSource {
    const std::vector<float>* d; size_t pos = 0; int ch; long long calls = 0;
    MemSource(const std::vector<float>* v, int c) : d(v), ch(c) {}
    int ReadFrames(float* dst, int frames) override {
        const size_t have = d->size() / ch - pos, n = std::min<size_t>((size_t)frames, have);
        memcpy(dst, d->data() + pos * ch, n * ch * sizeof(float));
        pos += n; ++calls;
        return (int)n;
    }
};
struct MemSink : folve::FrameSink {
    std::vector<float> d; int ch;
    explicit MemSink(int c) : ch(c) {}
    int WriteFrames(const float* src, int frames) override { d.insert(d.end(), src, src + (size_t)frames * ch); return frames; }
};

std::vector<float> direct(const std::vector<float>& x, int ch, const std::vector<std::vector<float>>& h /* per channel */) {
    const size_t n = x.size() / ch;
    std::vector<float> y(x.size(), 0.f);
    for (int c = 0; c < ch; ++c) {
        std::vector<std::pair<size_t, float>> nz;
        for (size_t t = 0; t < h[(size_t)c].size(); ++t) if (h[(size_t)c][t] != 0.f) nz.push_back({t, h[(size_t)c][t]});
        for (size_t i = 0; i < n; ++i) {
            double acc = 0.0;
            for (const auto& th : nz) if (th.first <= i) acc += (double)th.second * x[(i - th.first) * ch + c];
            y[i * ch + c] = (float)acc;
        }
    }
    return y;
}
double rms(const std::vector<float>& a, const std::vector<float>& b) {
    if (a.size() != b.size()) return 1e9;
    double e = 0.0;
    for (size_t i = 0; i < a.size(); ++i) e += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
    return sqrt(e / std::max<size_t>(1, a.size()));
}
}  // namespace

// ---------------------------------------------------------------- the router scenario (`host_fake <dir> router`)
// Eight GPU slots, one of them sick in the ways a GPU can be: never comes up, fails its calls, hangs its probe.
// What must hold (device_router.h; the reference's discard-and-recreate loop, processor-pool.cc:71-77, and its
// pass-through fallback when no processor can be had, folve-filesystem.cc:78-88): no open returns NULL while any GPU
// works, new files stop going to the sick one, the pool hands out no processor that lives there, and the slot is
// back in service once it answers a probe.
namespace {
int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++g_fail; fprintf(stderr, "router scenario, line %d: ", __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

// one short file through a processor, as the handler's loop does; returns the rms against the direct form (1e9: engine failure)
double one_file(folve::SoundProcessor* p, unsigned seed, const std::vector<std::vector<float>>& h) {
    const int P = 64;
    std::mt19937 rng(seed);
    const size_t n = 3 * P + rng() % (5 * P);
    std::vector<float> a(n * 2);
    for (auto& v : a) v = (float)(rng() % 2001) / 1000.f - 1.f;
    MemSource src(&a, 2);
    MemSink out(2);
    long long left = (long long)n;
    p->Reset();                                      // (what ProcessorPool::Return does between files)
    while (left) {
        const int got = p->FillBuffer(&src);
        if (got <= 0) return 1e9;
        left -= got;
        p->WriteProcessed(&out, got);
    }
    if (!p->ok()) return 1e9;
    return rms(out.d, direct(a, 2, h));
}

int router_scenario(const std::string& dir, const std::vector<std::vector<float>>& h) {
    setenv("FOLVE_AMD_DEVICES", "0,1,2,3,4,5,6,7", 1);
    folve::DeviceRouter* R = folve::DeviceRouter::Default();
    EXPECT(R->device_count() == 8, "8 slots expected, %d", R->device_count());
    R->SetFenceAfter(3);
    R->SetReprobeSeconds(0.3);
    R->SetProbeWaitSeconds(0.25);
    folve::SoundProcessor::SetRunAhead(4);
    folve::ProcessorPool pool(64);
    auto per_slot = [&] { std::vector<int> v; for (int s = 0; s < 8; ++s) v.push_back(R->live_streams(s)); return v; };
    auto open_many = [&](int n, int threads, std::vector<folve::SoundProcessor*>* into) {
        std::vector<std::thread> th;
        std::mutex mu;
        std::atomic<int> next{0}, nulls{0};
        for (int t = 0; t < threads; ++t) th.emplace_back([&] {
            while (next.fetch_add(1) < n) {
                std::string err;
                folve::SoundProcessor* p = pool.GetOrCreate(dir, 44100, 2, 16, &err);
                if (!p) { nulls.fetch_add(1); fprintf(stderr, "open failed: %s\n", err.c_str()); continue; }
                std::lock_guard<std::mutex> lk(mu);
                into->push_back(p);
            }
        });
        for (auto& x : th) x.join();
        return nulls.load();
    };

    // 1. GPU 5 never comes up: 63 opens from 16 threads all succeed, 9 on each of the other seven, slot 5 is fenced
    g_no_create[5] = 1;
    std::vector<folve::SoundProcessor*> procs;
    EXPECT(open_many(63, 16, &procs) == 0, "an open failed while seven GPUs work");
    {
        const std::vector<int> live = per_slot();
        for (int s = 0; s < 8; ++s) EXPECT(live[(size_t)s] == (s == 5 ? 0 : 9), "slot %d holds %d streams", s, live[(size_t)s]);
        EXPECT(R->slot_state(5) == folve::DeviceRouter::kFenced, "slot 5 state %d", (int)R->slot_state(5));
    }
    // 2. it comes up: after the re-probe interval the next open looks at it again and, healthy and empty, it takes the next nine
    g_no_create[5] = 0;
    std::this_thread::sleep_for(std::chrono::milliseconds(350));
    EXPECT(open_many(1, 1, &procs) == 0, "an open failed");      // (this one probes; opens that arrive meanwhile pass the slot over)
    EXPECT(open_many(8, 4, &procs) == 0, "an open failed");
    EXPECT(R->slot_state(5) == folve::DeviceRouter::kHealthy && R->live_streams(5) == 9, "slot 5: state %d, %d streams",
           (int)R->slot_state(5), R->live_streams(5));
    // every processor computes
    for (size_t i = 0; i < procs.size(); ++i) EXPECT(one_file(procs[i], (unsigned)i, h) <= 1e-6, "processor %zu", i);

    // 3. GPU 2 starts failing its calls: the files that live there MOVE to other GPUs (each from its kept input) and come out
    // right; their failures fence the slot; the others never notice
    g_dead[2] = 1;
    int moved = 0;
    for (size_t i = 0; i < procs.size(); ++i) {
        const bool on2 = procs[i]->device() == 2;
        const double e = one_file(procs[i], 1000u + (unsigned)i, h);
        EXPECT(e <= 1e-6 && procs[i]->ok(), "a file %s failed (device %d now, rms %g)", on2 ? "of the dead GPU" : "on a healthy GPU", procs[i]->device(), e);
        if (on2) { EXPECT(procs[i]->device() != 2 && procs[i]->moves() == 1, "a file of the dead GPU did not move (device %d, %d moves)", procs[i]->device(), procs[i]->moves()); ++moved; }
        else EXPECT(procs[i]->moves() == 0, "a file on a healthy GPU moved");
    }
    EXPECT(moved == 9, "%d files were on GPU 2", moved);
    EXPECT(R->slot_state(2) == folve::DeviceRouter::kFenced, "slot 2 state %d after %lld failures", (int)R->slot_state(2), R->slot_failures(2));
    EXPECT(R->live_streams(2) == 0, "slot 2 still holds %d streams", R->live_streams(2));
    // new files while it is fenced (and inside the re-probe interval most of the time; a probe says no anyway): never there, never NULL
    std::vector<folve::SoundProcessor*> more;
    EXPECT(open_many(21, 8, &more) == 0, "an open failed while seven GPUs work");
    for (folve::SoundProcessor* p : more) EXPECT(p->device() != 2, "a new file went to the fenced GPU");
    for (size_t i = 0; i < more.size(); ++i) EXPECT(one_file(more[i], 2000u + (unsigned)i, h) <= 1e-6, "new processor %zu", i);
    {
        const std::vector<int> live = per_slot();
        int total = 0;
        for (int s = 0; s < 8; ++s) {
            total += live[(size_t)s];
            EXPECT(s == 2 ? live[(size_t)s] == 0 : (live[(size_t)s] >= 13 && live[(size_t)s] <= 14), "slot %d holds %d streams", s, live[(size_t)s]);
        }
        EXPECT(total == 93, "%d streams live", total);
    }
    // 4. everything goes back to the pool (which keeps 64 per configuration); the pool hands out none of GPU 2
    for (folve::SoundProcessor* p : procs) pool.Return(p);
    for (folve::SoundProcessor* p : more) pool.Return(p);
    procs.clear(); more.clear();
    EXPECT(R->live_streams(2) == 0, "slot 2 still holds %d streams", R->live_streams(2));
    EXPECT(pool.pooled_count(dir + "/filter-44100.conf") == 64, "pooled %zu", pool.pooled_count(dir + "/filter-44100.conf"));
    EXPECT(open_many(64, 8, &procs) == 0, "an open failed");
    for (folve::SoundProcessor* p : procs) EXPECT(p->device() != 2, "the pool handed out a processor of the fenced GPU");
    // 5. a GPU that fails only after processors were pooled there: the pool's checkout discards them (its engine is fenced by then)
    for (folve::SoundProcessor* p : procs) pool.Return(p);
    procs.clear();
    g_dead[6] = 1;
    for (int i = 0; i < 3; ++i) R->ReportFailure(R->EngineIfCreated(6));
    EXPECT(R->slot_state(6) == folve::DeviceRouter::kFenced, "slot 6 state %d", (int)R->slot_state(6));
    EXPECT(open_many(64, 8, &procs) == 0, "an open failed");
    for (folve::SoundProcessor* p : procs) EXPECT(p->device() != 6 && p->device() != 2, "a processor of a fenced GPU (%d) was handed out", p->device());
    for (size_t i = 0; i < procs.size(); ++i) EXPECT(one_file(procs[i], 3000u + (unsigned)i, h) <= 1e-6, "processor %zu", i);
    // 6. both recover: the next opens probe them and, empty, they fill up first
    g_dead[2] = 0; g_dead[6] = 0;
    std::this_thread::sleep_for(std::chrono::milliseconds(350));
    EXPECT(open_many(1, 1, &more) == 0, "an open failed");        // probes both (one after the other: each is due)
    EXPECT(open_many(7, 2, &more) == 0, "an open failed");
    EXPECT(R->slot_state(2) == folve::DeviceRouter::kHealthy && R->slot_state(6) == folve::DeviceRouter::kHealthy, "states %d %d",
           (int)R->slot_state(2), (int)R->slot_state(6));
    int back = 0;
    for (folve::SoundProcessor* p : more) back += p->device() == 2 || p->device() == 6;
    EXPECT(back == 8, "%d of 8 new files on the recovered GPUs", back);
    for (size_t i = 0; i < more.size(); ++i) EXPECT(one_file(more[i], 4000u + (unsigned)i, h) <= 1e-6, "recovered processor %zu", i);
    // 7. a GPU whose probe hangs: the open does not wait for it beyond the probe wait, and goes elsewhere
    g_dead[3] = 1;
    for (int i = 0; i < 3; ++i) R->ReportFailure(R->EngineIfCreated(3));
    g_probe_hangs_ms[3] = 1500;
    std::this_thread::sleep_for(std::chrono::milliseconds(350));
    {
        const auto t0 = std::chrono::steady_clock::now();
        std::string err;
        folve::SoundProcessor* p = pool.GetOrCreate(dir, 44100, 2, 16, &err);    // pool is empty for this config: a Create
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(p != NULL && p->device() != 3, "open during a hanging probe");
        EXPECT(dt < 1.3, "the open waited %.2f s for a hanging probe", dt);      // (the probe hangs for 1.5 s; the open waits 0.25 s for it)
        if (p) more.push_back(p);
    }
    g_probe_hangs_ms[3] = 0;
    std::this_thread::sleep_for(std::chrono::milliseconds(1500));                // let the late probe come home (it says no)
    EXPECT(R->slot_state(3) == folve::DeviceRouter::kFenced, "slot 3 state %d", (int)R->slot_state(3));
    // 8. every GPU dead: opens fail, at once, with the reference's message for an unusable configuration
    for (folve::SoundProcessor* p : procs) delete p;
    for (folve::SoundProcessor* p : more) delete p;
    procs.clear(); more.clear();
    for (int d = 0; d < 8; ++d) g_dead[d] = 1;
    {
        std::string err;
        folve::SoundProcessor* p = pool.GetOrCreate(dir, 44100, 2, 16, &err);
        EXPECT(p == NULL && err == "Problem parsing " + dir + "/filter-44100.conf", "all GPUs dead: %p '%s'", (void*)p, err.c_str());
        for (int s = 0; s < 8; ++s) EXPECT(R->live_streams(s) == 0, "slot %d keeps %d reservations", s, R->live_streams(s));
    }
    // ... and one comes back — slowly: its probe takes 150 ms, and four files are opened at that moment.  One of them
    // probes, the others find nothing else left and wait for that answer instead of failing.
    g_dead[4] = 0;
    g_probe_hangs_ms[4] = 150;
    R->SetProbeWaitSeconds(3.0);                     // (generous: a loaded test machine may take its time to run the probe thread)
    std::this_thread::sleep_for(std::chrono::milliseconds(350));
    {
        std::vector<folve::SoundProcessor*> four;
        EXPECT(open_many(4, 4, &four) == 0, "an open failed while the only GPU left was being probed");
        for (folve::SoundProcessor* p : four) {
            EXPECT(p->device() == 4, "one GPU back: device %d", p->device());
            EXPECT(one_file(p, 5u, h) <= 1e-6, "the file on the GPU that came back");
            delete p;
        }
    }
    g_probe_hangs_ms[4] = 0;
    printf("{\"router_scenario\": \"%s\", \"failed_checks\": %d}\n", g_fail ? "bad" : "ok", g_fail);
    return g_fail ? 1 : 0;
}

// ---------------------------------------------------------------- the survival scenario (`host_fake <dir> survive`)
// An open file survives its GPU (sound_processor.h): a filter of three partitions (20 000 taps: P = 8192, the state of a
// stream is its last three input blocks), eight file threads on eight slots at run-ahead depths 1 .. 8, and GPUs dying
// under them in mid-file — once, twice, and at last all of them.  Every file's output must equal the convolution of the
// whole file and its peak the maximum of what was written, whatever moved; only when nothing is left does a file end in
// silence, and says so.
int survive_scenario(const std::string& dir) {
    setenv("FOLVE_AMD_DEVICES", "0,1,2,3,4,5,6,7", 1);
    const std::string sub = dir + "/long";
    if (system(("mkdir -p " + sub).c_str()) != 0) return 1;
    {
        FILE* f = fopen((sub + "/filter-44100.conf").c_str(), "w");
        fprintf(f, "/convolver/new 2 2 64 20000\n/impulse/dirac 1 1 0.5 0\n/impulse/dirac 1 1 0.25 9000\n/impulse/dirac 1 1 -0.125 19999\n"
                   "/impulse/dirac 2 2 -0.75 3\n/impulse/dirac 2 2 0.125 17000\n");
        fclose(f);
    }
    std::vector<std::vector<float>> h(2, std::vector<float>(20000, 0.f));
    h[0][0] = 0.5f; h[0][9000] = 0.25f; h[0][19999] = -0.125f; h[1][3] = -0.75f; h[1][17000] = 0.125f;
    folve::DeviceRouter* R = folve::DeviceRouter::Default();
    R->SetFenceAfter(3);
    R->SetReprobeSeconds(1000.0);                    // (a fenced GPU stays fenced for the length of this scenario)
    R->SetProbeWaitSeconds(0.25);
    const int P = 8192;
    std::atomic<int> bad{0};
    // kills[r]: what happens in round r while the files run: device numbers to kill when a thread's file reaches block `at`
    struct Round { int depth; std::vector<int> kill; int at; bool all_dead_at_end; };
    const Round rounds[] = {{1, {2}, 5, false}, {4, {5}, 7, false}, {8, {0, 1}, 9, false}, {3, {3, 4, 6}, 4, false}, {4, {}, 6, true}};   // (the last round runs on the one GPU left; then that one goes too)
    for (const Round& rd : rounds) {
        folve::SoundProcessor::SetRunAhead(rd.depth);
        std::vector<folve::SoundProcessor*> procs;
        for (int t = 0; t < 8; ++t) {
            folve::SoundProcessor* p = folve::SoundProcessor::Create(sub + "/filter-44100.conf", 44100, 2);
            if (!p) { if (!rd.all_dead_at_end) { ++g_fail; fprintf(stderr, "survive: Create failed\n"); } continue; }
            procs.push_back(p);
        }
        std::atomic<int> reached{0};
        std::vector<std::thread> th;
        for (size_t t = 0; t < procs.size(); ++t) th.emplace_back([&, t] {
            folve::SoundProcessor* p = procs[t];
            const int dev0 = p->device();
            std::mt19937 rng((unsigned)(t * 31 + rd.depth));
            // even threads: one file; odd threads: two files handed over gaplessly (convolve-file-handler.cc:328-351) with the
            // GPUs dying two blocks into the SECOND file — the state to replay then spans the first file's last long chunk
            // (saved from its buffer just before the second file's first chunk is read into it), its short last chunk and the
            // topped-up block in between
            const bool two = (t & 1) != 0;
            const size_t na = (size_t)(14 + t) * P + 1 + rng() % (P - 1);     // 14 .. 21 blocks and a short last one
            const size_t nb2 = two ? (size_t)9 * P + 1 + rng() % (P - 1) : 0;
            std::vector<float> a((na + nb2) * 2);
            for (auto& v : a) v = (float)(rng() % 2001) / 1000.f - 1.f;
            std::vector<float> fa(a.begin(), a.begin() + (long)na * 2), fb(a.begin() + (long)na * 2, a.end());
            MemSource sa(&fa, 2), sb(&fb, 2);
            MemSource* srcs[2] = {&sa, &sb};
            const int nsrc = two ? 2 : 1;
            const int at = two ? (int)(na / P) + 3 : rd.at;
            const size_t n = na + nb2;
            MemSink out(2);
            long long left = (long long)n;
            int blocks = 0, si = 0;
            while (left) {
                int got = p->FillBuffer(srcs[si]);
                if (got <= 0) { if (++si >= nsrc) break; continue; }
                left -= got;
                if (!p->is_input_buffer_complete() && si + 1 < nsrc && srcs[si]->pos * 2 == srcs[si]->d->size()) {
                    const int more = p->FillBuffer(srcs[++si]);                // the next file tops the block up
                    got += more;
                    left -= more;
                }
                p->WriteProcessed(&out, got);
                if (++blocks == at) {                                    // every file is in mid-conversion: now the GPUs die
                    if (reached.fetch_add(1) + 1 == (int)procs.size())
                        for (int d : rd.kill) g_dead[d] = 1;
                    while (reached.load() < (int)procs.size()) std::this_thread::yield();
                }
            }
            const bool was_on_killed = std::find(rd.kill.begin(), rd.kill.end(), dev0) != rd.kill.end();
            const std::vector<float> ref = direct(a, 2, h);
            const double e = rms(out.d, ref);
            float mx = 0.f;
            for (float v : out.d) mx = std::max(mx, v);
            if (!(e <= 1e-6) || !p->ok() || fabsf(p->max_output_value() - mx) > 1e-6f || (was_on_killed != (p->moves() == 1)) ||
                (was_on_killed && p->device() == dev0)) {
                bad.fetch_add(1);
                fprintf(stderr, "survive (depth %d): file %zu, device %d -> %d, %d moves, rms %g, peak %g vs %g, ok %d\n", rd.depth, t, dev0,
                        p->device(), p->moves(), e, p->max_output_value(), mx, (int)p->ok());
            }
            // no block of the output is silence (the input is dense noise, the filter has a tap at delay 0 / 3)
            for (size_t b = 0; b + P <= n; b += P) {
                bool any = false;
                for (size_t i = b * 2; i < (b + P) * 2 && !any; ++i) any = out.d[i] != 0.f;
                if (!any) { bad.fetch_add(1); fprintf(stderr, "survive: file %zu has a silent block at %zu\n", t, b / P); break; }
            }
        });
        for (auto& x : th) x.join();
        if (rd.all_dead_at_end) {
            // the last GPU goes too: what is left is silence, said so (ok() false), and the pool would not keep such a processor
            for (int d = 0; d < 8; ++d) g_dead[d] = 1;
            folve::SoundProcessor* p = procs.empty() ? NULL : procs[0];
            if (p) {
                std::vector<float> a((size_t)3 * P * 2, 0.5f);
                MemSource src(&a, 2);
                MemSink out(2);
                long long left = 3 * P;
                p->Reset();
                while (left) { const int got = p->FillBuffer(&src); if (got <= 0) break; left -= got; p->WriteProcessed(&out, got); }
                bool any = false;
                for (float v : out.d) any = any || v != 0.f;
                if (any || p->ok()) { bad.fetch_add(1); fprintf(stderr, "survive: every GPU dead, yet ok %d / sound %d\n", (int)p->ok(), (int)any); }
            }
        }
        for (folve::SoundProcessor* p : procs) delete p;
    }
    for (int s = 0; s < 8; ++s) if (R->live_streams(s) != 0) { bad.fetch_add(1); fprintf(stderr, "survive: slot %d keeps %d streams\n", s, R->live_streams(s)); }
    printf("{\"survive_scenario\": \"%s\", \"failed_checks\": %d}\n", (bad.load() || g_fail) ? "bad" : "ok", bad.load() + g_fail);
    return (bad.load() || g_fail) ? 1 : 0;
}
}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const int nthreads = argc > 2 ? atoi(argv[2]) : 8, rounds = argc > 3 ? atoi(argv[3]) : 6;
    // a 2-channel configuration of 20 taps (block size 64): two diracs per channel and a cross-free diagonal
    const std::string conf = dir + "/filter-44100.conf";
    {
        FILE* f = fopen(conf.c_str(), "w");
        fprintf(f, "/convolver/new 2 2 64 20\n/impulse/dirac 1 1 0.5 0\n/impulse/dirac 1 1 0.25 7\n/impulse/dirac 2 2 -0.75 3\n/impulse/dirac 2 2 0.125 19\n");
        fclose(f);
    }
    std::vector<std::vector<float>> h(2, std::vector<float>(20, 0.f));
    h[0][0] = 0.5f; h[0][7] = 0.25f; h[1][3] = -0.75f; h[1][19] = 0.125f;
    if (argc > 2 && std::string(argv[2]) == "router") return router_scenario(dir, h);
    if (argc > 2 && std::string(argv[2]) == "survive") return survive_scenario(dir);
    std::atomic<long long> bad{0}, files{0};
    folve::ProcessorPool pool(3);
    const int P = 64;
    for (int r = 0; r < rounds; ++r) {
        const int depths[] = {1, 2, 3, 8, 32, 64};
        folve::SoundProcessor::SetRunAhead(depths[r % 6]);
        folve::SoundProcessor::SetDevicePeaks((r & 1) == 0);
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) {
            th.emplace_back([&, t, r] {
                std::mt19937 rng((unsigned)(r * 100 + t));
                std::string err;
                folve::SoundProcessor* p = pool.GetOrCreate(dir, 44100, 2, 16, &err);
                if (!p) { bad.fetch_add(1); fprintf(stderr, "GetOrCreate: %s\n", err.c_str()); return; }
                // file A (ends in a short block most of the time) and, gapless, file B on the same processor
                const size_t na = 1 + rng() % (40 * P), nb = (t & 1) ? P + rng() % (20 * P) : 0;   // (B at least a block: it completes A's last one)
                std::vector<float> a(na * 2), b(nb * 2);
                for (auto& v : a) v = (float)(rng() % 2001) / 1000.f - 1.f;
                for (auto& v : b) v = (float)(rng() % 2001) / 1000.f - 1.f;
                MemSource sa(&a, 2), sb(&b, 2);
                MemSink oa(2), ob(2);
                long long left = (long long)na;
                while (left) {                                  // AddMoreSoundData (convolve-file-handler.cc:370-424)
                    if (p->pending_writes() > 0) { p->WriteProcessed(&oa, p->pending_writes()); continue; }
                    const int want = (int)std::min<long long>(P, left);            // what the reference's FillBuffer returns here
                    const int got = p->FillBuffer(&sa);
                    if (got != want) { bad.fetch_add(1); fprintf(stderr, "FillBuffer returned %d, the reference returns %d\n", got, want); return; }
                    left -= got;
                    if (p->is_input_buffer_complete() != (got == P)) { bad.fetch_add(1); fprintf(stderr, "is_input_buffer_complete\n"); return; }
                    if (!left && !p->is_input_buffer_complete() && nb) {           // gapless: B tops the block up before A's last write
                        const int top = p->FillBuffer(&sb);                        // (PassoverProcessor, cc:328-351)
                        if (top != (int)std::min<size_t>((size_t)(P - got), nb)) { bad.fetch_add(1); fprintf(stderr, "top-up %d\n", top); return; }
                    }
                    const int cut = (int)(rng() % (unsigned)(got + 1));           // drain in two uneven pieces
                    if (cut) p->WriteProcessed(&oa, cut);
                    if (p->pending_writes() != P - cut && cut) { bad.fetch_add(1); fprintf(stderr, "pending_writes %d after %d\n", p->pending_writes(), cut); return; }
                    if (got - cut) p->WriteProcessed(&oa, got - cut);
                }
                if (nb) {
                    long long leftb = (long long)nb - (long long)sb.pos;
                    while (leftb) {                             // B's AddMoreSoundData: first what the donor left processed (cc:373-376)
                        if (p->pending_writes() > 0) { p->WriteProcessed(&ob, p->pending_writes()); continue; }
                        const int got = p->FillBuffer(&sb);
                        if (got != (int)std::min<long long>(P, leftb)) { bad.fetch_add(1); fprintf(stderr, "B: FillBuffer %d\n", got); return; }
                        leftb -= got;
                        p->WriteProcessed(&ob, got);
                    }
                    if (p->pending_writes() > 0 && p->is_input_buffer_complete()) p->WriteProcessed(&ob, p->pending_writes());
                }
                std::vector<float> both = a;
                both.insert(both.end(), b.begin(), b.end());
                std::vector<float> got = oa.d;
                got.insert(got.end(), ob.d.begin(), ob.d.end());
                const std::vector<float> ref = direct(both, 2, h);
                const double e = rms(got, ref);
                float mx = 0.f;
                for (float v : got) mx = std::max(mx, v);
                if (!(e <= 1e-6) || fabsf(p->max_output_value() - mx) > 1e-6f || !p->ok()) {
                    bad.fetch_add(1);
                    fprintf(stderr, "round %d thread %d: frames %zu + %zu, rms %g, peak %g vs %g, reads %lld\n", r, t, na, nb, e,
                            p->max_output_value(), mx, sa.calls);
                }
                files.fetch_add(1);
                pool.Return(p);
            });
        }
        for (auto& x : th) x.join();
    }
    printf("{\"files\": %lld, \"bad\": %lld}\n", files.load(), bad.load());
    return bad.load() == 0 && files.load() == (long long)nthreads * rounds ? 0 : 1;
}
