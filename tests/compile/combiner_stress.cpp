// CPU stress test of folve::BatchScheduler (the per-GPU combiner) against a FAKE engine: this file defines the handful of
// folve_engine.h entry points the scheduler calls — tickets that complete after a random delay, a submit that is refused now
// and then — and drives the real batch_scheduler.cpp from many threads with the synchronous (Process) and the asynchronous
// (Submit ... Wait) pattern mixed, as SoundProcessors with and without run-ahead do.  Every request must be computed exactly
// once, with its own status, and nothing may deadlock.  tests/test_host_cpu.py builds it with -fsanitize=thread and runs it:
// the success path of the combiner (tickets, waiters, per-request wake-ups) needs a GPU otherwise.
#include "../../folve_amd/csrc/host/batch_scheduler.cpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <random>
#include <thread>

struct fe_stream {
    std::atomic<long long> blocks_done{0};
    int P = 64;
};
struct fe_ticket {
    std::chrono::steady_clock::time_point ready;
    std::vector<const float*> in;
    std::vector<float*> out, peaks;
    std::vector<long long> frames;
};

namespace {
std::atomic<long long> g_submits{0}, g_refused{0}, g_sync{0}, g_inflight{0}, g_max_inflight{0};
thread_local std::string g_err;
void compute(const float* in, float* out, long long frames) {
    for (long long i = 0; i < frames; ++i) out[i] = in[i] + 1.0f;            // "the convolution"
}
}  // namespace

extern "C" {
const char* fe_last_error(void) { return g_err.c_str(); }
int fe_stream_block_size(const fe_stream* s) { return s ? s->P : 0; }
long long fe_stream_blocks_done(const fe_stream* s) { return s ? s->blocks_done.load() : 0; }
int fe_batch_submit_peaks(fe_stream* const* streams, int n, const float* const* in, const long long* nframes, float* const* out,
                          float* const* block_peaks, fe_ticket** ticket) {
    const long long k = g_submits.fetch_add(1);
    if (k % 17 == 5) { g_refused.fetch_add(1); g_err = "refused (fake)"; return -4; }   // nothing enqueued: the scheduler retries one by one
    fe_ticket* t = new fe_ticket();
    t->ready = std::chrono::steady_clock::now() + std::chrono::microseconds(30 + (k * 37) % 200);
    for (int i = 0; i < n; ++i) {
        t->in.push_back(in[i]); t->out.push_back(out[i]); t->frames.push_back(nframes[i]);
        t->peaks.push_back(block_peaks ? block_peaks[i] : nullptr);
        streams[i]->blocks_done.fetch_add((nframes[i] + streams[i]->P - 1) / streams[i]->P);
    }
    const long long f = g_inflight.fetch_add(1) + 1;
    long long m = g_max_inflight.load();
    while (f > m && !g_max_inflight.compare_exchange_weak(m, f)) {}
    *ticket = t;
    return 0;
}
int fe_ticket_done(fe_ticket* t) { return std::chrono::steady_clock::now() >= t->ready ? 1 : 0; }
int fe_ticket_wait(fe_ticket* t) {
    std::this_thread::sleep_until(t->ready);
    for (size_t i = 0; i < t->in.size(); ++i) {
        compute(t->in[i], t->out[i], t->frames[i]);
        if (t->peaks[i]) t->peaks[i][0] = 42.0f;
    }
    g_inflight.fetch_sub(1);
    delete t;
    return 0;
}
int fe_stream_process(fe_stream* s, const float* in, int valid_frames, float* out, float*, float*) {
    g_sync.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(40));
    compute(in, out, valid_frames);
    s->blocks_done.fetch_add(1);
    return 0;
}
int fe_batch_process(fe_stream* const* streams, int n, const float* const* in, const long long* nframes, float* const* out, int) {
    g_sync.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(60));
    for (int i = 0; i < n; ++i) {
        compute(in[i], out[i], nframes[i]);
        streams[i]->blocks_done.fetch_add((nframes[i] + streams[i]->P - 1) / streams[i]->P);
    }
    return 0;
}
}

int main(int argc, char** argv) {
    const int nthreads = argc > 1 ? atoi(argv[1]) : 24, iters = argc > 2 ? atoi(argv[2]) : 400;
    folve::BatchScheduler* sch = folve::BatchScheduler::ForEngine(nullptr);
    std::atomic<long long> bad{0}, done{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&, t] {
            std::mt19937 rng(1000 + t);
            fe_stream s;
            std::vector<float> a(64 * 8), b(64 * 8), oa(64 * 8), ob(64 * 8);
            float pk[16];
            for (int it = 0; it < iters; ++it) {
                const bool async = (t % 3) != 0;                       // two thirds of the "files" run ahead
                const long long frames = async ? 64 * (1 + (long long)(rng() % 8)) : 1 + (long long)(rng() % 64);
                for (long long i = 0; i < frames; ++i) a[(size_t)i] = (float)(it * 1000 + t + i);
                std::string err;
                if (!async) {
                    if (sch->Process(&s, a.data(), (int)frames, oa.data(), &err) != 0) bad.fetch_add(1);
                } else {
                    folve::BatchScheduler::Request* r = sch->Submit(&s, a.data(), frames, oa.data(), (it & 1) ? pk : nullptr);
                    if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 120));   // serving the other chunk
                    if (rng() % 5 == 0) (void)sch->Ready(r);
                    bool filled = false;
                    if (sch->Wait(r, &err, &filled) != 0) bad.fetch_add(1);
                }
                for (long long i = 0; i < frames; ++i)
                    if (oa[(size_t)i] != a[(size_t)i] + 1.0f) { bad.fetch_add(1); break; }
                done.fetch_add(1);
            }
        });
    }
    for (auto& x : th) x.join();
    const folve::BatchScheduler::Stats st = sch->stats();
    printf("{\"requests\": %lld, \"batches\": %lld, \"largest\": %lld, \"overlapped\": %lld, \"submits\": %lld, \"refused\": %lld, "
           "\"sync_calls\": %lld, \"max_tickets_in_flight\": %lld, \"done\": %lld, \"bad\": %lld}\n",
           st.requests, st.batches, st.largest, st.overlapped, g_submits.load(), g_refused.load(), g_sync.load(), g_max_inflight.load(),
           done.load(), bad.load());
    return (bad.load() == 0 && done.load() == (long long)nthreads * iters && st.requests == done.load() && g_max_inflight.load() <= 2) ? 0 : 1;
}
