#include "batch_scheduler.h"

#include <stdlib.h>

#include <atomic>
#include <chrono>

namespace folve {

namespace {
std::atomic<int> g_enabled{-1};        // -1: not decided yet (environment)
std::atomic<int> g_window_us{150};
std::atomic<int> g_max_batch{256};
std::mutex g_map_mu;
std::map<fe_engine*, BatchScheduler*> g_schedulers;
}  // namespace

void BatchScheduler::SetEnabled(bool on) { g_enabled.store(on ? 1 : 0); }

bool BatchScheduler::Enabled() {
    int v = g_enabled.load();
    if (v < 0) {
        const char* env = getenv("FOLVE_AMD_BATCH");
        v = (env && atoi(env) != 0) ? 1 : 0;
        g_enabled.store(v);
    }
    return v == 1;
}

void BatchScheduler::Configure(int window_us, int max_batch) {
    if (window_us >= 0) g_window_us.store(window_us);
    if (max_batch >= 1) g_max_batch.store(max_batch);
}

BatchScheduler* BatchScheduler::ForEngine(fe_engine* engine) {
    std::lock_guard<std::mutex> lk(g_map_mu);
    BatchScheduler*& s = g_schedulers[engine];
    if (!s) s = new BatchScheduler(engine);
    return s;
}

BatchScheduler::BatchScheduler(fe_engine*) : worker_([this] { Loop(); }) { worker_.detach(); }

int BatchScheduler::Process(fe_stream* s, const float* in, int valid_frames, float* out, float* peak_signed,
                            float* peak_abs) {
    Request r{s, in, valid_frames, out, 0.f, 0.f, 0, false};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&r);
    stats_.requests++;
    arrived_.notify_one();
    finished_.wait(lk, [&r] { return r.done; });
    if (peak_signed) *peak_signed = r.peak_signed;
    if (peak_abs) *peak_abs = r.peak_abs;
    return r.rc;
}

void BatchScheduler::Loop() {
    std::vector<Request*> batch;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu_);
            arrived_.wait(lk, [this] { return !queue_.empty(); });
            // collection window: other files' threads are usually a few microseconds behind
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(g_window_us.load());
            const size_t cap = static_cast<size_t>(g_max_batch.load());
            while (queue_.size() < cap) {
                if (arrived_.wait_until(lk, deadline) == std::cv_status::timeout) break;
            }
            const size_t n = queue_.size() < cap ? queue_.size() : cap;
            batch.assign(queue_.begin(), queue_.begin() + static_cast<long>(n));
            queue_.erase(queue_.begin(), queue_.begin() + static_cast<long>(n));
        }
        const int n = static_cast<int>(batch.size());
        std::vector<fe_stream*> ss(batch.size());
        std::vector<const float*> ins(batch.size());
        std::vector<float*> outs(batch.size());
        std::vector<long long> nfr(batch.size());
        for (int i = 0; i < n; ++i) {
            ss[static_cast<size_t>(i)] = batch[static_cast<size_t>(i)]->s;
            ins[static_cast<size_t>(i)] = batch[static_cast<size_t>(i)]->in;
            outs[static_cast<size_t>(i)] = batch[static_cast<size_t>(i)]->out;
            nfr[static_cast<size_t>(i)] = batch[static_cast<size_t>(i)]->frames;
        }
        const int rc = fe_batch_process(ss.data(), n, ins.data(), nfr.data(), outs.data(), FE_HOST_PTRS);
        std::vector<float> ps(batch.size(), 0.f), pa(batch.size(), 0.f);
        if (rc == 0) fe_batch_get_peaks(ss.data(), n, ps.data(), pa.data());
        {
            std::lock_guard<std::mutex> lk(mu_);
            stats_.batches++;
            if (n > stats_.largest) stats_.largest = n;
            for (int i = 0; i < n; ++i) {
                Request* r = batch[static_cast<size_t>(i)];
                r->rc = rc;
                r->peak_signed = ps[static_cast<size_t>(i)];
                r->peak_abs = pa[static_cast<size_t>(i)];
                r->done = true;
            }
        }
        finished_.notify_all();
    }
}

BatchScheduler::Stats BatchScheduler::stats() {
    std::lock_guard<std::mutex> lk(mu_);
    return stats_;
}

}  // namespace folve
