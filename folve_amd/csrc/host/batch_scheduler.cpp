#include "batch_scheduler.h"

#include <stdlib.h>

#include <atomic>
#include <map>
#include <memory>

namespace folve {

namespace {
std::atomic<int> g_enabled{-1};        // -1: not decided yet (environment)
std::atomic<int> g_max_batch{256};
std::mutex g_map_mu;
// owned here: destroyed with the process (no threads to stop)
std::map<fe_engine*, std::unique_ptr<BatchScheduler>>& Schedulers() {
    static std::map<fe_engine*, std::unique_ptr<BatchScheduler>> m;
    return m;
}
}  // namespace

void BatchScheduler::SetEnabled(bool on) { g_enabled.store(on ? 1 : 0); }

bool BatchScheduler::Enabled() {
    int v = g_enabled.load();
    if (v < 0) {
        const char* env = getenv("FOLVE_AMD_BATCH");
        v = (env && atoi(env) == 0) ? 0 : 1;
        g_enabled.store(v);
    }
    return v == 1;
}

void BatchScheduler::Configure(int max_batch) {
    if (max_batch >= 1) g_max_batch.store(max_batch);
}

BatchScheduler* BatchScheduler::ForEngine(fe_engine* engine) {
    std::lock_guard<std::mutex> lk(g_map_mu);
    std::unique_ptr<BatchScheduler>& s = Schedulers()[engine];
    if (!s) s.reset(new BatchScheduler());
    return s.get();
}

void BatchScheduler::ReleaseEngine(fe_engine* engine) {
    std::lock_guard<std::mutex> lk(g_map_mu);
    Schedulers().erase(engine);
}

int BatchScheduler::Process(fe_stream* s, const float* in, int valid_frames, float* out, std::string* error) {
    Request r{s, in, valid_frames, out, 0, false, std::string()};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&r);
    stats_.requests++;
    while (!r.done) {
        if (busy_) {                       // a call is in flight: park until its thread has looked at the queue
            finished_.wait(lk);
            continue;
        }
        // The GPU is idle: this thread takes everything that is parked (its own block included).
        busy_ = true;
        const size_t cap = static_cast<size_t>(g_max_batch.load());
        const size_t n = queue_.size() < cap ? queue_.size() : cap;
        std::vector<Request*> batch(queue_.begin(), queue_.begin() + static_cast<long>(n));
        queue_.erase(queue_.begin(), queue_.begin() + static_cast<long>(n));
        lk.unlock();
        Run(batch);
        lk.lock();
        busy_ = false;
        stats_.batches++;
        if (static_cast<long long>(n) > stats_.largest) stats_.largest = static_cast<long long>(n);
        for (Request* q : batch) q->done = true;
        finished_.notify_all();            // the served threads leave; one parked thread becomes the next leader
    }
    if (error) *error = r.error;
    return r.rc;
}

void BatchScheduler::Run(std::vector<Request*>& batch) {
    const int n = static_cast<int>(batch.size());
    if (n == 1) {
        Request* r = batch[0];
        r->rc = fe_stream_process(r->s, r->in, r->frames, r->out, NULL, NULL);
        if (r->rc != 0) r->error = fe_last_error();
        return;
    }
    std::vector<fe_stream*> ss(batch.size());
    std::vector<const float*> ins(batch.size());
    std::vector<float*> outs(batch.size());
    std::vector<long long> nfr(batch.size());
    for (int i = 0; i < n; ++i) {
        const Request* r = batch[static_cast<size_t>(i)];
        ss[static_cast<size_t>(i)] = r->s;
        ins[static_cast<size_t>(i)] = r->in;
        outs[static_cast<size_t>(i)] = r->out;
        nfr[static_cast<size_t>(i)] = r->frames;
    }
    const int rc = fe_batch_process(ss.data(), n, ins.data(), nfr.data(), outs.data(), FE_HOST_PTRS);
    if (rc == 0) {
        for (Request* r : batch) r->rc = 0;
        return;
    }
    // The batch was refused as a whole (a launch round that fails leaves its streams where they
    // were): run the blocks one by one so that one bad stream does not fail its neighbours, and
    // every block gets its own status and message.
    for (Request* r : batch) {
        r->rc = fe_stream_process(r->s, r->in, r->frames, r->out, NULL, NULL);
        if (r->rc != 0) r->error = fe_last_error();
    }
}

BatchScheduler::Stats BatchScheduler::stats() {
    std::lock_guard<std::mutex> lk(mu_);
    return stats_;
}

}  // namespace folve
