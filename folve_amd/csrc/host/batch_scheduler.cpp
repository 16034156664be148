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
    Request r{s, in, valid_frames, out, 0, std::string(), false, kParked, nullptr, 0};
    std::unique_lock<std::mutex> lk(mu_);
    stats_.requests++;
    if (!busy_) {
        // The GPU is idle: this block runs at once, by itself (nothing can be parked while the GPU is idle).
        busy_ = true;
        lk.unlock();
        Batch* b = new Batch();
        b->reqs.push_back(&r);
        RunNow(b);
        Finish(b, &r);
    } else {
        r.slot = next_slot_;
        next_slot_ = (next_slot_ + 1) % kSlots;
        queue_.push_back(&r);
        while (r.state == kParked) cv_[r.slot].wait(lk);
        const State st = r.state;
        Batch* b = r.batch;
        lk.unlock();
        if (st == kWait) {
            AwaitTicket(b);
            Finish(b, &r);
        } else if (st == kLead) {
            RunNow(b);
            Finish(b, &r);
        }
    }
    if (error) *error = r.error;
    return r.rc;
}

void BatchScheduler::Finish(Batch* b, Request* self) {
    // 1. everything that parked meanwhile becomes the next batch and goes to the GPU before anybody is woken
    Batch* next = nullptr;
    bool slots[kSlots] = {};
    {
        std::lock_guard<std::mutex> lk(mu_);
        stats_.batches++;
        if (static_cast<long long>(b->reqs.size()) > stats_.largest) stats_.largest = static_cast<long long>(b->reqs.size());
        if (queue_.empty()) {
            busy_ = false;
        } else {
            const size_t cap = static_cast<size_t>(g_max_batch.load());
            const size_t n = queue_.size() < cap ? queue_.size() : cap;
            next = new Batch();
            next->reqs.assign(queue_.begin(), queue_.begin() + static_cast<long>(n));
            queue_.erase(queue_.begin(), queue_.begin() + static_cast<long>(n));
        }
        for (Request* q : b->reqs)
            if (q != self) { q->state = kDone; slots[q->slot] = true; }       // (rc and error were written before)
    }
    const bool submitted = next && Submit(next);
    // 2. the finished batch's threads leave — unless the next batch still waits for somebody to run it
    // (a request is its thread's stack: once its state is published it may be gone — nothing of it is read afterwards)
    Request* heir = next ? next->reqs[0] : nullptr;
    auto appoint = [&](State st) {
        int slot;
        { std::lock_guard<std::mutex> lk(mu_); slot = heir->slot; heir->batch = next; heir->state = st; }
        cv_[slot].notify_all();
    };
    if (heir && !submitted) appoint(kLead);
    for (int i = 0; i < kSlots; ++i)
        if (slots[i]) cv_[i].notify_all();
    // 3. one thread of the submitted batch waits for it
    if (heir && submitted) appoint(kWait);
    delete b;
}

bool BatchScheduler::Submit(Batch* b) {
    const size_t n = b->reqs.size();
    std::vector<fe_stream*> ss(n);
    std::vector<const float*> ins(n);
    std::vector<float*> outs(n);
    std::vector<long long> nfr(n);
    for (size_t i = 0; i < n; ++i) {
        const Request* r = b->reqs[i];
        if (!r->s) return false;
        ss[i] = r->s;
        ins[i] = r->in;
        outs[i] = r->out;
        nfr[i] = r->frames;
    }
    std::vector<long long> before(n);
    for (size_t i = 0; i < n; ++i) before[i] = fe_stream_blocks_done(ss[i]);
    const int rc = fe_batch_submit(ss.data(), static_cast<int>(n), ins.data(), nfr.data(), outs.data(), &b->ticket);
    if (rc == 0) return true;
    // Not submitted (buffers not bound, or a launch round refused).  Streams of an earlier round of the same
    // call have consumed their block: they are settled with the error; the others are run by RunNow.
    const std::string msg = fe_last_error();
    for (size_t i = 0; i < n; ++i) {
        Request* r = b->reqs[i];
        if (fe_stream_blocks_done(r->s) != before[i]) {
            r->rc = rc;
            r->error = "block consumed by a batch that failed later: " + msg;
            r->settled = true;
        }
    }
    return false;
}

void BatchScheduler::AwaitTicket(Batch* b) {
    const int rc = fe_ticket_wait(b->ticket);
    b->ticket = nullptr;
    const std::string msg = rc != 0 ? fe_last_error() : "";
    for (Request* r : b->reqs) {
        r->rc = rc;
        if (rc != 0) r->error = msg;
    }
}

void BatchScheduler::RunNow(Batch* b) {
    std::vector<Request*> batch;
    for (Request* r : b->reqs)
        if (!r->settled) batch.push_back(r);
    const int n = static_cast<int>(batch.size());
    if (n == 0) return;
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
    std::vector<long long> before(batch.size());
    for (int i = 0; i < n; ++i) before[static_cast<size_t>(i)] = fe_stream_blocks_done(ss[static_cast<size_t>(i)]);
    const int rc = fe_batch_process(ss.data(), n, ins.data(), nfr.data(), outs.data(), FE_HOST_PTRS);
    if (rc == 0) {
        for (Request* r : batch) r->rc = 0;
        return;
    }
    const std::string msg = fe_last_error();
    // The batch was refused.  A launch round that fails leaves its streams where they were: those blocks
    // are run one by one, so that one bad stream does not fail its neighbours and every block gets its own
    // status and message.  Streams of an EARLIER round of the same call (another filter's group) have
    // consumed their block, but its output may never have been fetched: they fail with the batch's error
    // rather than run the block twice.
    for (int i = 0; i < n; ++i) {
        Request* r = batch[static_cast<size_t>(i)];
        if (fe_stream_blocks_done(r->s) != before[static_cast<size_t>(i)]) {
            r->rc = rc;
            r->error = "block consumed by a batch that failed later: " + msg;
            continue;
        }
        r->rc = fe_stream_process(r->s, r->in, r->frames, r->out, NULL, NULL);
        if (r->rc != 0) r->error = fe_last_error();
    }
}

BatchScheduler::Stats BatchScheduler::stats() {
    std::lock_guard<std::mutex> lk(mu_);
    return stats_;
}

}  // namespace folve
