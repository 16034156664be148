#include "batch_scheduler.h"

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <map>

namespace folve {

struct BatchScheduler::Request {
    fe_stream* s = nullptr;
    const float* in = nullptr;
    long long frames = 0;
    float* out = nullptr;
    float* peaks = nullptr;                 // per-block maxima wanted here
    bool peaks_filled = false;
    long long blocks = 0;
    int rc = 0;
    std::string error;
    // written under mu_; kDone is stored LAST (release) and may be read by the request's own thread without the lock:
    // from then on the scheduler does not touch the request, and that thread may free it
    std::atomic<int> state{kQueued};
    std::shared_ptr<Batch> batch;           // once taken out of the queue
    // Where its thread sleeps (in Wait): the gate of the queue generation it arrived in.  Requests that queue up together
    // leave together as one batch, so ONE notify_all on their gate wakes exactly that batch's threads when it completes
    // (22 separate wake-ups from the completing thread took 70 - 110 us of every round), and ONE notify_one picks the
    // batch's waiter when it is submitted.  A gate has its own mutex: the woken threads do not meet on mu_ — a thread whose
    // request is settled never takes mu_ again for it (40 threads re-acquiring one mutex after every batch was most of the
    // cost of a block with 64 and more one-block threads).
    std::shared_ptr<Gate> gate;
    bool sleeping = false;                  // under mu_ (its own thread clears it only on the path that re-takes mu_)
};

namespace {
std::atomic<int> g_enabled{-1};        // -1: not decided yet (environment)
std::atomic<int> g_max_batch{256};
std::atomic<int> g_early_quarters{2};   // a second batch leaves early when queued blocks * 4 >= blocks in flight * this
                                        // (64 files, run-ahead 64: 4 -> 8.6, 2 -> 8.9, 1 -> 9.1 Gsamples/s; one-block calls: no difference)
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

void BatchScheduler::SetEarlyQuarters(int quarters) {
    if (quarters >= 1) g_early_quarters.store(quarters);
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

// Wake the sleepers of a gate (mu_ held or not; lock order mu_ -> gate).  The epoch makes a wake-up that comes between a
// thread's decision to sleep (under mu_) and its wait (under the gate's mutex) count.
void BatchScheduler::WakeGate(Gate* g, bool all) {
    {
        std::lock_guard<std::mutex> gl(g->m);
        ++g->epoch;
    }
    if (all) g->cv.notify_all();
    else g->cv.notify_one();
}

// Called with mu_ held; sleeps on r's gate with mu_ released.  True: the request is settled and mu_ is NOT held (nothing
// of the scheduler's is touched again for it); false: woken for something else (a batch needs a waiter, the queue may
// move), mu_ held again.
bool BatchScheduler::SleepOnGate(std::unique_lock<std::mutex>& lk, Request* r) {
    const std::shared_ptr<Gate> g = r->gate;
    unsigned long long seen;
    {
        std::lock_guard<std::mutex> gl(g->m);
        seen = g->epoch;
    }
    r->sleeping = true;
    lk.unlock();
    {
        std::unique_lock<std::mutex> gl(g->m);
        g->cv.wait(gl, [&] { return g->epoch != seen; });
    }
    if (r->state.load(std::memory_order_acquire) == kDone) return true;
    lk.lock();
    r->sleeping = false;
    return false;
}

BatchScheduler::Request* BatchScheduler::Submit(fe_stream* s, const float* in, long long frames, float* out, float* block_peaks) {
    std::unique_lock<std::mutex> lk(mu_);
    return SubmitLocked(lk, s, in, frames, out, block_peaks);
}

BatchScheduler::Request* BatchScheduler::SubmitLocked(std::unique_lock<std::mutex>& lk, fe_stream* s, const float* in, long long frames,
                                                      float* out, float* block_peaks) {
    Request* r = new Request();
    r->s = s;
    r->in = in;
    r->frames = frames;
    r->out = out;
    r->peaks = block_peaks;
    const int P = fe_stream_block_size(s);
    r->blocks = P > 0 ? (frames + P - 1) / P : 0;
    stats_.requests++;
    stats_.blocks += r->blocks;
    if (!next_gate_) next_gate_ = std::make_shared<Gate>();
    r->gate = next_gate_;
    queue_.push_back(r);
    queued_blocks_ += r->blocks;
    if (lanes_busy_ > 0 && !pumping_) {
        // The GPU is taken.  If one of the batches has finished unobserved (its threads are all busy serving
        // what they got earlier), retire it now so that the queue — this request included — moves on.
        std::shared_ptr<Batch> finished;
        for (const std::shared_ptr<Batch>& b : flying_)
            if (b->ticket && !b->has_waiter && fe_ticket_done(b->ticket) == 1) { finished = b; break; }
        if (finished) CompleteLocked(lk, finished);
    }
    PumpLocked(lk);
    return r;
}

bool BatchScheduler::Ready(Request* r) {
    if (r->state.load(std::memory_order_acquire) == kDone) return true;
    std::unique_lock<std::mutex> lk(mu_);
    if (r->state == kDone) return true;
    if (r->state == kFlying) {
        const std::shared_ptr<Batch> b = r->batch;
        if (b->ticket && !b->has_waiter && fe_ticket_done(b->ticket) == 1) {
            CompleteLocked(lk, b);
            return r->state == kDone;
        }
    }
    return false;
}

int BatchScheduler::Wait(Request* r, std::string* error, bool* peaks_filled) {
    if (r->state.load(std::memory_order_acquire) != kDone) {
        std::unique_lock<std::mutex> lk(mu_);
        WaitLocked(lk, r);
    }
    const int rc = r->rc;
    if (error) *error = r->error;
    if (peaks_filled) *peaks_filled = r->peaks_filled;
    delete r;
    return rc;
}

// mu_ held on entry, released on return; the request is settled then.
void BatchScheduler::WaitLocked(std::unique_lock<std::mutex>& lk, Request* r) {
    while (r->state != kDone) {
        if (r->state == kFlying) {
            const std::shared_ptr<Batch> b = r->batch;
            if (b->ticket && !b->has_waiter) {                            // nobody waits for this batch yet: this thread does
                CompleteLocked(lk, b);
            } else {                                                      // (being submitted, or somebody else waits)
                // a wake-up meant to find a waiter for ANOTHER batch may have landed here (one gate can span two
                // batches when a queue generation was cut at the batch-size cap): take that batch before sleeping again
                std::shared_ptr<Batch> other;
                for (const std::shared_ptr<Batch>& o : flying_)
                    if (o->ticket && !o->has_waiter) { other = o; break; }
                if (other) { CompleteLocked(lk, other); continue; }
                if (SleepOnGate(lk, r)) return;
            }
            continue;
        }
        // still queued: both lanes are taken (or a pump is under way).  Help the oldest batch nobody waits for.
        std::shared_ptr<Batch> help;
        for (const std::shared_ptr<Batch>& b : flying_)
            if (b->ticket && !b->has_waiter) { help = b; break; }
        if (help) { CompleteLocked(lk, help); continue; }
        if (MayPumpLocked()) { PumpLocked(lk); continue; }
        if (SleepOnGate(lk, r)) return;
    }
    lk.unlock();
}

int BatchScheduler::Process(fe_stream* s, const float* in, int valid_frames, float* out, std::string* error) {
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (queue_.empty() && lanes_busy_ == 0 && !pumping_) {
            // The GPU is idle: this block runs at once, by itself, through the engine's synchronous latency path.
            lanes_busy_++;
            flying_blocks_++;
            stats_.requests++;
            stats_.blocks++;
            stats_.batches++;
            if (stats_.largest < 1) stats_.largest = 1;
            lk.unlock();
            const int rc = fe_stream_process(s, in, valid_frames, out, NULL, NULL);
            if (rc != 0 && error) *error = fe_last_error();
            lk.lock();
            lanes_busy_--;
            flying_blocks_--;
            PumpLocked(lk);
            return rc;
        }
        Request* r = SubmitLocked(lk, s, in, valid_frames, out, nullptr);
        WaitLocked(lk, r);                    // (releases mu_)
        const int rc = r->rc;
        if (error) *error = r->error;
        delete r;
        return rc;
    }
}

// Wait for b's ticket (outside the lock), settle its requests, put the next batch on the GPU, then wake b's threads.
void BatchScheduler::CompleteLocked(std::unique_lock<std::mutex>& lk, const std::shared_ptr<Batch>& b) {
    b->has_waiter = true;
    fe_ticket* t = b->ticket;
    lk.unlock();
    const int rc = fe_ticket_wait(t);
    const std::string msg = rc != 0 ? fe_last_error() : "";
    lk.lock();
    b->ticket = nullptr;
    std::vector<Request*> settled;
    settled.swap(b->reqs);
    b->done = true;
    flying_.erase(std::remove(flying_.begin(), flying_.end(), b), flying_.end());
    lanes_busy_--;
    flying_blocks_ -= b->blocks;
    // Wake the batch's threads FIRST — one notify_all per gate, a single system call as a rule — and submit the next batch
    // while they are waking up: a woken thread needs 20 - 30 us to run again, the engine call below 40 - 60 us; done the
    // other way round every thread of the batch lost that engine call's time before it could queue its next block.
    std::vector<std::shared_ptr<Gate>> gates;                     // (a batch's requests nearly always share one gate)
    for (Request* q : settled) {
        q->rc = rc;
        if (rc != 0) q->error = msg;
        q->peaks_filled = rc == 0 && q->peaks != nullptr;
        if (q->sleeping && std::find(gates.begin(), gates.end(), q->gate) == gates.end()) gates.push_back(q->gate);
    }
    // (a request is not touched by the scheduler after this store: its thread may see it without the lock and free it)
    for (Request* q : settled) q->state.store(kDone, std::memory_order_release);
    for (const std::shared_ptr<Gate>& g : gates) WakeGate(g.get(), true);
    PumpLocked(lk);                         // everything that queued up meanwhile leaves as the next batch
}

// A batch may leave now: the GPU is idle, or it holds one batch and the queue has grown at least as big.
bool BatchScheduler::MayPumpLocked() const {
    if (pumping_ || queue_.empty() || lanes_busy_ >= kLanes) return false;
    return lanes_busy_ == 0 || queued_blocks_ * 4 >= flying_blocks_ * g_early_quarters.load();
}

void BatchScheduler::PumpLocked(std::unique_lock<std::mutex>& lk) {
    while (MayPumpLocked()) {
        pumping_ = true;
        const std::shared_ptr<Batch> b = std::make_shared<Batch>();
        next_gate_.reset();                     // later arrivals are another generation: they sleep on a gate of their own
        const size_t cap = static_cast<size_t>(g_max_batch.load());
        const size_t n = std::min(queue_.size(), cap);
        long long blocks = 0;
        for (size_t i = 0; i < n; ++i) {
            Request* r = queue_.front();
            queue_.pop_front();
            r->state = kFlying;
            r->batch = b;
            blocks += r->blocks;
            b->reqs.push_back(r);
        }
        b->blocks = blocks;
        queued_blocks_ -= blocks;
        flying_blocks_ += blocks;
        if (lanes_busy_ > 0) stats_.overlapped++;
        lanes_busy_++;
        stats_.batches++;
        if (blocks > stats_.largest) stats_.largest = blocks;
        lk.unlock();

        // ---- outside the lock: the engine call (a few tens of microseconds of launches) ----
        std::vector<fe_stream*> ss(n);
        std::vector<const float*> ins(n);
        std::vector<float*> outs(n), pks(n);
        std::vector<long long> nfr(n), before(n);
        bool any_peaks = false;
        for (size_t i = 0; i < n; ++i) {
            const Request* r = b->reqs[i];
            ss[i] = r->s;
            ins[i] = r->in;
            outs[i] = r->out;
            pks[i] = r->peaks;
            any_peaks = any_peaks || r->peaks != nullptr;
            nfr[i] = r->frames;
            before[i] = fe_stream_blocks_done(r->s);
        }
        fe_ticket* ticket = nullptr;
        const int rc = fe_batch_submit_peaks(ss.data(), static_cast<int>(n), ins.data(), nfr.data(), outs.data(),
                                             any_peaks ? pks.data() : nullptr, &ticket);
        if (rc != 0) {
            // Not submitted (a buffer outside its stream's bound memory, or a launch round refused).  A round that
            // fails leaves its streams where they were: those requests run one by one, each with its own status, so
            // that one bad stream does not fail its neighbours.  Streams of an EARLIER round of the same call
            // (another filter's group) have consumed their blocks, whose output may be incomplete: they fail with
            // the batch's error rather than run twice.  (The engine has drained that round before returning.)
            const std::string msg = fe_last_error();
            for (size_t i = 0; i < n; ++i) {
                Request* r = b->reqs[i];
                if (fe_stream_blocks_done(r->s) != before[i]) {
                    r->rc = rc;
                    r->error = "blocks consumed by a batch that failed later: " + msg;
                } else {
                    RunAlone(r);
                }
            }
        }
        lk.lock();
        pumping_ = false;
        if (rc == 0) {
            b->ticket = ticket;
            flying_.push_back(b);
        } else {
            std::vector<std::shared_ptr<Gate>> gates;
            for (Request* r : b->reqs)
                if (r->sleeping && std::find(gates.begin(), gates.end(), r->gate) == gates.end()) gates.push_back(r->gate);
            for (Request* r : b->reqs) r->state.store(kDone, std::memory_order_release);    // settled (one by one, above)
            b->reqs.clear();
            b->done = true;
            lanes_busy_--;
            flying_blocks_ -= b->blocks;
            for (const std::shared_ptr<Gate>& g : gates) WakeGate(g.get(), true);
        }
        if (rc == 0) {
            // ONE sleeping thread becomes the batch's waiter: one of its own, else one whose request is still queued
            // (it helps, Wait); the others sleep on until their request is settled.  If every thread is busy elsewhere
            // the batch is picked up by the first that comes to wait or to submit.
            Request* waker = nullptr;
            for (Request* r : b->reqs)
                if (r->sleeping) { waker = r; break; }
            if (!waker)
                for (Request* r : queue_)
                    if (r->sleeping) { waker = r; break; }
            if (waker) WakeGate(waker->gate.get(), false);    // (whoever wakes on that gate finds a batch without a waiter: Wait)
        }
    }
}

void BatchScheduler::RunAlone(Request* r) {
    fe_stream* ss[1] = {r->s};
    const float* ii[1] = {r->in};
    float* oo[1] = {r->out};
    long long nn[1] = {r->frames};
    r->rc = fe_batch_process(ss, 1, ii, nn, oo, FE_HOST_PTRS);
    if (r->rc != 0) r->error = fe_last_error();
}

BatchScheduler::Stats BatchScheduler::stats() {
    std::lock_guard<std::mutex> lk(mu_);
    return stats_;
}

}  // namespace folve
