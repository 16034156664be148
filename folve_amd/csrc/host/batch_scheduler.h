// batch_scheduler.h — coalesces per-file block requests into GPU batches, without a timer.
//
// In folve every open file is pulled by its own thread: a FUSE worker in
// ConversionBuffer::FillUntil (conversion-buffer.cc:151-163) or the BufferThread running ahead
// of the reader (buffer-thread.cc:73-105), one 8192-frame block per SoundProcessor::Process call.
// The GPU of a device serves one launch chain at a time, so concurrent calls queue anyway; this
// class turns that queue into batches ("combining"): a thread that finds its GPU idle runs its own
// block at once — no hand-off, no collection window, nothing added to a lone stream's latency —
// and while a batch is on the GPU every other thread's block is parked.  The thread that sees a batch
// complete first SUBMITS all parked blocks as the next batch (fe_batch_submit: the kernels are enqueued,
// nobody waits yet), then wakes the threads of the finished batch, then one thread of the new batch,
// which waits for its ticket and does the same in turn.  The GPU therefore never waits for a sleeping
// thread to wake up (50 - 100 us, which used to be a third of every round with 64 file threads); the
// wake-ups happen while the next batch runs.  Batches form exactly when there is contention and grow
// with it.  No dispatcher thread exists, so there is nothing to join at exit.
// Results are bit-identical to unbatched calls (same kernels, same per-stream arithmetic).
#pragma once

#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/folve_engine.h"

namespace folve {

class BatchScheduler {
public:
    struct Stats {
        long long requests = 0;     // blocks submitted
        long long batches = 0;      // engine calls issued
        long long largest = 0;      // most blocks in one call
    };

    // The scheduler serving `engine` (created on first use; freed by ReleaseEngine or at exit).
    static BatchScheduler* ForEngine(fe_engine* engine);
    static void ReleaseEngine(fe_engine* engine);
    // Process-wide switch: on by default; off = every Process call goes to the engine by itself
    // (FOLVE_AMD_BATCH=0 in the environment does the same).
    static void SetEnabled(bool on);
    static bool Enabled();
    // Upper bound of blocks per engine call.
    static void Configure(int max_batch);

    // One block for one stream, exactly fe_stream_process without the peaks (the caller scans its
    // own output, as sound-processor.cc:116-125 does); blocks until the block has been computed.
    // Returns the engine's status for THIS block; *error receives the engine's message on failure.
    int Process(fe_stream* s, const float* in, int valid_frames, float* out, std::string* error);

    Stats stats();

private:
    // What a parked thread is told when it is woken.
    enum State {
        kParked = 0,
        kDone,      // your block has been computed: take rc and leave
        kWait,      // your batch is on the GPU: wait for its ticket, then finish the batch
        kLead       // your batch could not be submitted ahead: run it yourself, then finish it
    };
    struct Batch;
    struct Request {
        fe_stream* s;
        const float* in;
        int frames;
        float* out;
        int rc;
        std::string error;
        bool settled;           // rc is final already (its block was consumed by a submission that failed later)
        State state;            // under mu_
        Batch* batch;           // with kWait / kLead
        int slot;               // which of cv_ this thread sleeps on
    };
    struct Batch {
        std::vector<Request*> reqs;
        fe_ticket* ticket = nullptr;
    };
    static const int kSlots = 64;

    BatchScheduler() {}
    void RunNow(Batch* b);                       // synchronous engine call(s) for b
    bool Submit(Batch* b);                       // fe_batch_submit; false: not possible, nothing enqueued
    void AwaitTicket(Batch* b);                  // fe_ticket_wait, status into the requests
    void Finish(Batch* b, Request* self);        // next batch to the GPU, wake b's threads, appoint the next batch's thread

    std::mutex mu_;
    // Parked threads sleep on one of a few condition variables owned by the scheduler (not by the
    // request, whose thread may be gone the moment it is released): a wake-up reaches the threads of
    // one slot, not all parked threads.
    std::condition_variable cv_[kSlots];
    int next_slot_ = 0;
    std::vector<Request*> queue_;
    bool busy_ = false;         // a batch is on the GPU or being run: new blocks park
    Stats stats_;
};

}  // namespace folve
