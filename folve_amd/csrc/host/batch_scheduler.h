// batch_scheduler.h — coalesces the block requests of many open files into GPU batches, without a timer.
//
// In folve every open file is pulled by its own thread: a FUSE worker in
// ConversionBuffer::FillUntil (conversion-buffer.cc:151-163) or the BufferThread running ahead
// of the reader (buffer-thread.cc:73-105).  A SoundProcessor hands the combiner REQUESTS: a span of
// consecutive blocks of its stream (one block for the reference's synchronous Process(), sound-processor.cc:98-127;
// a run-ahead chunk of up to N blocks when the processor reads ahead of its reader, sound_processor.h).
//
// A request is submitted without waiting (Submit) and collected later (Wait); Process() is the two
// back to back.  A request that finds the GPU idle leaves at once as a batch of its own — nothing is ever
// added to a lone stream's latency, there is no collection window and no dispatcher thread — and requests
// that arrive while a batch is on the GPU queue up; whichever thread next sees a batch complete submits
// everything queued as the next batch BEFORE it wakes anybody, so the GPU never waits for a sleeping thread.
// Batches form exactly when there is contention and grow with it.  A SECOND batch goes to the GPU while one is
// still running only when the queue holds at least half as many blocks as that batch (two populations of files
// taking turns: the GPU always has the next batch queued behind the current one); a trickle of small batches
// beside a big one would only multiply the fixed cost of a launch chain.  Inside the engine a big batch is a
// duplex pipeline over two launch lanes (folve_engine.h, fe_batch_submit): while one chunk's K3 writes
// results to host memory, the next chunk's K1 already reads its PCM — both directions of the bus at work.
//
// The threads that wait are the only workers: a thread whose request is in a batch nobody waits for yet
// becomes that batch's waiter (fe_ticket_wait), settles every request in it, submits the next batch and
// wakes the others — with ONE notify_all: the requests that queue up together sleep on one gate and leave together.
// The gate has a mutex of its own and a settled request is visible to its thread without the scheduler's lock, so the
// woken threads do not queue up on that lock (they did: 128 one-block threads got 57 k blocks/s, now 165 k).
//
// Results do not depend on the combiner being on or off beyond float32 rounding: a stream's arithmetic is
// the same, but the K1/K2/K3 launch forms are chosen from the batch shape, and forms differ in summation
// order.  One-block requests that travel in batches of at most 64 blocks use the same forms
// as a lone block and are bit-identical to it.
#pragma once

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/folve_engine.h"

namespace folve {

class BatchScheduler {
public:
    struct Stats {
        long long requests = 0;     // requests submitted
        long long blocks = 0;       // blocks in them
        long long batches = 0;      // engine calls issued
        long long largest = 0;      // most blocks in one engine call
        long long overlapped = 0;   // batches submitted while another was still on the GPU
    };
    struct Request;                 // a submitted span; owned by the scheduler until Wait returns

    // The scheduler serving `engine` (created on first use; freed by ReleaseEngine or at exit).
    static BatchScheduler* ForEngine(fe_engine* engine);
    static void ReleaseEngine(fe_engine* engine);
    // Process-wide switch: on by default; off = every Process call goes to the engine by itself
    // (FOLVE_AMD_BATCH=0 in the environment does the same).
    static void SetEnabled(bool on);
    static bool Enabled();
    // Upper bound of requests per engine call.
    static void Configure(int max_batch);
    // How full the queue must be for a second batch to leave while one is running: queued blocks >= quarters / 4 of the
    // blocks in flight (default 2: half as many).
    static void SetEarlyQuarters(int quarters);

    // Enqueue `frames` interleaved frames (ceil(frames / block) blocks, the last zero-padded) of stream `s`:
    // never waits for the GPU.  `in` and `out` must lie in page-locked memory bound to the stream and stay
    // untouched until Wait returns.  One request per stream at a time.
    // block_peaks (optional): room for 2 floats per block — when the request comes back through a ticket the engine has
    // filled in every block's signed maximum (never below 0) and maximum magnitude (fe_batch_submit_peaks).
    Request* Submit(fe_stream* s, const float* in, long long frames, float* out, float* block_peaks = nullptr);
    // Blocks until the request has been computed; returns the engine's status for THIS request and frees it.
    // *peaks_filled: block_peaks holds the maxima (false after the block-by-block fallback of a refused batch: scan the output).
    int Wait(Request* r, std::string* error, bool* peaks_filled = nullptr);
    // True once Wait would not block.
    bool Ready(Request* r);

    // One block for one stream, exactly fe_stream_process without the peaks (the caller scans its
    // own output, as sound-processor.cc:116-125 does); blocks until the block has been computed.
    int Process(fe_stream* s, const float* in, int valid_frames, float* out, std::string* error);

    Stats stats();

private:
    static const int kLanes = 2;
    struct Gate {                                   // what the threads of one queue generation sleep on
        std::mutex m;
        std::condition_variable cv;
        unsigned long long epoch = 0;               // under m: counts the wake-up calls
    };
    struct Batch {
        std::vector<Request*> reqs;
        fe_ticket* ticket = nullptr;
        bool has_waiter = false;
        bool done = false;
        long long blocks = 0;
    };
    enum State { kQueued = 0, kFlying, kDone };

    BatchScheduler() {}
    void PumpLocked(std::unique_lock<std::mutex>& lk);                    // queue -> batches while a lane is free
    void CompleteLocked(std::unique_lock<std::mutex>& lk, const std::shared_ptr<Batch>& b);   // wait for b's ticket, settle, pump
    void RunAlone(Request* r);                                            // synchronous engine call for one request
    Request* SubmitLocked(std::unique_lock<std::mutex>& lk, fe_stream* s, const float* in, long long frames, float* out, float* block_peaks);
    void WaitLocked(std::unique_lock<std::mutex>& lk, Request* r);        // returns with the lock released, the request settled
    bool SleepOnGate(std::unique_lock<std::mutex>& lk, Request* r);
    static void WakeGate(Gate* g, bool all);

    std::mutex mu_;
    std::deque<Request*> queue_;
    std::shared_ptr<Gate> next_gate_;           // the gate of the requests queueing up now
    std::vector<std::shared_ptr<Batch>> flying_;
    int lanes_busy_ = 0;                        // batches on the GPU + batches being submitted + lone synchronous calls
    long long queued_blocks_ = 0;               // blocks in queue_
    long long flying_blocks_ = 0;               // blocks in the batches counted by lanes_busy_
    bool MayPumpLocked() const;
    bool pumping_ = false;                      // one thread at a time turns the queue into batches (keeps submission order)
    Stats stats_;
};

}  // namespace folve
