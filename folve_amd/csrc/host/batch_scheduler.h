// batch_scheduler.h — coalesces per-file block requests into GPU batches, without a timer.
//
// In folve every open file is pulled by its own thread: a FUSE worker in
// ConversionBuffer::FillUntil (conversion-buffer.cc:151-163) or the BufferThread running ahead
// of the reader (buffer-thread.cc:73-105), one 8192-frame block per SoundProcessor::Process call.
// The GPU of a device serves one launch chain at a time, so concurrent calls queue anyway; this
// class turns that queue into batches ("combining"): a thread that finds its GPU idle runs its own
// block at once — no hand-off, no collection window, nothing added to a lone stream's latency —
// and while it is in flight every other thread's block is parked; whoever finishes takes ALL
// parked blocks with it as ONE fe_batch_process call.  Batches form exactly when there is
// contention and grow with it.  No dispatcher thread exists, so there is nothing to join at exit.
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
    struct Request {
        fe_stream* s;
        const float* in;
        int frames;
        float* out;
        int rc;
        bool done;
        std::string error;
    };
    BatchScheduler() {}
    void Run(std::vector<Request*>& batch);

    std::mutex mu_;
    std::condition_variable finished_;
    std::vector<Request*> queue_;
    bool busy_ = false;
    Stats stats_;
};

}  // namespace folve
