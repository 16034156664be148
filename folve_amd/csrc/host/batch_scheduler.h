// batch_scheduler.h — coalesces per-file block requests into GPU batches.
//
// In folve every open file is pulled by its own thread: a FUSE worker in
// ConversionBuffer::FillUntil (conversion-buffer.cc:151-163) or the BufferThread
// running ahead of the reader (buffer-thread.cc:73-105), one 8192-frame block per
// SoundProcessor::Process call.  One block is far too little work for a GPU launch,
// so with batching enabled Process() does not launch: it hands its block to the
// scheduler of its GPU and sleeps; a dispatcher thread collects the blocks that
// arrive within a short window and submits them as ONE fe_batch_process call.
// Results are bit-identical to unbatched calls (same kernels, same per-stream math).
#pragma once

#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../include/folve_engine.h"

namespace folve {

class BatchScheduler {
public:
    struct Stats {
        long long requests = 0;     // blocks submitted
        long long batches = 0;      // fe_batch_process calls issued
        long long largest = 0;      // most blocks in one batch
    };

    // The scheduler serving `engine` (created on first use, lives for the process).
    static BatchScheduler* ForEngine(fe_engine* engine);
    // Process-wide switch (also FOLVE_AMD_BATCH=1); off by default.
    static void SetEnabled(bool on);
    static bool Enabled();
    // Collection window in microseconds and batch size cap.
    static void Configure(int window_us, int max_batch);

    // One block for one stream, exactly fe_stream_process; blocks the caller until its
    // batch has run.  Returns the engine's status for the batch.
    int Process(fe_stream* s, const float* in, int valid_frames, float* out, float* peak_signed, float* peak_abs);

    Stats stats();

private:
    struct Request {
        fe_stream* s;
        const float* in;
        int frames;
        float* out;
        float peak_signed, peak_abs;
        int rc;
        bool done;
    };
    explicit BatchScheduler(fe_engine* e);
    void Loop();

    std::mutex mu_;
    std::condition_variable arrived_, finished_;
    std::vector<Request*> queue_;
    std::thread worker_;
    Stats stats_;
};

}  // namespace folve
