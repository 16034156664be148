// This file restates the interface and behaviour of folve's sound-processor.h, Copyright (C) 2012 Henner Zeller
// <h.zeller@acm.org>, free software under the GNU General Public License, version 3 or (at your option) any later
// version; this restatement is distributed under the same terms, WITHOUT ANY WARRANTY (<http://www.gnu.org/licenses/>).
// sound_processor.h — drop-in for folve's SoundProcessor on top of the GPU engine.
//
// Public surface and behaviour follow /root/reference/sound-processor.h:28-85 and
// sound-processor.cc:34-145 method for method.  The reference's two libsndfile members
//     int  FillBuffer(SNDFILE *in);                          (sound-processor.h:35)
//     void WriteProcessed(SNDFILE *out, int sample_count);   (sound-processor.h:55)
// are declared here against an opaque SNDFILE and defined in sndfile_adapter.cpp, which is
// compiled only where <sndfile.h> exists (the folve build); everything else — the block
// machine itself — works on FrameSource / FrameSink (libsndfile's sf_readf_float /
// sf_writef_float contract), so the library builds and is tested without libsndfile.
//
// Run-ahead (the reference's BufferThread goal arithmetic, buffer-thread.cc:34,73-105,
// and the pull loop conversion-buffer.cc:151-163).  folve is file based and already converts ahead of its
// reader; one 8192-frame block per engine call is far too little for a GPU.  A processor therefore reads
// AHEAD of the block machine: when FillBuffer finds its ring empty it asks the source for up to N whole
// blocks at once, hands them to the GPU as ONE multi-block request (through the per-GPU combiner, which
// merges the requests of all open files), and serves FillBuffer / WriteProcessed block by block from the
// ring with exactly the return values and state (`pending_writes`, `is_input_buffer_complete`) the
// reference's calls produce (convolve-file-handler.cc:370-424).  Two chunks alternate: while the reader is
// served from one, the other is on the GPU.  The depth ramps 1, 2, 4 .. N blocks, so the first block of a
// file costs one block's latency.  Only WHOLE blocks run ahead: the short last block of a file stays in the
// block buffer unprocessed, as in the reference, so that the gapless hand-over (PassoverProcessor,
// convolve-file-handler.cc:328-351) can top it up from the next file — at that point the ring is empty by
// construction.  max_output_value() advances block by block as blocks are handed out, not as they are computed: K3 reduces
// every block's maxima on the GPU (fe_batch_submit_peaks) and the processor folds a block's two numbers in when the block
// is handed out — the reference's scan over the returned frames (sound-processor.cc:116-125) happens once, on the device.
//
// An open file survives its GPU (round 5).  The reference can never emit silence from Process() (sound-processor.cc:98-127)
// and moves a stream's state between owners at the gapless hand-over (convolve-file-handler.cc:328-351); on a node with
// eight GPUs one of them failing must not end the files that were converting on it.  The state of a stream is its last K
// input blocks (K partitions).  A run-ahead chunk's input simply stays where it was read to — its output has its own half
// of the chunk — until that buffer is read into again two chunks later; what would be overwritten while it still counts
// (single blocks, which are computed in place; chunks shorter than K: the ramp, a file's end; the tail of a long chunk
// whose buffer is re-used early) goes to a host ring of 2 K + 2 blocks first.  In the steady state of long chunks nothing
// is copied at all.  When an
// engine call fails, the processor asks the DeviceRouter for another GPU, opens a stream of the same configuration there,
// replays the kept K blocks through it (outputs discarded: that rebuilds the delay line), re-runs the failed call from
// its kept input, and carries on there — same samples as if nothing had happened.  Silence is what is left only when no
// other GPU can take the stream (or the ring would pass its memory budget: then nothing is kept).
#pragma once

#include <time.h>

#include <atomic>
#include <string>
#include <utility>
#include <vector>

#include "zita_config.h"

// <sndfile.h>'s own declaration of the handle type (a typedef may be repeated)
typedef struct SNDFILE_tag SNDFILE;

namespace folve {

// sf_readf_float(in, dst, frames): fills up to `frames` interleaved frames, returns frames read.
class FrameSource {
public:
    virtual ~FrameSource() {}
    virtual int ReadFrames(float* dst, int frames) = 0;
};

// sf_writef_float(out, src, frames): consumes `frames` interleaved frames.
class FrameSink {
public:
    virtual ~FrameSink() {}
    virtual int WriteFrames(const float* src, int frames) = 0;
};

class FilterCache;

// The workhorse of processing data from soundfiles.
class SoundProcessor {
public:
    // As the reference: NULL if the configuration cannot be parsed or does not
    // define a convolver.  The GPU is chosen by the process-wide DeviceRouter.
    static SoundProcessor* Create(const std::string& config_file, int samplerate, int channels);
    // Same, on a given engine (used by ProcessorPool's sharder).
    static SoundProcessor* CreateOn(fe_engine* engine, const std::string& config_file, int samplerate, int channels);
    // Blocks a processor may read ahead of its reader (1 = off: one block per engine call, the reference's
    // pattern; 0 = automatic).  Applies to processors created afterwards.  Default: automatic (or FOLVE_AMD_RUN_AHEAD) —
    // 64 blocks of 8192 frames, the same number of frames for shorter blocks (up to 1024 blocks), fewer blocks where
    // the page-locked ring would pass 64 MB (many channels); run_ahead() tells what a processor got.
    static void SetRunAhead(int blocks);
    static int RunAhead();              // the configured depth in blocks (64 when automatic)
    // Where a run-ahead block's maxima come from: the GPU (K3 reduces every block; default) or a scan of the block on
    // the caller's thread when it is handed out (what the reference does, sound-processor.cc:116-125).
    static void SetDevicePeaks(bool on);
    // Keep the input history that lets an open file move to another GPU when its own fails (default on; FOLVE_AMD_SURVIVE=0).
    // Applies to processors created afterwards.
    static void SetSurvival(bool on);
private:
    static SoundProcessor* CreateOnReserved(fe_engine* engine, const std::string& config_file, int samplerate, int channels,
                                            bool* engine_fault);
public:
    ~SoundProcessor();

    // Fill buffer from given source.  Returns number of frames read.
    int FillBuffer(FrameSource* in);
    // The reference's signature (sound-processor.h:35); defined in sndfile_adapter.cpp.
    int FillBuffer(SNDFILE* in);

    inline int input_channels() const { return zita_config_.ninp; }
    inline int output_channels() const { return zita_config_.nout; }

    // True if the input buffer holds a whole block for the FIR filter.
    bool is_input_buffer_complete() const { return zita_config_.fragm == input_pos_; }

    // Frames processed but not yet written (gapless hand-over, see
    // convolve-file-handler.cc:373-376).
    int pending_writes() const { return output_pos_ >= 0 ? zita_config_.fragm - output_pos_ : 0; }

    // Write `sample_count` processed frames to `out`; processes first if necessary.
    void WriteProcessed(FrameSink* out, int sample_count);
    // The reference's signature (sound-processor.h:55); defined in sndfile_adapter.cpp.
    void WriteProcessed(SNDFILE* out, int sample_count);

    // Reset processor for re-use.
    void Reset();

    // Maximum output value observed.  As in the reference (sound-processor.cc:120-123)
    // the comparison is on the signed sample; max_abs_output_value() is the magnitude.
    float max_output_value() const { return max_out_value_observed_; }
    float max_abs_output_value() const { return max_abs_value_observed_; }
    void ResetMaxValues();

    const std::string& config_file() const { return config_file_; }
    time_t config_file_timestamp() const { return config_file_timestamp_; }
    bool ConfigStillUpToDate() const;

    int block_size() const { return zita_config_.fragm; }
    int frames_wanted() const { return zita_config_.fragm - input_pos_; }   // what the reference's FillBuffer would ask its file for now
    int run_ahead() const { return run_depth_; }         // this processor's run-ahead depth in blocks
    fe_stream* stream() const { return stream_; }      // for batched submission
    fe_engine* engine() const { return engine_; }       // (changes when the processor has moved to another GPU)
    int moves() const { return moves_; }                // times this processor's stream has moved to another GPU
    int device() const;
    bool ok() const { return ok_; }                     // false after an engine failure

private:
    // A run-ahead chunk: up to run_depth_ consecutive whole blocks, read from the source in one go and
    // computed by one engine request.
    struct Chunk {
        float* in = nullptr;            // page-locked, bound to the stream
        float* out = nullptr;           // == in when ninp == nout
        int blocks = 0;                 // whole blocks it holds
        int next = 0;                   // next block to hand out
        void* request = nullptr;        // BatchScheduler::Request while on the GPU
        float* peaks = nullptr;         // [blocks][2]: every block's signed maximum and maximum magnitude, from the GPU
        bool peaks_valid = false;       // ... filled in (else the block is scanned when it is handed out)
        long long first = 0;            // its first block, counted from the last Reset (the history ring's index)
        bool holds = false;             // in[] still holds the input of blocks [first, first + blocks) (it has not been read into since)
    };
    SoundProcessor(const ZitaConfig& config, const std::string& cfg_file, fe_stream* stream, int run_depth,
                   const std::vector<std::pair<std::string, time_t>>& impulse_files, int samplerate, int channels,
                   bool keep_history);
    void Process();
    bool ReadChunk(FrameSource* in, Chunk* c);    // true if the chunk holds at least one whole block
    void SubmitChunk(Chunk* c);
    void SettleChunk(Chunk* c);                   // wait for its request; on failure: the stream moves to another GPU (or silence, ok_ = false)
    void DrainRing();
    void ScanPeaks(const float* v, size_t n);
    void EngineCallSucceeded();
    // the input of `blocks` blocks (the last one `last_frames` long, the rest of it zeros) about to be handed to the engine
    void KeepInput(const float* in, int blocks, int last_frames);
    void SaveBlocks(const float* in, long long b0, int n, int last_frames);
    const float* BlockInput(long long b) const;
    // The call for blocks [first, first + blocks) has failed: move to another GPU and produce its output there (`frames`
    // frames into `out`).  False if nothing could take the stream: the caller falls back to silence.
    bool MoveToAnotherGpu(long long first, int blocks, long long frames, float* out);

    const ZitaConfig zita_config_;
    const std::string config_file_;
    const time_t config_file_timestamp_;
    const std::vector<std::pair<std::string, time_t>> impulse_files_;   // what /impulse/read opened, and when it was last modified then
    const int samplerate_, channels_;   // what Create was asked for (another GPU's filter is looked up by them)
    fe_engine* engine_;                 // the GPU this processor's stream lives on: zita_config_.engine until a move
    fe_stream* stream_;
    float* hist_;                       // input history: hist_cap_ = 2 K + 2 blocks of fragm * ninp floats, block b in slot b % hist_cap_ (NULL: none kept)
    std::vector<long long> hist_tag_;   // the block number each slot holds (-1: none)
    int hist_cap_;
    int hist_k_;                        // the filter's partitions: blocks of history a stream's state consists of
    long long blocks_fed_;              // blocks handed to the engine since the last Reset
    int moves_;

    const int run_depth_;               // 1: no run-ahead
    const bool in_place_;               // run-ahead chunks are computed in place (only without the input history)
    const size_t buffer_floats_;
    const size_t arena_floats_;         // block buffer + both chunks, one page-locked allocation
    bool buffer_pinned_;
    float* const buffer_;
    Chunk chunks_[2];
    Chunk* cur_;                        // chunk blocks are handed out from (settled), or NULL
    Chunk* ahead_;                      // chunk on the GPU, or NULL
    const float* ring_block_;           // the current block's output inside cur_, NULL: the block buffer
    float* tail_;                       // frames read ahead that do not fill a block (the file's short last block)
    int tail_frames_;
    bool source_short_;                 // the last read-ahead came back short: no more reads until the ring is empty
    int depth_next_;                    // blocks the next chunk asks for (ramp)
    int input_pos_;
    int output_pos_;   // written position. -1, if not processed yet.
    float max_out_value_observed_;
    float max_abs_value_observed_;
    bool ok_;
    const std::atomic<int>* slot_health_;         // the router's state of this processor's GPU slot (0: healthy)
};

}  // namespace folve
