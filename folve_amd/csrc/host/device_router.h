// device_router.h — the ProcessorPool's GPU sharder.
//
// folve hands one SoundProcessor to one open file; streams are independent, so
// the multi-GPU form of the pool is "pick a GPU per new processor, keep it there"
// (SURVEY.md §8e).  One engine per visible device, created lazily; a new stream
// goes to the device with the fewest live streams.  Committed filters are cached
// per (config path, mtime, slot) so that all streams of one configuration on a
// GPU share a single set of spectra — which is also what lets them be batched.
// Parsing a configuration and transforming its taps (hundreds of milliseconds for a
// long impulse response) happens OUTSIDE the router's lock: the key is marked as
// being built, other threads asking for the same key wait for it, and everybody
// else — PickEngine / StreamClosed for any GPU, other configurations — carries on.
//
// GPU health.  On an 8-GPU node one bad device must not turn the whole daemon into
// pass-through (folve falls back to the unfiltered file when no processor can be had,
// folve-filesystem.cc:78-88).  The reference's unit of failure handling is the processor:
// the pool discards what it cannot use and creates a new one (processor-pool.cc:71-77).
// Here the unit underneath is the GPU slot: a slot whose engine cannot be created, whose
// filter cannot be transformed or whose calls fail is SUSPECT (new streams prefer the
// other slots) and after kFenceAfter consecutive failures FENCED (new streams never go
// there).  A slot that is not healthy is re-probed — a small round trip through its
// engine, fe_engine_probe — when a stream is opened and the last look at it is older than
// the re-probe interval (FOLVE_AMD_REPROBE_SECONDS, default 10), and at once when no other
// slot is left; a successful call of any stream still living there clears it too.
// SoundProcessor::Create moves on to the next slot when one fails under it, so an open
// fails only when every GPU has.
#pragma once

#include <time.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "zita_config.h"

namespace folve {

class DeviceRouter {
public:
    // Process-wide router over all visible devices (or FOLVE_AMD_DEVICES="0,2,3").
    static DeviceRouter* Default();
    explicit DeviceRouter(const std::vector<int>& devices);
    ~DeviceRouter();

    int device_count() const { return static_cast<int>(slots_.size()); }
    // Engine of the least-loaded device (created on first use); NULL without a GPU.  The pick
    // reserves a stream on that device (as StreamOpened would): give it back with StreamClosed
    // if no stream comes of it.
    // `tried` (optional): engines this open has already failed on; their slots are passed over.
    fe_engine* PickEngine(const std::vector<fe_engine*>* tried = NULL);
    fe_engine* EngineForDevice(int device);
    fe_engine* EngineIfCreated(int slot);      // NULL if that slot's engine was never needed
    void StreamOpened(fe_engine* e);
    void StreamClosed(fe_engine* e);
    int live_streams(int slot) const;

    // ---- health (see the head of this file) ----
    enum SlotState { kHealthy = 0, kSuspect = 1, kFenced = 2 };
    // A call, a stream open or a filter transform on `e` failed for a reason that is the GPU's, not the configuration's.
    void ReportFailure(fe_engine* e);
    // A call on `e` succeeded.  Cheap when the slot is healthy (one relaxed load: see HealthFlag).
    void ReportSuccess(fe_engine* e);
    // Non-zero while the slot of `e` is not healthy; stays valid for the router's lifetime.  A processor keeps the pointer
    // and calls ReportSuccess only when it reads non-zero, so the steady state never touches the router's lock.
    const std::atomic<int>* HealthFlag(fe_engine* e);
    bool EngineUsable(fe_engine* e) const;     // false while the slot of `e` is fenced: pooled processors there are discarded
    SlotState slot_state(int slot) const;
    long long slot_failures(int slot) const;   // failures reported for the slot since the process started
    void SetFenceAfter(int consecutive_failures);       // default 3 (FOLVE_AMD_FENCE_AFTER)
    void SetReprobeSeconds(double seconds);             // default 10 (FOLVE_AMD_REPROBE_SECONDS); 0: look again at every open
    void SetProbeWaitSeconds(double seconds);           // how long an open waits for a probe to answer (default 2)
    int cached_filters() const;                // committed filters held: one per (configuration, slot) in use

    // Parsed + committed filter for (config, mtime) on `engine`; NULL if the
    // configuration is broken.  The caller gets its own reference.
    // *impulse_files (optional) receives the impulse files the configuration reads, with their modification times at
    // the time the filter was built.
    typedef std::vector<std::pair<std::string, time_t>> FileStamps;
    fe_filter* GetFilter(fe_engine* engine, const std::string& config_file, time_t mtime, int samplerate,
                         int channels, ZitaConfig* out_cfg, FileStamps* impulse_files = NULL, bool* engine_fault = NULL);
    // *engine_fault (optional) is set when the failure was the GPU's (the filter parsed, its transform did not run): the same
    // configuration may well work on another slot.
    // True if every file still has the modification time recorded for it.
    static bool StampsCurrent(const FileStamps& files);

private:
    struct Slot {
        int device; fe_engine* engine = NULL; int live = 0;
        std::atomic<int> state{kHealthy};      // written under mu_; read without it by HealthFlag's holders
        int fail_streak = 0;                   // consecutive failures (reset by a success or a good probe)
        long long failures = 0;
        double looked_at = 0;                  // monotonic seconds of the last failure or probe
        bool probing = false;                  // some thread is probing it right now (outside the lock)
        explicit Slot(int d) : device(d) {}
    };
    Slot* SlotOfLocked(fe_engine* e) const;
    bool ProbeSlot(Slot* s, std::unique_lock<std::mutex>* lk);   // drops the lock around the probe; true if the slot is healthy now
    struct CachedFilter { fe_filter* filter; ZitaConfig cfg; time_t mtime; FileStamps files; };
    void SweepLocked();                        // drop cached filters nobody uses whose configuration changed or vanished
    mutable std::mutex mu_;
    std::condition_variable built_;            // a filter that was being built has been cached (or has failed)
    std::set<std::pair<std::string, fe_engine*>> building_;   // keys whose filter some thread is parsing / transforming right now
    std::vector<std::unique_ptr<Slot>> slots_;     // fixed at construction (Slot holds an atomic: not movable)
    double last_sweep_ = -1e9;
    int fence_after_ = 3;
    double reprobe_s_ = 10.0;
    double probe_wait_s_ = 2.0;                // how long an open waits for a probe before it moves on
    int probes_in_flight_ = 0;
    std::condition_variable probed_;
    std::map<std::pair<std::string, fe_engine*>, CachedFilter> filters_;   // (config path, engine of a slot) -> filter
};

}  // namespace folve
