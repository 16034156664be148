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
#pragma once

#include <time.h>

#include <condition_variable>
#include <map>
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
    fe_engine* PickEngine();
    fe_engine* EngineForDevice(int device);
    fe_engine* EngineIfCreated(int slot);      // NULL if that slot's engine was never needed
    void StreamOpened(fe_engine* e);
    void StreamClosed(fe_engine* e);
    int live_streams(int slot) const;
    int cached_filters() const;                // committed filters held: one per (configuration, slot) in use

    // Parsed + committed filter for (config, mtime) on `engine`; NULL if the
    // configuration is broken.  The caller gets its own reference.
    // *impulse_files (optional) receives the impulse files the configuration reads, with their modification times at
    // the time the filter was built.
    typedef std::vector<std::pair<std::string, time_t>> FileStamps;
    fe_filter* GetFilter(fe_engine* engine, const std::string& config_file, time_t mtime, int samplerate,
                         int channels, ZitaConfig* out_cfg, FileStamps* impulse_files = NULL);
    // True if every file still has the modification time recorded for it.
    static bool StampsCurrent(const FileStamps& files);

private:
    struct Slot { int device; fe_engine* engine; int live; };
    struct CachedFilter { fe_filter* filter; ZitaConfig cfg; time_t mtime; FileStamps files; };
    void SweepLocked();                        // drop cached filters nobody uses whose configuration changed or vanished
    mutable std::mutex mu_;
    std::condition_variable built_;            // a filter that was being built has been cached (or has failed)
    std::set<std::pair<std::string, fe_engine*>> building_;   // keys whose filter some thread is parsing / transforming right now
    std::vector<Slot> slots_;
    std::map<std::pair<std::string, fe_engine*>, CachedFilter> filters_;   // (config path, engine of a slot) -> filter
};

}  // namespace folve
