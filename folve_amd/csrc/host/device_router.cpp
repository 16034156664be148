#include "device_router.h"

#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

namespace folve {

DeviceRouter* DeviceRouter::Default() {
    static DeviceRouter* router = [] {
        std::vector<int> devs;
        if (const char* env = getenv("FOLVE_AMD_DEVICES")) {
            const char* p = env;
            while (*p) {
                char* end = NULL;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                devs.push_back(static_cast<int>(d));
                p = (*end == ',') ? end + 1 : end;
            }
        } else {
            const int n = fe_device_count();
            for (int d = 0; d < n; ++d) devs.push_back(d);
        }
        return new DeviceRouter(devs);
    }();
    return router;
}

DeviceRouter::DeviceRouter(const std::vector<int>& devices) {
    for (int d : devices) slots_.push_back(Slot{d, NULL, 0});
}

DeviceRouter::~DeviceRouter() {
    for (auto& kv : filters_) fe_filter_release(kv.second.filter);
    for (Slot& s : slots_)
        if (s.engine) fe_engine_destroy(s.engine);
}

fe_engine* DeviceRouter::PickEngine() {
    std::lock_guard<std::mutex> lk(mu_);
    Slot* best = NULL;
    for (Slot& s : slots_)
        if (!best || s.live < best->live) best = &s;
    if (!best) return NULL;
    if (!best->engine && fe_engine_create(best->device, NULL, &best->engine) != 0) {
        Logf("GPU %d unusable: %s", best->device, fe_last_error());
        return NULL;
    }
    best->live++;          // reserved under the same lock as the choice: concurrent opens alternate exactly
    return best->engine;
}

fe_engine* DeviceRouter::EngineForDevice(int device) {
    std::lock_guard<std::mutex> lk(mu_);
    for (Slot& s : slots_) {
        if (s.device != device) continue;
        if (!s.engine && fe_engine_create(s.device, NULL, &s.engine) != 0) {
            Logf("GPU %d unusable: %s", s.device, fe_last_error());
            return NULL;
        }
        return s.engine;
    }
    return NULL;
}

fe_engine* DeviceRouter::EngineIfCreated(int slot) {
    std::lock_guard<std::mutex> lk(mu_);
    return (slot >= 0 && slot < static_cast<int>(slots_.size())) ? slots_[static_cast<size_t>(slot)].engine : NULL;
}

void DeviceRouter::StreamOpened(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    for (Slot& s : slots_) if (s.engine == e) s.live++;
}

void DeviceRouter::StreamClosed(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    for (Slot& s : slots_) if (s.engine == e && s.live > 0) s.live--;
}

int DeviceRouter::cached_filters() const {
    std::lock_guard<std::mutex> lk(mu_);
    return static_cast<int>(filters_.size());
}

int DeviceRouter::live_streams(int slot) const {
    std::lock_guard<std::mutex> lk(mu_);
    return (slot >= 0 && slot < static_cast<int>(slots_.size())) ? slots_[static_cast<size_t>(slot)].live : 0;
}

// A cached filter that only the cache still holds (no stream uses it) and whose configuration file has changed or
// is gone will never be asked for again under this mtime: let go of its spectra.
bool DeviceRouter::StampsCurrent(const FileStamps& files) {
    for (const auto& f : files) {
        struct stat st;
        if (stat(f.first.c_str(), &st) != 0 || st.st_mtime != f.second) return false;
    }
    return true;
}

void DeviceRouter::SweepLocked() {
    for (auto it = filters_.begin(); it != filters_.end();) {
        struct stat st;
        const bool current = stat(it->first.first.c_str(), &st) == 0 && st.st_mtime == it->second.mtime &&
                             StampsCurrent(it->second.files);
        if (!current && fe_filter_use_count(it->second.filter) == 1) {
            fe_filter_release(it->second.filter);
            it = filters_.erase(it);
        } else {
            ++it;
        }
    }
}

fe_filter* DeviceRouter::GetFilter(fe_engine* engine, const std::string& config_file, time_t mtime, int samplerate,
                                   int channels, ZitaConfig* out_cfg, FileStamps* impulse_files) {
    const std::pair<std::string, fe_engine*> key(config_file, engine);
    {
        // The reference serialises Create() as a whole (sound-processor.cc:43); here only the bookkeeping is
        // serialised, and two threads never build the same (configuration, GPU) at once.
        std::unique_lock<std::mutex> lk(mu_);
        SweepLocked();
        for (;;) {
            auto it = filters_.find(key);
            if (it != filters_.end()) {
                if (it->second.mtime == mtime && StampsCurrent(it->second.files)) {
                    *out_cfg = it->second.cfg;
                    out_cfg->config_file = NULL;
                    if (impulse_files) *impulse_files = it->second.files;
                    fe_filter_retain(it->second.filter);
                    return it->second.filter;
                }
                fe_filter_release(it->second.filter);    // the configuration or one of its impulse files was touched: rebuild
                filters_.erase(it);
            }
            if (!building_.count(key)) break;
            built_.wait(lk);                             // somebody is building this very filter: take theirs
        }
        building_.insert(key);
    }
    struct Done {                                        // whatever happens below: the key is no longer being built
        DeviceRouter* r; const std::pair<std::string, fe_engine*>& key;
        ~Done() {
            { std::lock_guard<std::mutex> lk(r->mu_); r->building_.erase(key); }
            r->built_.notify_all();
        }
    } done{this, key};
    ZitaConfig zita;
    memset(&zita, 0, sizeof(zita));
    zita.engine = engine;
    zita.fsamp = samplerate;
    zita.ninp = channels;
    zita.nout = channels;
    FileStamps files;
    zita.on_impulse_user = &files;
    zita.on_impulse_file = [](void* user, const char* path) {
        struct stat st;
        static_cast<FileStamps*>(user)->push_back(std::make_pair(std::string(path), stat(path, &st) == 0 ? st.st_mtime : (time_t)0));
    };
    // As SoundProcessor::Create (sound-processor.cc:44-46): parse errors, or a
    // configuration that never defined a convolver, make creation fail.
    if (config(&zita, config_file.c_str()) != 0 || zita.filter == NULL) {
        if (zita.filter) fe_filter_release(zita.filter);
        return NULL;
    }
    if (fe_filter_commit(zita.filter) != 0) {
        Logf("Cannot transform filter %s on GPU %d: %s", config_file.c_str(), fe_engine_device(engine), fe_last_error());
        fe_filter_release(zita.filter);
        return NULL;
    }
    zita.config_file = NULL;
    zita.on_impulse_file = NULL;
    zita.on_impulse_user = NULL;
    if (impulse_files) *impulse_files = files;
    {
        std::lock_guard<std::mutex> lk(mu_);
        filters_[key] = CachedFilter{zita.filter, zita, mtime, files};
    }
    *out_cfg = zita;
    fe_filter_retain(zita.filter);
    return zita.filter;
}

}  // namespace folve
