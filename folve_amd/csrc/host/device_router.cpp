#include "device_router.h"

#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <thread>

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

namespace {
double Now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return static_cast<double>(ts.tv_sec) + 1e-9 * static_cast<double>(ts.tv_nsec);
}
}  // namespace

DeviceRouter::DeviceRouter(const std::vector<int>& devices) {
    for (int d : devices) slots_.push_back(std::unique_ptr<Slot>(new Slot(d)));
    if (const char* env = getenv("FOLVE_AMD_FENCE_AFTER")) if (atoi(env) > 0) fence_after_ = atoi(env);
    if (const char* env = getenv("FOLVE_AMD_REPROBE_SECONDS")) if (atof(env) >= 0) reprobe_s_ = atof(env);
}

DeviceRouter::~DeviceRouter() {
    {
        // A probe thread still out there touches the slots: wait for it — for a bounded time.  A probe that hangs for good
        // (the case its thread is detached for) must not hang the teardown too: then the slots and engines are left to
        // the process's end instead of being destroyed under a thread that may still wake up.
        std::unique_lock<std::mutex> lk(mu_);
        const bool quiet = probed_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds(10),
                                              [this] { return probes_in_flight_ == 0; });
        if (!quiet) {
            Logf("DeviceRouter: a GPU probe has not come back after 10 s: leaving %zu slots to the end of the process", slots_.size());
            for (auto& sp : slots_) (void)sp.release();                          // (leaked on purpose: the probe thread holds pointers into them)
            filters_.clear();
            return;
        }
    }
    for (auto& kv : filters_) fe_filter_release(kv.second.filter);
    for (auto& s : slots_)
        if (s->engine) fe_engine_destroy(s->engine);
}

DeviceRouter::Slot* DeviceRouter::SlotOfLocked(fe_engine* e) const {
    if (!e) return NULL;
    for (auto& s : slots_) if (s->engine == e) return s.get();
    return NULL;
}

// One look at a slot that is not healthy: (re)create its engine if it never came up, then a small round trip through
// it.  The look runs on a thread of its own and the opener waits for it for a bounded time only — a GPU that hangs
// must not hang the file that is being opened; `probing` keeps everybody off the slot until the look has returned,
// however late.  The lock is dropped while waiting.
bool DeviceRouter::ProbeSlot(Slot* s, std::unique_lock<std::mutex>* lk) {
    s->probing = true;
    probes_in_flight_++;
    const int slot_index = static_cast<int>(std::find_if(slots_.begin(), slots_.end(),
                                                         [&](const std::unique_ptr<Slot>& p) { return p.get() == s; }) - slots_.begin());
    auto look = [this, s, slot_index] {
        fe_engine* e;
        { std::lock_guard<std::mutex> g(mu_); e = s->engine; }
        bool ok = true;
        fe_engine* created = NULL;
        if (!e) {
            ok = fe_engine_create(s->device, NULL, &created) == 0;
            e = created;
        }
        if (ok) ok = fe_engine_probe(e) == 0;
        const std::string why = ok ? std::string() : std::string(fe_last_error());
        {
            std::lock_guard<std::mutex> g(mu_);
            if (created) s->engine = created;        // nobody else creates it while `probing`
            s->looked_at = Now();
            if (ok) {
                if (s->state.load() != kHealthy) Logf("GPU %d (slot %d) answers again: back in service", s->device, slot_index);
                s->state.store(kHealthy);
                s->fail_streak = 0;
            } else {
                s->failures++;
                s->fail_streak++;
                s->state.store(kFenced);             // it was asked directly and said no
                Logf("GPU %d (slot %d) still unusable: %s", s->device, slot_index, why.c_str());
            }
            s->probing = false;
            probes_in_flight_--;
            probed_.notify_all();                    // (under the lock: ~DeviceRouter may destroy the condition variable the moment it sees 0)
        }
    };
    try {
        std::thread(look).detach();
    } catch (...) {
        // no thread to be had: look from here (unbounded, but better than leaving the slot marked as being probed for good)
        lk->unlock();
        look();
        lk->lock();
    }
    // (wait_until on the system clock: pthread_cond_timedwait, which ThreadSanitizer knows, unlike the clockwait of wait_for)
    probed_.wait_until(*lk, std::chrono::system_clock::now() + std::chrono::microseconds(static_cast<long long>(probe_wait_s_ * 1e6)),
                       [s] { return !s->probing; });
    if (s->probing) {
        Logf("GPU %d (slot %d) does not answer its probe: left out until it does", s->device, slot_index);
        return false;
    }
    return s->state.load() == kHealthy;
}

// The least-loaded healthy slot; suspect ones only when no healthy one is left, fenced ones never — but any slot that is
// not healthy and has not been looked at for the re-probe interval is probed first, and when nothing else is left the
// least recently probed ones are probed at once (a daemon whose only GPU hiccupped must find it again).
fe_engine* DeviceRouter::PickEngine(const std::vector<fe_engine*>* tried) {
    std::unique_lock<std::mutex> lk(mu_);
    auto was_tried = [&](const Slot* s) {
        if (!tried || !s->engine) return false;
        for (fe_engine* t : *tried) if (t == s->engine) return true;
        return false;
    };
    std::set<const Slot*> gave_up;                   // slots this very call has probed (or waited for) without success
    // What one open may spend on probes and waits in all: with every GPU hung, each slot in turn would cost a probe wait
    // (8 x 2 s per open); past this point sick slots are no longer looked at and the open takes what is healthy, or fails.
    const double t_end = Now() + 2.0 * probe_wait_s_ + 0.5;
    for (;;) {
        const double now = Now();
        Slot* best = NULL;
        Slot* due = NULL;                            // a sick slot whose re-probe is due
        Slot* sick = NULL;                           // the sick slot looked at longest ago (last resort)
        Slot* busy = NULL;                           // a slot another thread is probing right now
        for (auto& sp : slots_) {
            Slot* s = sp.get();
            if (was_tried(s) || gave_up.count(s)) continue;
            if (s->probing) { busy = s; continue; }
            const int st = s->state.load();
            if (st != kHealthy) {
                if (now - s->looked_at >= reprobe_s_ && (!due || s->looked_at < due->looked_at)) due = s;
                if (!sick || s->looked_at < sick->looked_at) sick = s;
            }
            if (st == kFenced) continue;
            if (!best || st < best->state.load() || (st == best->state.load() && s->live < best->live)) best = s;
        }
        if (due && now < t_end) {
            if (!ProbeSlot(due, &lk)) gave_up.insert(due);
            continue;                                // choose again with what the probe found
        }
        if (!best) {
            if (now >= t_end) return NULL;
            if (sick) {
                if (!ProbeSlot(sick, &lk)) gave_up.insert(sick);
                continue;
            }
            if (!busy) return NULL;
            // nothing else is left and somebody is asking that GPU right now: its answer is worth a bounded wait (two files
            // opened at the same moment on a box whose only GPU hiccupped must not fail the second one)
            probed_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(static_cast<long long>(probe_wait_s_ * 1e6)),
                               [busy] { return !busy->probing; });
            if (busy->probing || busy->state.load() != kHealthy) gave_up.insert(busy);
            continue;
        }
        if (!best->engine) {
            // first use of this GPU.  Creating an engine takes a while (context, tables): still under the lock, as in
            // round 3 — it happens once per GPU.
            if (fe_engine_create(best->device, NULL, &best->engine) != 0) {
                Logf("GPU %d unusable: %s", best->device, fe_last_error());
                best->engine = NULL;
                best->failures++;
                best->fail_streak = fence_after_;
                best->state.store(kFenced);
                best->looked_at = Now();
                gave_up.insert(best);
                continue;                            // the next GPU
            }
        }
        best->live++;      // reserved under the same lock as the choice: concurrent opens alternate exactly
        return best->engine;
    }
}

fe_engine* DeviceRouter::EngineForDevice(int device) {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& sp : slots_) {
        Slot& s = *sp;
        if (s.device != device) continue;
        if (!s.engine && fe_engine_create(s.device, NULL, &s.engine) != 0) {
            Logf("GPU %d unusable: %s", s.device, fe_last_error());
            s.engine = NULL;
            return NULL;
        }
        return s.engine;
    }
    return NULL;
}

fe_engine* DeviceRouter::EngineIfCreated(int slot) {
    std::lock_guard<std::mutex> lk(mu_);
    return (slot >= 0 && slot < static_cast<int>(slots_.size())) ? slots_[static_cast<size_t>(slot)]->engine : NULL;
}

void DeviceRouter::StreamOpened(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    if (Slot* s = SlotOfLocked(e)) s->live++;
}

void DeviceRouter::StreamClosed(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    Slot* s = SlotOfLocked(e);
    if (s && s->live > 0) s->live--;
}

void DeviceRouter::ReportFailure(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    Slot* s = SlotOfLocked(e);
    if (!s) return;
    s->failures++;
    s->fail_streak++;
    s->looked_at = Now();
    const int was = s->state.load();
    const int now = s->fail_streak >= fence_after_ ? kFenced : kSuspect;
    if (now > was) {
        s->state.store(now);
        if (now == kFenced)
            Logf("GPU %d (slot %d) fenced after %d consecutive failures: new files go to the other GPUs", s->device,
                 static_cast<int>(std::find_if(slots_.begin(), slots_.end(), [&](const std::unique_ptr<Slot>& q) { return q.get() == s; }) - slots_.begin()),
                 s->fail_streak);
    }
}

void DeviceRouter::ReportSuccess(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    Slot* s = SlotOfLocked(e);
    if (!s || (s->state.load() == kHealthy && s->fail_streak == 0)) return;
    s->fail_streak = 0;
    s->state.store(kHealthy);
}

const std::atomic<int>* DeviceRouter::HealthFlag(fe_engine* e) {
    std::lock_guard<std::mutex> lk(mu_);
    Slot* s = SlotOfLocked(e);
    return s ? &s->state : NULL;
}

bool DeviceRouter::EngineUsable(fe_engine* e) const {
    std::lock_guard<std::mutex> lk(mu_);
    const Slot* s = SlotOfLocked(e);
    return !s || s->state.load() != kFenced;
}

DeviceRouter::SlotState DeviceRouter::slot_state(int slot) const {
    std::lock_guard<std::mutex> lk(mu_);
    return (slot >= 0 && slot < static_cast<int>(slots_.size()))
               ? static_cast<SlotState>(slots_[static_cast<size_t>(slot)]->state.load()) : kFenced;
}

long long DeviceRouter::slot_failures(int slot) const {
    std::lock_guard<std::mutex> lk(mu_);
    return (slot >= 0 && slot < static_cast<int>(slots_.size())) ? slots_[static_cast<size_t>(slot)]->failures : 0;
}

void DeviceRouter::SetFenceAfter(int n) {
    std::lock_guard<std::mutex> lk(mu_);
    fence_after_ = n > 0 ? n : 3;
}

void DeviceRouter::SetProbeWaitSeconds(double seconds) {
    std::lock_guard<std::mutex> lk(mu_);
    probe_wait_s_ = seconds > 0 ? seconds : 2.0;
}

void DeviceRouter::SetReprobeSeconds(double seconds) {
    std::lock_guard<std::mutex> lk(mu_);
    reprobe_s_ = seconds >= 0 ? seconds : 10.0;
}

int DeviceRouter::cached_filters() const {
    std::lock_guard<std::mutex> lk(mu_);
    return static_cast<int>(filters_.size());
}

int DeviceRouter::live_streams(int slot) const {
    std::lock_guard<std::mutex> lk(mu_);
    return (slot >= 0 && slot < static_cast<int>(slots_.size())) ? slots_[static_cast<size_t>(slot)]->live : 0;
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
                                   int channels, ZitaConfig* out_cfg, FileStamps* impulse_files, bool* engine_fault) {
    const std::pair<std::string, fe_engine*> key(config_file, engine);
    {
        // The reference serialises Create() as a whole (sound-processor.cc:43); here only the bookkeeping is
        // serialised, and two threads never build the same (configuration, GPU) at once.
        std::unique_lock<std::mutex> lk(mu_);
        // (the sweep stats every cached configuration and its impulse files under the lock: at most once a second,
        // not on every open — the reference looks at one mtime per open, sound-processor.cc:129-133)
        if (const double now = Now(); now - last_sweep_ >= 1.0) {
            last_sweep_ = now;
            SweepLocked();
        }
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
        if (engine_fault) *engine_fault = true;          // it parsed: the transform is what failed
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
