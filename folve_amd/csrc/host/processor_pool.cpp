// This file restates the interface and behaviour of folve's processor-pool.cc, Copyright (C) 2012 Henner Zeller
// <h.zeller@acm.org>, free software under the GNU General Public License, version 3 or (at your option) any later
// version; this restatement is distributed under the same terms, WITHOUT ANY WARRANTY (<http://www.gnu.org/licenses/>).
#include "processor_pool.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "device_router.h"
#include "../trace.h"
#include "sound_processor.h"

namespace folve {

namespace {

std::string Fmt(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
std::string Fmt(const char* fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return buf;
}

bool FirstReadable(const std::vector<std::string>& candidates, std::string* match) {
    for (const std::string& c : candidates) {
        if (access(c.c_str(), R_OK) == 0) {
            *match = c;
            return true;
        }
    }
    return false;
}

}  // namespace

ProcessorPool::ProcessorPool(int max_available) : max_per_config_(static_cast<size_t>(max_available)) {}

ProcessorPool::~ProcessorPool() {
    for (auto& kv : pool_) {
        for (SoundProcessor* p : *kv.second) delete p;
        delete kv.second;
    }
}

SoundProcessor* ProcessorPool::GetOrCreate(const std::string& base_dir, int sampling_rate, int channels, int bits,
                                           std::string* errmsg) {
    // From specific to non-specific (processor-pool.cc:53-61).
    std::vector<std::string> path_choices;
    path_choices.push_back(Fmt("%s/filter-%d-%d-%d.conf", base_dir.c_str(), sampling_rate, channels, bits));
    path_choices.push_back(Fmt("%s/filter-%d-%d.conf", base_dir.c_str(), sampling_rate, channels));
    path_choices.push_back(Fmt("%s/filter-%d.conf", base_dir.c_str(), sampling_rate));

    std::string config_path;
    if (!FirstReadable(path_choices, &config_path)) {
        const char* slash = strrchr(base_dir.c_str(), '/');
        const char* short_dir = slash ? slash + 1 : base_dir.c_str();
        *errmsg = Fmt("No filter in %s for %.1fkHz/%d ch/%d bits", short_dir, sampling_rate / 1000.0, channels, bits);
        return NULL;
    }
    SoundProcessor* result;
    while ((result = CheckOutOfPool(config_path)) != NULL) {
        if (!result->ConfigStillUpToDate()) {
            Logf("Processor %p: outdated; config file changed %s", static_cast<void*>(result), config_path.c_str());
            delete result;
            continue;
        }
        // the same discard-and-look-again for a processor whose GPU has been fenced since it was pooled: it would
        // only hand its next file silence
        if (!DeviceRouter::Default()->EngineUsable(result->engine())) {
            Logf("Processor %p: discarded; its GPU %d is fenced", static_cast<void*>(result), result->device());
            delete result;
            continue;
        }
        break;
    }
    if (result != NULL) {
        if (ftrace::events_on()) ftrace::event("GetOrCreate pooled processor=%p gpu=%d config=%s", static_cast<void*>(result), result->device(), config_path.c_str());
        return result;
    }

    result = SoundProcessor::Create(config_path, sampling_rate, channels);
    if (result == NULL) {
        *errmsg = "Problem parsing " + config_path;
        Logf("filter-config %s is broken.", config_path.c_str());
    }
    if (ftrace::events_on()) ftrace::event("GetOrCreate created processor=%p gpu=%d config=%s", static_cast<void*>(result), result ? result->device() : -1, config_path.c_str());
    return result;
}

void ProcessorPool::Return(SoundProcessor* processor) {
    if (processor == NULL) return;
    if (ftrace::events_on()) ftrace::event("Return processor=%p gpu=%d ok=%d moves=%d peak=%g", static_cast<void*>(processor), processor->device(), (int)processor->ok(), processor->moves(), processor->max_output_value());
    if (!processor->ConfigStillUpToDate()) {
        delete processor;     // outdated: not returning it to the pool
        return;
    }
    if (!processor->ok() || !DeviceRouter::Default()->EngineUsable(processor->engine())) {
        // The GPU failed under this processor: its convolver state is undefined (folve_engine.h),
        // so it must not be handed to the next file.
        Logf("Processor %p: discarded after an engine failure", static_cast<void*>(processor));
        delete processor;
        return;
    }
    std::lock_guard<std::mutex> l(pool_mutex_);
    ProcessorList*& list = pool_[processor->config_file()];
    if (list == NULL) list = new ProcessorList();
    if (list->size() < max_per_config_) {
        processor->Reset();
        list->push_back(processor);
    } else {
        delete processor;     // enough processors in pool
    }
}

SoundProcessor* ProcessorPool::CheckOutOfPool(const std::string& config_path) {
    std::lock_guard<std::mutex> l(pool_mutex_);
    PoolMap::iterator found = pool_.find(config_path);
    if (found == pool_.end()) return NULL;
    ProcessorList* list = found->second;
    if (list->empty()) return NULL;
    SoundProcessor* result = list->front();
    list->pop_front();
    return result;
}

size_t ProcessorPool::pooled_count(const std::string& config_path) {
    std::lock_guard<std::mutex> l(pool_mutex_);
    PoolMap::iterator found = pool_.find(config_path);
    return (found == pool_.end()) ? 0 : found->second->size();
}

}  // namespace folve
