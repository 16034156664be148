// This file restates the interface and behaviour of folve's processor-pool.h, Copyright (C) 2012 Henner Zeller
// <h.zeller@acm.org>, free software under the GNU General Public License, version 3 or (at your option) any later
// version; this restatement is distributed under the same terms, WITHOUT ANY WARRANTY (<http://www.gnu.org/licenses/>).
// processor_pool.h — drop-in for folve's ProcessorPool (processor-pool.h:30-55).
//
// Same contract: an object pool of SoundProcessors keyed by the resolved
// configuration path, at most `max_per_config` idle processors per key, FIFO
// reuse, stale processors (config file touched) discarded on checkout and on
// return.  New here: processors are created on the GPU the DeviceRouter picks,
// so concurrent files spread over all MI355X of the node.
#pragma once

#include <deque>
#include <map>
#include <mutex>
#include <string>

namespace folve {

class SoundProcessor;

class ProcessorPool {
public:
    // Stores at most "max_per_config" processors in pool per configuration file.
    explicit ProcessorPool(int max_per_config);
    ~ProcessorPool();

    // Get a SoundProcessor from this pool with the given configuration.  If this
    // isn't possible, NULL is returned and an error message stored in "errmsg".
    SoundProcessor* GetOrCreate(const std::string& base_dir, int sampling_rate, int channels, int bits,
                                std::string* errmsg);

    // Return a processor back to the pool.
    void Return(SoundProcessor* processor);

    size_t pooled_count(const std::string& config_path);

private:
    typedef std::deque<SoundProcessor*> ProcessorList;
    typedef std::map<std::string, ProcessorList*> PoolMap;

    SoundProcessor* CheckOutOfPool(const std::string& config_path);

    const size_t max_per_config_;
    std::mutex pool_mutex_;
    PoolMap pool_;
};

}  // namespace folve
