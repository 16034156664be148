#include "numa_placement.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../../../include/folve_engine.h"

namespace folve {

namespace {
std::atomic<int> g_numa{-1};

// "0-31,128-159" -> set
bool ParseCpuList(const char* s, cpu_set_t* out) {
    CPU_ZERO(out);
    bool any = false;
    while (*s) {
        char* end = NULL;
        const long a = strtol(s, &end, 10);
        if (end == s) break;
        long b = a;
        s = end;
        if (*s == '-') {
            b = strtol(s + 1, &end, 10);
            if (end == s + 1) break;
            s = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0) { CPU_SET(static_cast<int>(c), out); any = true; }
        if (*s == ',') ++s;
        else break;
    }
    return any;
}
}  // namespace

void SetNumaPlacement(bool on) { g_numa.store(on ? 1 : 0); }

bool NumaPlacement() {
    int v = g_numa.load();
    if (v < 0) {
        const char* env = getenv("FOLVE_AMD_NUMA");
        v = (env && atoi(env) != 0) ? 1 : 0;
        g_numa.store(v);
    }
    return v == 1;
}

bool DeviceLocalCpus(int device, cpu_set_t* out) {
    char list[1024];
    if (fe_device_local_cpulist(device, list, sizeof(list)) != 0) return false;
    cpu_set_t local, allowed;
    if (!ParseCpuList(list, &local)) return false;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    CPU_AND(out, &local, &allowed);
    return CPU_COUNT(out) > 0;
}

bool PinThreadNearDevice(int device) {
    cpu_set_t set;
    if (!DeviceLocalCpus(device, &set)) return false;
    return sched_setaffinity(0, sizeof(set), &set) == 0;
}

ScopedDeviceAffinity::ScopedDeviceAffinity(int device) : moved_(false) {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(saved_), &saved_) != 0) return;
    if (!DeviceLocalCpus(device, &set)) return;
    moved_ = sched_setaffinity(0, sizeof(set), &set) == 0;
}

ScopedDeviceAffinity::~ScopedDeviceAffinity() {
    if (moved_) (void)sched_setaffinity(0, sizeof(saved_), &saved_);
}

}  // namespace folve
