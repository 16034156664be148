// trace.cpp — see trace.h.
#include "trace.h"

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <mutex>

namespace ftrace {
namespace {

typedef int (*push_fn)(const char*);
typedef int (*pop_fn)();
struct Roctx {
    push_fn push = nullptr;
    pop_fn pop = nullptr;
    Roctx() {
        const char* env = getenv("FOLVE_AMD_ROCTX");
        const bool asked = env && env[0] && strcmp(env, "0") != 0;
        if (env && !asked) return;                                   // FOLVE_AMD_ROCTX=0: off, whatever is loaded
        static const char* const libs[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        for (const char* name : libs) {
            // asked for: load it; not asked: only if the process (a profiler) has it already
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL | (asked ? 0 : RTLD_NOLOAD));
            if (!h) continue;
            push_fn p = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
            pop_fn q = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
            if (p && q) { push = p; pop = q; return; }
        }
        if (asked) fprintf(stderr, "folve_amd: FOLVE_AMD_ROCTX is set but no roctx library could be loaded: no ranges\n");
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}

struct Events {
    FILE* f = nullptr;
    std::mutex mu;
    struct timespec t0;
    Events() {
        clock_gettime(CLOCK_MONOTONIC, &t0);
        const char* path = getenv("FOLVE_AMD_TRACE");
        if (path && path[0]) {
            f = strcmp(path, "-") == 0 ? stderr : fopen(path, "a");
            if (!f) fprintf(stderr, "folve_amd: cannot open FOLVE_AMD_TRACE=%s\n", path);
        }
    }
};
Events& events() {
    static Events e;
    return e;
}

}  // namespace

bool roctx_on() { return roctx().push != nullptr; }

void range_push(const char* fmt, ...) {
    const Roctx& r = roctx();
    if (!r.push) return;
    char buf[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    r.push(buf);
}

void range_pop() {
    const Roctx& r = roctx();
    if (r.pop) r.pop();
}

bool events_on() { return events().f != nullptr; }

void event(const char* fmt, ...) {
    Events& e = events();
    if (!e.f) return;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    const long long us = (t.tv_sec - e.t0.tv_sec) * 1000000LL + (t.tv_nsec - e.t0.tv_nsec) / 1000;
    char buf[320];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(e.mu);
    fprintf(e.f, "%lld %ld %s\n", us, static_cast<long>(syscall(SYS_gettid)), buf);
    fflush(e.f);
}

}  // namespace ftrace
