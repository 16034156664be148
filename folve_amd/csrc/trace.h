// trace.h — tracing hooks of the library: roctx ranges around what the engine enqueues, and a
// host-side event log, the counterpart of folve's -D / -R traces (/root/reference/util.cc:62-80, folve-main.cc:63-97).
//
//   roctx   FOLVE_AMD_ROCTX=1 (or a roctx library already loaded in the process): every launch round of the engine
//           (K1 -> K2 -> K3 of one filter group: filter, streams, blocks, lane) and every chunk of the duplex DMA pipeline is
//           a roctxRangePush / Pop pair on the enqueuing thread.  `rocprofv3 --marker-trace --kernel-trace` then shows the
//           kernels under named ranges.  The library is looked up at run time (librocprofiler-sdk-roctx, then libroctx64):
//           nothing links against it, and without it the calls are two predictable branches.
//   events  FOLVE_AMD_TRACE=<file>: one line per host-layer event — pool GetOrCreate / Return, chunk submit / settle, a
//           stream moving to another GPU — as "<microseconds since start> <thread id> <event> <details>".
#pragma once

namespace ftrace {

bool roctx_on();
void range_push(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void range_pop();
struct Range {                      // a range that ends with its scope
    bool on;
    template <class... A>
    explicit Range(const char* fmt, A... a) : on(roctx_on()) { if (on) range_push(fmt, a...); }
    ~Range() { if (on) range_pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

bool events_on();
void event(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

}  // namespace ftrace
