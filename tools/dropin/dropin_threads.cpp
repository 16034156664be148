// The drop-in call pattern under load: N "file" threads, each with its own folve::SoundProcessor (through the C view of
// the host classes, include/folve_host.h), each pulling 8192-frame blocks the way ConvolveFileHandler does
// (/root/reference/convolve-file-handler.cc:335-347,370-424): FillBuffer -> WriteProcessed (which runs Process()).
// Prints blocks per second over all threads, the latency of a block as a thread sees it, and what the per-GPU
// combiner made of the calls.   usage: dropin_threads <filter.conf> <threads> <blocks per thread> <batching 0|1> [json]
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "folve_host.h"

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s conf threads blocks batching\n", argv[0]); return 2; }
    const char* conf = argv[1];
    const int nthreads = atoi(argv[2]), nblocks = atoi(argv[3]), batching = atoi(argv[4]);
    fh_batching_set(batching, 0, 64);
    std::vector<fh_processor*> procs;
    for (int i = 0; i < nthreads; ++i) {
        fh_processor* p = fh_processor_create(conf, 44100, 2);
        if (!p) { fprintf(stderr, "processor %d: creation failed\n", i); return 1; }
        procs.push_back(p);
    }
    const int P = fh_processor_block_size(procs[0]);
    const int cin = fh_processor_input_channels(procs[0]), cout = fh_processor_output_channels(procs[0]);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::vector<float>> lat((size_t)nthreads);
    std::vector<std::thread> th;
    long long r0, b0, l0;
    fh_batching_stats(&r0, &b0, &l0);
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&, t] {
            std::mt19937 rng(100 + t);
            std::uniform_real_distribution<float> u(-1.f, 1.f);
            std::vector<float> src((size_t)P * cin), dst((size_t)P * cout);
            for (auto& v : src) v = u(rng);
            fh_processor* p = procs[(size_t)t];
            for (int w = 0; w < 8; ++w) {                     // warm: first launches, clocks
                fh_processor_fill_buffer(p, src.data(), P);
                fh_processor_write_processed(p, dst.data(), P);
            }
            lat[(size_t)t].reserve((size_t)nblocks);
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            for (int b = 0; b < nblocks; ++b) {
                const auto a = std::chrono::steady_clock::now();
                fh_processor_fill_buffer(p, src.data(), P);
                fh_processor_write_processed(p, dst.data(), P);
                lat[(size_t)t].push_back(std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - a).count());
            }
        });
    }
    while (ready.load() < nthreads) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long r1, b1, l1;
    fh_batching_stats(&r1, &b1, &l1);
    std::vector<float> all;
    for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    const double blocks = (double)nthreads * nblocks;
    int ok = 1;
    for (auto* p : procs) ok &= fh_processor_ok(p);
    if (argc > 5 && std::string(argv[5]) == "json") {
        printf("{\"threads\": %d, \"combiner\": %s, \"blocks_per_s\": %.0f, \"msamples_per_s\": %.1f, \"block_latency_us_median\": %.1f, "
               "\"block_latency_us_p99\": %.1f, \"engine_calls\": %lld, \"largest_batch\": %lld, \"ok\": %s}\n",
               nthreads, batching ? "true" : "false", blocks / dt, blocks * P * cout / dt / 1e6, all[all.size() / 2], all[all.size() * 99 / 100],
               b1 - b0, l1, ok ? "true" : "false");
        for (auto* p : procs) fh_processor_destroy(p);
        return ok ? 0 : 1;
    }
    printf("threads %3d batching %d: %9.0f blocks/s = %7.1f Msamples/s (%d ch), block latency median %6.1f us, p99 %7.1f us; "
           "combiner: %lld calls in %lld launches (largest %lld); ok %d\n",
           nthreads, batching, blocks / dt, blocks * P * cout / dt / 1e6, cout, all[all.size() / 2], all[all.size() * 99 / 100],
           r1 - r0, b1 - b0, l1, ok);
    // with the TRACE build of the library (make -C folve_amd/csrc TRACE=1; LD_LIBRARY_PATH / rpath to it): where the
    // host's time between two engine calls goes
    typedef int (*host_times_fn)(unsigned long long*, int);
    if (host_times_fn ht = (host_times_fn)dlsym(RTLD_DEFAULT, "fe_debug_host_times")) {
        unsigned long long v[8] = {};
        ht(v, 0);
        const double n = v[2] ? (double)v[2] : 1.0;
        printf("   engine calls %llu: entry -> launches enqueued %.1f us, -> completion seen %.1f us, completion -> next engine call entered %.1f us\n",
               v[2], v[0] / n / 1e3, v[1] / n / 1e3, v[4] / (n > 1 ? n - 1 : 1) / 1e3);
    }
    for (auto* p : procs) fh_processor_destroy(p);
    return ok ? 0 : 1;
}
