// The drop-in call pattern under load: N "file" threads, each with its own folve::SoundProcessor (through the C view of
// the host classes, include/folve_host.h), each pulling 8192-frame blocks the way ConvolveFileHandler does
// (/root/reference/convolve-file-handler.cc:335-347,370-424): FillBuffer -> WriteProcessed (which runs Process()).
// A thread's "file" is `blocks` blocks long; its source is a callback with sf_readf_float's contract over a cyclic
// buffer of seeded noise, its sink a callback with sf_writef_float's that copies the block out — so that a processor
// with run-ahead on can ask for many blocks at once, exactly as it would ask libsndfile.
// Prints blocks per second over all threads and per GPU, the latency of a block as a thread sees it, and what the
// per-GPU combiner made of the calls.
//   usage: dropin_threads <filter.conf> <threads> <blocks per thread> <batching 0|1> [json] [run_ahead=N] [file_blocks=N] [pin=0|1]
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "folve_host.h"

namespace {
struct File {                      // what a thread reads from and writes to
    std::vector<float> data;       // cyclic source, `cycle` frames
    std::vector<float> sink;       // one block
    std::vector<float>* keep = nullptr;   // verify=1: everything written, in order
    long long left = 0;            // frames until end of file
    size_t pos = 0;                // frame position inside the cycle
    size_t cycle = 0;
    int cin = 0, cout = 0;
};
int read_cb(void* user, float* dst, int frames) {
    File* f = static_cast<File*>(user);
    const int n = (int)std::min<long long>(frames, f->left);
    int done = 0;
    while (done < n) {
        const int run = (int)std::min<size_t>((size_t)(n - done), f->cycle - f->pos);
        memcpy(dst + (size_t)done * f->cin, f->data.data() + f->pos * f->cin, sizeof(float) * (size_t)run * f->cin);
        f->pos = (f->pos + (size_t)run) % f->cycle;
        done += run;
    }
    f->left -= n;
    return n;
}
int write_cb(void* user, const float* src, int frames) {
    File* f = static_cast<File*>(user);
    memcpy(f->sink.data(), src, sizeof(float) * (size_t)frames * f->cout);
    if (f->keep) f->keep->insert(f->keep->end(), src, src + (size_t)frames * f->cout);
    return frames;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s conf threads blocks batching [json] [run_ahead=N] [file_blocks=N] [pin=0|1]\n", argv[0]); return 2; }
    const char* conf = argv[1];
    const int nthreads = atoi(argv[2]), nblocks = atoi(argv[3]), batching = atoi(argv[4]);
    bool json = false;
    int verify = 0;                // 1: keep every thread's output and compare it with the same file pulled one block per call by one thread
    int run_ahead = 1, file_blocks = 64, pin = 0, tune_knob = -1, tune_value = 0;   // tune=K:V: fe_engine_set_tuning on every engine (experiments)
    for (int i = 5; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "json") json = true;
        else if (a.rfind("run_ahead=", 0) == 0) run_ahead = atoi(a.c_str() + 10);
        else if (a.rfind("file_blocks=", 0) == 0) file_blocks = atoi(a.c_str() + 12);
        else if (a.rfind("pin=", 0) == 0) pin = atoi(a.c_str() + 4);
        else if (a.rfind("peaks=", 0) == 0) fh_device_peaks_set(atoi(a.c_str() + 6));
        else if (a.rfind("verify=", 0) == 0) verify = atoi(a.c_str() + 7);
        else if (a.rfind("early=", 0) == 0) fh_batching_early_quarters(atoi(a.c_str() + 6));
        else if (a.rfind("tune=", 0) == 0) { tune_knob = atoi(a.c_str() + 5); tune_value = atoi(strchr(a.c_str(), ':') ? strchr(a.c_str(), ':') + 1 : "0"); }
    }
    fh_batching_set(batching, 0, 256);
    fh_run_ahead_set(run_ahead);
    fh_numa_placement_set(pin);
    std::vector<fh_processor*> procs;
    for (int i = 0; i < nthreads; ++i) {
        fh_processor* p = fh_processor_create(conf, 44100, 2);
        if (!p) { fprintf(stderr, "processor %d: creation failed\n", i); return 1; }
        procs.push_back(p);
    }
    if (tune_knob >= 0)
        for (auto* p : procs) fe_engine_set_tuning(fh_processor_engine(p), tune_knob, tune_value);
    const int P = fh_processor_block_size(procs[0]);
    const int cin = fh_processor_input_channels(procs[0]), cout = fh_processor_output_channels(procs[0]);
    const int warm = std::max(8, 3 * run_ahead);                 // past the run-ahead ramp, clocks up
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::vector<float>> lat((size_t)nthreads);
    std::vector<std::vector<float>> kept((size_t)nthreads);
    std::vector<float> peak_seen((size_t)nthreads, 0.f);
    std::vector<std::thread> th;
    long long r0, k0, b0, l0, o0;
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&, t] {
            fh_processor* p = procs[(size_t)t];
            if (pin) fh_pin_thread_near_device(fh_processor_device(p));
            std::mt19937 rng(100 + t);
            std::uniform_real_distribution<float> u(-1.f, 1.f);
            File f;
            f.cin = cin; f.cout = cout;
            f.cycle = (size_t)std::max(1, std::min(file_blocks, nblocks)) * P;
            f.data.resize(f.cycle * cin);
            f.sink.resize((size_t)P * cout);
            for (auto& v : f.data) v = u(rng);
            auto pull = [&](int blocks, std::vector<float>* lats) {       // AddMoreSoundData until the file ends
                f.left = (long long)blocks * P;
                long long todo = f.left;
                while (todo > 0) {
                    const auto a = std::chrono::steady_clock::now();
                    const int r = fh_processor_fill_buffer_from(p, read_cb, &f);
                    if (r <= 0) break;
                    fh_processor_write_processed_to(p, write_cb, &f, r);
                    todo -= r;
                    if (lats) lats->push_back(std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - a).count());
                }
            };
            pull(warm, nullptr);
            if (verify) {                                         // the timed file starts from silence, from the start of the cycle
                fh_processor_reset(p);
                f.pos = 0;
                f.keep = &kept[(size_t)t];
            }
            lat[(size_t)t].reserve((size_t)nblocks);
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            pull(nblocks, &lat[(size_t)t]);
            peak_seen[(size_t)t] = fh_processor_max_output_value(p);
        });
    }
    while (ready.load() < nthreads) std::this_thread::yield();
    fh_batching_stats2(&r0, &k0, &b0, &l0, &o0);
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long r1, k1, b1, l1, o1;
    fh_batching_stats2(&r1, &k1, &b1, &l1, &o1);
    std::vector<float> all;
    for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    const double blocks = (double)nthreads * nblocks;
    int ok = 1;
    double worst_rms = 0.0;
    if (verify) {
        // the same files once more, one after the other, one block per engine call, no combiner: what the reference's call
        // pattern computes on this engine; the run above must agree within float32 rounding (other kernel forms), peaks too
        fh_run_ahead_set(1);
        fh_batching_set(0, 0, -1);
        fh_processor* q = fh_processor_create(conf, 44100, 2);
        for (int t = 0; t < nthreads && q; ++t) {
            std::mt19937 rng(100 + t);
            std::uniform_real_distribution<float> u(-1.f, 1.f);
            File f;
            f.cin = cin; f.cout = cout;
            f.cycle = (size_t)std::max(1, std::min(file_blocks, nblocks)) * P;
            f.data.resize(f.cycle * cin);
            f.sink.resize((size_t)P * cout);
            for (auto& v : f.data) v = u(rng);
            std::vector<float> ref;
            f.keep = &ref;
            f.left = (long long)nblocks * P;
            fh_processor_reset(q);
            for (long long todo = f.left; todo > 0;) {
                const int r = fh_processor_fill_buffer_from(q, read_cb, &f);
                if (r <= 0) break;
                fh_processor_write_processed_to(q, write_cb, &f, r);
                todo -= r;
            }
            const std::vector<float>& got = kept[(size_t)t];
            double e = 0.0;
            if (got.size() != ref.size()) { e = 1e9; }
            else { for (size_t i = 0; i < ref.size(); ++i) { const double dlt = (double)got[i] - ref[i]; e += dlt * dlt; } e = std::sqrt(e / (double)std::max<size_t>(1, ref.size())); }
            if (std::fabs(peak_seen[(size_t)t] - fh_processor_max_output_value(q)) > 1e-5) e = std::max(e, 1.0);
            worst_rms = std::max(worst_rms, e);
        }
        if (q) fh_processor_destroy(q);
        if (!(worst_rms <= 2e-6)) ok = 0;
    }
    std::map<int, int> per_gpu;
    for (auto* p : procs) { ok &= fh_processor_ok(p); per_gpu[fh_processor_device(p)]++; }
    std::string gpus = "{";
    for (auto& kv : per_gpu) {
        char b[128];
        snprintf(b, sizeof(b), "%s\"%d\": {\"streams\": %d, \"blocks_per_s\": %.0f}", gpus.size() > 1 ? ", " : "", kv.first, kv.second,
                 (double)kv.second * nblocks / dt);
        gpus += b;
    }
    gpus += "}";
    std::string slots = "[";                                  // streams per router slot while every file is open (one slot per GPU, or as FOLVE_AMD_DEVICES says)
    for (int k = 0; k < fh_router_device_count(); ++k) slots += (k ? ", " : "") + std::to_string(fh_router_live_streams(k));
    slots += "]";
    char rms_txt[32];
    snprintf(rms_txt, sizeof(rms_txt), "%.3g", worst_rms);
    if (json) {
        printf("{\"threads\": %d, \"combiner\": %s, \"run_ahead\": %d, \"blocks_per_s\": %.0f, \"msamples_per_s\": %.1f, \"block_latency_us_median\": %.1f, "
               "\"block_latency_us_p99\": %.1f, \"requests\": %lld, \"engine_calls\": %lld, \"largest_batch_blocks\": %lld, \"overlapped_batches\": %lld, "
               "\"gpus\": %s, \"router_slots\": %s, \"numa_pin\": %s, \"verified_rms\": %s, \"ok\": %s}\n",
               nthreads, batching ? "true" : "false", run_ahead, blocks / dt, blocks * P * cout / dt / 1e6, all[all.size() / 2], all[all.size() * 99 / 100],
               r1 - r0, b1 - b0, l1, o1 - o0, gpus.c_str(), slots.c_str(), pin ? "true" : "false", verify ? rms_txt : "null", ok ? "true" : "false");
        for (auto* p : procs) fh_processor_destroy(p);
        return ok ? 0 : 1;
    }
    printf("threads %3d batching %d run-ahead %3d: %9.0f blocks/s = %7.1f Msamples/s (%d ch), block latency median %6.1f us, p99 %7.1f us; "
           "combiner: %lld requests (%lld blocks) in %lld launches (largest %lld blocks, %lld overlapped); gpus %s; ok %d\n",
           nthreads, batching, run_ahead, blocks / dt, blocks * P * cout / dt / 1e6, cout, all[all.size() / 2], all[all.size() * 99 / 100],
           r1 - r0, k1 - k0, b1 - b0, l1, o1 - o0, gpus.c_str(), ok);
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
