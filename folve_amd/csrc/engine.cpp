// engine.cpp — host side of the gfx950 convolution engine behind include/folve_engine.h.
//
// Owns device memory and launch order; all arithmetic is in kernels/kernels.hip.
// There is deliberately no CPU compute path here: every failure to reach the
// GPU surfaces as FE_ERR_DEVICE.
#include "../../include/folve_engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <time.h>
#include <new>
#include <string>
#include <vector>

#include "kernels/kernels.h"
#include "trace.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(FE_ERR_DEVICE, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kJobSlots = 64;    // (rounds whose descriptors may be in flight at once: a pipeline of more chunks than slots blocks its caller)
constexpr int kSlabJobs = 128;   // descriptors a slot holds without an allocation of its own (a combined batch of 128 streams)

// Frees a temporary device allocation on every exit path.
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
};

}  // namespace

struct TableOffsets { int o[4]; int total; };

constexpr int kMaxChunks = 32;    // (how many a batch gets: run_pipelined.  Until round 5 this was 8 and "16: launches too small" — what made 16
                                  // chunks slow was the eight descriptor slots: the ninth round in flight blocked its caller)
constexpr int kLanes = 2;

// A launch lane: a HIP stream with its own K2 scratch.  Synchronous and device-pointer calls run on lane 0
// (the engine's stream); fe_batch_submit puts a batch on the lane with the least outstanding work, so that
// two submitted batches of DIFFERENT streams overlap on the GPU — K1 of one reads its PCM over the bus
// while K3 of the other writes its results back (PCIe is full duplex; one lane uses one direction at a time).
// A stream's consecutive calls must still execute in order: every stream remembers the lane and the
// sequence number of its last call, and a call on the other lane first waits (on the device) for that
// lane unless the earlier call is already known to have completed.
struct Lane {
    hipStream_t st = nullptr;
    float2* Y = nullptr;                 // batch scratch: accumulated spectra of one launch round
    size_t Y_bytes = 0;
    hipEvent_t xev = nullptr;            // "everything submitted to this lane so far", for cross-lane ordering
    long long submitted = 0;             // calls enqueued on this lane
    long long done = 0;                  // ... of which known to have completed (a lane executes in order)
    int outstanding = 0;                 // tickets not yet waited for
};

struct fe_engine {
    std::atomic<int> refs{1};     // creator + one per live filter (streams hold their filter)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;
    std::map<int, float2*> tw;              // log2P -> [exp(-2 pi i k / 2P), k < 2P | stage A | stage B tables]
    std::map<int, TableOffsets> tw_off;
    Lane lanes[kLanes];                  // lanes[0].st == stream
    int lane_toggle = 0;
    fk::Tuning tuning;                   // launch-shape overrides (fe_engine_set_tuning)
    bool host_io = false;                // the call in progress runs zero-copy on page-locked host buffers
    bool in_resident = false;            // ... its input has been staged into device memory by DMA (run_duplex)
    float* dx_stage[2] = {};             // run_duplex: input staging, alternating between consecutive batches
    size_t dx_stage_bytes[2] = {};
    float* dx_stage_out[2] = {};         // ... and output staging when the results leave by DMA too (FE_TUNE_DUPLEX_OUT = 2)
    size_t dx_stage_out_bytes[2] = {};
    int dx_stage_idle[2] = {}, dx_stage_out_idle[2] = {};   // consecutive batches that needed less than a quarter of the staging
    int dx_quiet = 0;                    // consecutive batches that did not take the duplex pipeline at all (its staging is let go after 256)
    hipEvent_t dx_k3[16] = {};           // "K3 of chunk c has finished"
    // per-block maxima of submitted batches (fe_batch_submit_peaks): a few rotating device / page-locked pairs
    struct PeakBuf {
        unsigned int* dev = nullptr; unsigned int* host = nullptr; size_t cap = 0;    // blocks
        bool in_use = false;             // a ticket has not handed its maxima out yet
    };
    PeakBuf pkb[4];
    int pkb_next = 0;
    int duplex_chunk_mb = 0;             // FE_TUNE_DUPLEX_CHUNK_MB (0: 32)
    int duplex_cap_mb = 0;               // FE_TUNE_DUPLEX_CAP_MB (0: 8192): a batch that needs more staging per direction keeps the zero-copy kernels
    int duplex_min_mb = 0;               // FE_TUNE_DUPLEX_MIN_MB (0: 32): smaller submitted batches keep the zero-copy kernels
    int duplex_out = 0;                  // 0 / 2: K3 -> device staging -> DMA out; 1: K3 stores into the callers' buffers
    hipEvent_t dx_free[2] = {};          // the batch that last used dx_stage[i] has finished
    bool dx_free_pending[2] = {};
    int dx_parity = 0;
    // rotating pinned/device buffers for job descriptors (async uploads)
    fk::StreamJob* jobs_host[kJobSlots] = {};
    fk::StreamJob* jobs_dev[kJobSlots] = {};
    size_t jobs_cap[kJobSlots] = {};
    bool jobs_own[kJobSlots] = {};       // the slot outgrew its share of the slabs and has allocations of its own
    fk::StreamJob* jobs_slab_host = nullptr;   // kJobSlots x kSlabJobs descriptors, page-locked / on the device: allocated with the engine,
    fk::StreamJob* jobs_slab_dev = nullptr;    // so that no launch round of a warm-up pays an allocation (hipMalloc can synchronise the device)
    hipEvent_t jobs_ev[kJobSlots] = {};
    bool jobs_ev_pending[kJobSlots] = {};
    int jobs_next = 0;
    // staging for host-pointer calls
    float* stage_in = nullptr;
    size_t stage_in_bytes = 0;
    float* stage_out = nullptr;
    size_t stage_out_bytes = 0;
    // Large host-pointer batches are cut into chunks of whole streams and pipelined over three
    // HIP streams: chunk c+1 rides the bus in while chunk c computes and chunk c-1 rides out
    // (PCIe is full duplex; a serial H2D - compute - D2H uses one direction at a time).
    hipStream_t cp_in = nullptr, cp_out = nullptr;
    hipEvent_t ev_in[kMaxChunks] = {}, ev_k[kMaxChunks] = {}, ev_fork = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> ticket_events;   // idle completion events of fe_batch_submit tickets
    hipEvent_t dx_ev[16] = {};               // run_duplex: "K1 of chunk c has finished"
    // profiling
    hipEvent_t split_ev[4] = {};         // a lone stream's time tiles on two lanes: "K1 of tile c has finished"
    int split_tiles = 0;                 // FE_TUNE_SPLIT: 0 / 1 never, 2 .. 8 time tiles for a lone stream's long call (one launch round only)
    int fail_round_in = 0;               // test hook: the n-th launch round from now fails with FE_ERR_DEVICE (0: none, < 0: every round)
    int sync_in_flight = 0;              // synchronous zero-copy calls waiting (lock released) on lane 0
    bool tuning_single_lane = false;     // FE_TUNE_LANES = 1: every submitted batch on lane 0 (measurements)
    int profiling = 0;                   // fe_engine_set_profiling: 0 off, 1 events recorded between the launches, 2 events bound to the dispatches
    hipEvent_t pev[4] = {};
    // mode 2: start / stop of each role's dispatch (hipExtLaunchKernelGGL), a ring of sets read back when a set comes round
    // again or when the profile is asked for — the profiled rounds run back to back like the timed ones
    static constexpr int kKevSets = 32;
    hipEvent_t kev[kKevSets][FE_K_COUNT][2] = {};
    bool kev_pending[kKevSets] = {};
    int kev_next = 0;
    long long prof_launches[FE_K_COUNT] = {};
    double prof_ms[FE_K_COUNT] = {};
    long long prof_kernel_launches[FE_K_COUNT] = {};
    double prof_kernel_ms[FE_K_COUNT] = {};
    fk::LaunchNames last_names = {};     // the kernels of the most recent launch round (fe_engine_last_kernels)
};

struct fe_ticket {                  // a submitted batch whose outputs are not yet known to be in the caller's buffers
    fe_engine* e = nullptr;          // (holds a reference)
    hipEvent_t ev = nullptr;
    int lane = 0;                    // the lane its completion event is recorded on
    long long seq[2] = {0, 0};       // per lane: the sequence number of this batch there (0: lane not used)
    // per-block maxima to hand out when the batch is done: (destination, first block in peaks_host, blocks)
    const unsigned int* peaks_host = nullptr;
    int peaks_slot = -1;
    struct PeakDst { float* dst; size_t first; size_t blocks; };
    std::vector<PeakDst> peaks_dst;
};

struct PathHost {
    bool used = false;
    int link = -1;                 // index of the path whose data this one shares
    std::vector<float> taps;       // K*P when populated
    uint32_t mask[4] = {0, 0, 0, 0};
};

struct fe_filter {
    fe_engine* eng = nullptr;
    std::atomic<int> refs{1};
    int ninp = 0, nout = 0, size = 0, P = 0, log2P = 0, K = 0;
    float density = 0.f;
    bool committed = false;
    std::vector<PathHost> paths;   // [inp * nout + out]
    // device side (after commit)
    int ndata = 0;
    float2* H = nullptr;
    uint64_t* mask_dev = nullptr;
    fk::PathEntry* paths_dev = nullptr;
    int* out_first_dev = nullptr;
    fk::FilterDev dev{};
    fk::MacShape mac_shape{};
};

struct fe_stream {
    fe_filter* f = nullptr;
    fe_engine* eng = nullptr;
    int max_blocks = 1, ring = 1;
    float2* fdl = nullptr;
    size_t fdl_bytes = 0;
    unsigned int* peaks = nullptr; // [2]
    long long blocks_done = 0;
    int slot0 = 0;
    int last_lane = -1;              // lane and sequence number of the last call that touched this stream's state
    long long last_seq = 0;
    // page-locked caller memory bound to this stream (fe_stream_bind_host_buffer): calls whose
    // buffers lie inside it are read and written by the kernels directly, over the bus
    const char* bound_host = nullptr;
    char* bound_dev = nullptr;
    size_t bound_bytes = 0;
};

namespace {

int resolve(const fe_filter* f, int idx) {
    int guard = 0;
    while (f->paths[idx].link >= 0 && guard++ < 8192) idx = f->paths[idx].link;
    return idx;
}

int get_twiddles(fe_engine* e, int log2P, fk::FftTables* out) {
    auto it = e->tw.find(log2P);
    if (it == e->tw.end()) {
        const int n = fk::fft_table_count(log2P);
        if (n <= 0) return fail(FE_ERR_PARAM, "unsupported block size 2^%d", log2P);
        std::vector<float2> host((size_t)n);
        TableOffsets off{};
        fk::fill_fft_tables(log2P, host.data(), off.o);
        off.total = n;
        float2* dev = nullptr;
        HIP_TRY(hipMalloc(&dev, sizeof(float2) * host.size()));
        HIP_TRY(hipMemcpy(dev, host.data(), sizeof(float2) * host.size(), hipMemcpyHostToDevice));
        e->tw[log2P] = dev;
        e->tw_off[log2P] = off;
        it = e->tw.find(log2P);
    }
    const TableOffsets off = e->tw_off[log2P];
    out->tw = it->second;
    out->twa = it->second + off.o[0];
    out->twb = it->second + off.o[1];
    const bool has2 = off.o[2] < off.total;           // the 2P-point (stereo) geometry exists for 512 <= P <= 4096
    out->twa2 = has2 ? it->second + off.o[2] : nullptr;
    out->twb2 = has2 ? it->second + off.o[3] : nullptr;
    return FE_OK;
}

int ensure_bytes(fe_engine* e, void** ptr, size_t* have, size_t need, hipStream_t user = nullptr) {
    if (*have >= need) return FE_OK;
    HIP_TRY(hipStreamSynchronize(user ? user : e->stream));   // nothing in flight may still use the old buffer
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *have = 0;
    size_t cap = need + need / 4;
    HIP_TRY(hipMalloc(ptr, cap));
    *have = cap;
    return FE_OK;
}

struct Item {
    fe_stream* s;
    const float* in;     // device
    float* out;          // device
    long long left;
    unsigned int* blk_peaks = nullptr;   // device: this stream's per-block maxima of the call (advances with the rounds)
};

// Profiling mode 2: read one set of dispatch-bound events back — each dispatch's own begin-to-end (the command
// processor's stamps of the packet: no launch boundary inside) — waiting for its round if it is still running.
int harvest_kernel_events(fe_engine* e, int set) {
    if (!e->kev_pending[set]) return FE_OK;
    e->kev_pending[set] = false;
    HIP_TRY(hipEventSynchronize(e->kev[set][FE_K_COUNT - 1][1]));
    for (int k = 0; k < FE_K_COUNT; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->kev[set][k][0], e->kev[set][k][1]));
        e->prof_kernel_ms[k] += ms;
        e->prof_kernel_launches[k] += 1;
    }
    return FE_OK;
}

// One launch round over streams that share a filter.  Host-side stream state (ring position, block
// count) advances only after all three launches were accepted: a failed round leaves every stream
// where it was.
//   after_k1 / before_k2: an event to record behind K1 / to wait for in front of K2 (a lone stream's time tiles on two
//   lanes: tile c's K2 reads the spectra tile c - 1's K1 writes on the other lane); limit_blocks: at most so many blocks
//   of every stream in this round (0: the stream's own bound).
int launch_round(fe_engine* e, fe_filter* f, std::vector<Item>& items, bool* any, int lane, hipEvent_t after_k1 = nullptr,
                 hipEvent_t before_k2 = nullptr, int limit_blocks = 0) {
    Lane& L = e->lanes[lane];
    hipStream_t st = L.st;
    const int P = f->P;
    std::vector<fk::StreamJob> jobs;
    std::vector<Item*> owners;
    jobs.reserve(items.size());
    owners.reserve(items.size());
    int yunits = 0, max_blocks = 0, max_ring = 0;
    bool in_pairs_ok = true, out_pairs_ok = true, want_block_peaks = false;
    for (Item& it : items) {
        if (it.left <= 0) continue;
        fe_stream* s = it.s;
        const long long cap = (long long)(limit_blocks > 0 ? std::min(limit_blocks, s->max_blocks) : s->max_blocks) * P;
        const long long take = std::min(it.left, cap);
        fk::StreamJob j{};
        j.in = it.in;
        j.out = it.out;
        j.fdl = s->fdl;
        j.peaks = s->peaks;
        j.blk_peaks = it.blk_peaks;
        if (it.blk_peaks) want_block_peaks = true;
        j.nframes = take;
        if (reinterpret_cast<uintptr_t>(it.in) & 15) in_pairs_ok = false;
        if (reinterpret_cast<uintptr_t>(it.out) & 15) out_pairs_ok = false;
        j.nblocks = (int)((take + P - 1) / P);
        j.slot0 = s->slot0;
        j.yunit0 = yunits;
        j.ring = s->ring;
        max_ring = std::max(max_ring, s->ring);
        yunits += j.nblocks * f->nout;
        max_blocks = std::max(max_blocks, j.nblocks);
        jobs.push_back(j);
        owners.push_back(&it);
    }
    *any = !jobs.empty();
    if (jobs.empty()) return FE_OK;

    int rc = ensure_bytes(e, (void**)&L.Y, &L.Y_bytes, (size_t)yunits * P * sizeof(float2), st);
    if (rc) return rc;

    // A launch of ONE stream carries its descriptor among the kernel arguments (Tuning::one_job, copied at launch): nothing to
    // upload, nothing to fetch, no slot.  Several streams: the descriptors go through a rotating pinned buffer; a slot is
    // reused only after the round that read it has finished (its event is recorded behind that round's last kernel).
    // Small launches (the combined one-block calls of many file threads, whose PCM crosses the bus anyway) read their
    // descriptors straight from the page-locked buffer — no upload command in front of K1 (10 - 15 us of copy-engine
    // latency per round); large ones upload them once (thousands of workgroups should not each fetch a descriptor over
    // the bus).
    const int nj = (int)jobs.size();
    const fk::StreamJob* dj = nullptr;
    int slot = -1;
    if (nj > 1) {
        slot = e->jobs_next;
        e->jobs_next = (e->jobs_next + 1) % kJobSlots;
        if (e->jobs_ev_pending[slot]) {
            HIP_TRY(hipEventSynchronize(e->jobs_ev[slot]));
            e->jobs_ev_pending[slot] = false;
        }
        const size_t bytes = jobs.size() * sizeof(fk::StreamJob);
        if (e->jobs_cap[slot] < bytes) {                        // more streams than a slot's share of the slabs holds
            if (e->jobs_own[slot]) {
                if (e->jobs_host[slot]) HIP_TRY(hipHostFree(e->jobs_host[slot]));
                if (e->jobs_dev[slot]) HIP_TRY(hipFree(e->jobs_dev[slot]));    // (its last reader has finished: the slot's event, above)
            }
            e->jobs_host[slot] = nullptr; e->jobs_dev[slot] = nullptr; e->jobs_cap[slot] = 0;
            e->jobs_own[slot] = true;
            const size_t cap = bytes * 2;
            HIP_TRY(hipHostMalloc((void**)&e->jobs_host[slot], cap, hipHostMallocDefault));
            HIP_TRY(hipMalloc((void**)&e->jobs_dev[slot], cap));
            e->jobs_cap[slot] = cap;
        }
        memcpy(e->jobs_host[slot], jobs.data(), bytes);
        if ((long long)nj * max_blocks <= (e->host_io ? 256 : 16)) {
            dj = e->jobs_host[slot];
        } else {
            dj = e->jobs_dev[slot];
            HIP_TRY(hipMemcpyAsync(e->jobs_dev[slot], e->jobs_host[slot], bytes, hipMemcpyHostToDevice, st));
        }
    }

    const bool prof = e->profiling == 1, kprof = e->profiling == 2;
    if (e->fail_round_in < 0 || (e->fail_round_in > 0 && --e->fail_round_in == 0)) {   // test hook (fe_engine_set_tuning FE_TUNE_FAIL_NEXT): an injected device failure
        return fail(FE_ERR_DEVICE, "injected device failure (test hook)");
    }
    fk::Tuning tn = e->tuning;
    tn.host_io = e->host_io;
    tn.in_resident = e->in_resident;
    tn.max_ring = max_ring;
    if (want_block_peaks) tn.inv_run = 1;      // K3's walker: one block per workgroup, whose maxima are the block's
    tn.one_job = nj == 1 ? &jobs[0] : nullptr;
    tn.names = &e->last_names;
    int kset = -1;
    if (kprof) {
        kset = e->kev_next;
        e->kev_next = (e->kev_next + 1) % fe_engine::kKevSets;
        int hrc = harvest_kernel_events(e, kset);
        if (hrc) return hrc;
        tn.kev = e->kev[kset];
    }
    // (roctx, when a profiler listens: the three launches of this round under one named range)
    ftrace::Range round_range("folve round: filter %p (%d->%d ch, K=%d, P=%d) streams=%d blocks<=%d lane=%d", static_cast<void*>(f), f->ninp,
                              f->nout, f->K, f->P, nj, max_blocks, lane);
    if (prof) HIP_TRY(hipEventRecord(e->pev[0], st));
    HIP_TRY(fk::launch_forward(dj, nj, max_blocks, f->dev, in_pairs_ok, tn, st));
    if (after_k1) HIP_TRY(hipEventRecord(after_k1, st));
    if (before_k2) HIP_TRY(hipStreamWaitEvent(st, before_k2, 0));
    if (prof) HIP_TRY(hipEventRecord(e->pev[1], st));
    HIP_TRY(fk::launch_mac(dj, nj, max_blocks, f->dev, L.Y, max_blocks, f->mac_shape, tn, st));
    if (prof) HIP_TRY(hipEventRecord(e->pev[2], st));
    HIP_TRY(fk::launch_inverse(dj, nj, max_blocks, f->dev, L.Y, out_pairs_ok, tn, st));
    // (a one-stream round's descriptor went by value: nothing on the device reads the slot, and a marker behind K3 is 3 us in
    // front of the next call's K1 — cfg1 61.9 -> 58.7 us per call, cfg2 54.0 -> 51.5, cfg4 220.5 -> 217.3)
    if (nj > 1) {
        HIP_TRY(hipEventRecord(e->jobs_ev[slot], st));
        e->jobs_ev_pending[slot] = true;
    }
    if (prof) {
        HIP_TRY(hipEventRecord(e->pev[3], st));
        HIP_TRY(hipEventSynchronize(e->pev[3]));
        for (int k = 0; k < FE_K_COUNT; ++k) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e->pev[k], e->pev[k + 1]));
            e->prof_ms[k] += ms;
            e->prof_launches[k] += 1;
        }
    }
    if (kprof) e->kev_pending[kset] = true;
    // everything is enqueued: advance the streams (all later work is stream-ordered behind it)
    for (size_t i = 0; i < jobs.size(); ++i) {
        Item& it = *owners[i];
        fe_stream* s = it.s;
        const fk::StreamJob& j = jobs[i];
        s->slot0 = (s->slot0 + j.nblocks) % s->ring;
        s->blocks_done += j.nblocks;
        it.in += (size_t)j.nframes * f->ninp;
        it.out += (size_t)j.nframes * f->nout;
        if (it.blk_peaks) it.blk_peaks += 2 * (size_t)j.nblocks;
        it.left -= j.nframes;
    }
    return FE_OK;
}

// Launch rounds for streams [i0, i1) of a call, grouped by filter; each group runs until its
// frames are consumed.
//
// spread: a call that holds SEVERAL filters' streams (the reference resolves a configuration per sampling rate,
// channel count and sample width, processor-pool.cc:53-61: a music library keeps a handful of filters live) puts its
// groups on BOTH launch lanes — each group's K1 -> K2 -> K3 chain on one lane, the groups dealt to the lane with less
// work so far — so that one group's dependency gaps and tails are filled by the other's kernels instead of the groups
// queueing one behind the other (64 streams over 4 filters, 64 blocks each: 0.756 ms one after the other against 0.647 ms
// for 64 streams of one filter).  The other lane starts behind everything `lane` holds at this point (the caller's
// earlier work, the cross-lane waits of this call) and `lane` ends behind the other lane's groups, so to the rest of
// the engine the whole call still lives on `lane`.  Results are the same bits: a group's launches do not depend on
// the lane.
int run_groups(fe_engine* e, fe_stream* const* streams, std::vector<Item>& all, int i0, int i1, int lane = 0,
               hipEvent_t after_k1 = nullptr, bool spread = false) {
    std::vector<char> done((size_t)(i1 - i0), 0);
    struct Group { fe_filter* f; std::vector<Item> items; long long work; };
    std::vector<Group> groups;
    for (int i = i0; i < i1; ++i) {
        if (done[(size_t)(i - i0)]) continue;
        Group g{streams[i]->f, {}, 0};
        for (int k = i; k < i1; ++k)
            if (!done[(size_t)(k - i0)] && streams[k]->f == g.f) {
                g.items.push_back(all[(size_t)k]);
                done[(size_t)(k - i0)] = 1;
                g.work += all[(size_t)k].left * (g.f->ninp + g.f->nout) * (long long)(g.f->K + 8);
            }
        groups.push_back(std::move(g));
    }
    const int other = lane ^ 1;
    const bool lanes_ok = spread && !after_k1 && !e->profiling && !e->tuning_single_lane && e->sync_in_flight == 0;
    bool two = lanes_ok && groups.size() >= 2;
    // A LONE stream's long call (one open file converting far ahead; cfg2, cfg4): its three kernels are too small to fill
    // the chip and a good part of the call is the launch chain itself — two dependency gaps, three launch ramps and
    // tails.  The call's blocks are cut into time tiles whose K1 -> K2 -> K3 chains alternate between the two lanes: tile
    // c's K2 needs the spectra of tile c - 1 (history), so it waits for that tile's K1 on the other lane, and nothing
    // else — while tile c computes its products tile c + 1 transforms its PCM and tile c - 1 its results.
    int split = 0;
    // (not for PCM that crosses the bus under the kernels: a K1 reading page-locked memory beside a K3 writing it takes
    // twice its time — reads queue behind posted writes)
    if (lanes_ok && groups.size() == 1 && groups[0].items.size() == 1 && e->split_tiles != 1 && !e->host_io) {
        const Item& it = groups[0].items[0];
        const int P = groups[0].f->P;
        const long long nb = (std::min<long long>(it.left, (long long)it.s->max_blocks * P) + P - 1) / P;
        if (e->split_tiles >= 2) split = (int)std::min<long long>(e->split_tiles, nb);
        // (automatic: never — measured on MI355X, cfg2 62 -> 88 us per 256-block call with two tiles, +14 .. 27 us per further
        // tile: a cross-lane event wait costs more than the gap it hides; the knob stays for measurements)
        if (split < 2) split = 0;
        // (a span longer than the stream's run-ahead depth takes several launch rounds: the first tile of a later round would
        // read history spectra that the previous round's last tile wrote on the OTHER lane with nothing ordering the two, and
        // its K1 could wrap the ring into slots a K2 over there still reads — such a call is not split)
        if (it.left > (long long)it.s->max_blocks * P) split = 0;
    }
    if (split) two = true;
    if (two) {
        if (!e->lanes[1].st) {
            HIP_TRY(hipStreamCreateWithFlags(&e->lanes[1].st, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&e->lanes[1].xev, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(e->lanes[lane].xev, e->lanes[lane].st));
        HIP_TRY(hipStreamWaitEvent(e->lanes[other].st, e->lanes[lane].xev, 0));
    }
    long long load[2] = {0, 0};
    bool enqueued[2] = {false, false};
    auto drain = [&] {
        // Kernels of earlier rounds of this call are already on the GPU and still write into the
        // callers' buffers: the error is reported only after they have drained, so that a caller
        // who is told "failed" owns its buffers again (best effort: the device may be gone).
        if (enqueued[0]) (void)hipStreamSynchronize(e->lanes[lane].st);
        if (enqueued[1]) (void)hipStreamSynchronize(e->lanes[other].st);
    };
    if (split) {
        if (!e->split_ev[0])
            for (hipEvent_t& ev : e->split_ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        Group& g = groups[0];
        const int P = g.f->P;
        bool any = true;
        while (any) {                                           // (a span longer than the stream's bound: round after round)
            const long long nb = (std::min<long long>(g.items[0].left, (long long)g.items[0].s->max_blocks * P) + P - 1) / P;
            if (nb <= 0) break;
            const int tiles = (int)std::min<long long>(split, nb);
            const int tlen = (int)((nb + tiles - 1) / tiles);
            for (int c = 0; c < tiles && any; ++c) {
                const int side = c & 1;
                hipEvent_t mine = e->split_ev[c % 4], prev = c ? e->split_ev[(c - 1) % 4] : nullptr;
                int rc = launch_round(e, g.f, g.items, &any, side ? other : lane, mine, prev, tlen);
                if (rc) { drain(); return rc; }
                if (any) enqueued[side] = true;
            }
        }
    } else {
        for (Group& g : groups) {
            const int side = two && load[1] < load[0] ? 1 : 0;  // 0: `lane`, 1: the other one
            load[side] += g.work;
            bool any = true;
            while (any) {
                int rc = launch_round(e, g.f, g.items, &any, side ? other : lane, after_k1);
                if (rc) { drain(); return rc; }
                enqueued[side] = true;
            }
        }
    }
    if (two && enqueued[1]) {
        hipError_t he = hipEventRecord(e->lanes[other].xev, e->lanes[other].st);
        if (he == hipSuccess) he = hipStreamWaitEvent(e->lanes[lane].st, e->lanes[other].xev, 0);
        if (he != hipSuccess) { drain(); return fail(FE_ERR_DEVICE, "joining the launch lanes: %s", hipGetErrorString(he)); }
    }
    return FE_OK;
}

// A host-pointer call as a three-stage pipeline over chunks of whole streams (see fe_engine).
// all[i].in / .out already point into the staging buffers.
int run_pipelined(fe_engine* e, fe_stream* const* streams, int n, const float* const* in, float* const* out,
                  const long long* nframes, std::vector<Item>& all) {
    if (!e->cp_out) {
        if (!e->cp_in) HIP_TRY(hipStreamCreateWithFlags(&e->cp_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&e->cp_out, hipStreamNonBlocking));
        for (int i = 0; i < kMaxChunks; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&e->ev_in[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&e->ev_k[i], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    // chunk boundaries: whole streams, about equal frame counts
    long long total = 0;
    for (int i = 0; i < n; ++i) total += nframes[i] * (streams[i]->f->ninp + streams[i]->f->nout);
    // Chunks of at least 64 MB of PCM (in + out), at most kMaxChunks and at most one per two streams: the pipeline's fill and
    // drain cost one chunk's copy each, so a big batch wants many (cfg3's 256-block batch, 2.1 GB: 8 chunks 25.2 - 28.3 ms,
    // 32 chunks 23.3 - 25.8), a small one few (its launches must not get too small).
    const int by_bytes = (int)std::min<long long>(kMaxChunks, total * (long long)sizeof(float) / ((long long)64 << 20));
    const int want = std::max(1, std::min(std::max(8, by_bytes), n / 2));
    int first[kMaxChunks + 1];
    int chunks = 0;
    long long acc = 0;
    first[0] = 0;
    for (int i = 0; i < n; ++i) {
        acc += nframes[i] * (streams[i]->f->ninp + streams[i]->f->nout);
        if (chunks + 1 < want && acc * want >= total * (chunks + 1) && i + 1 < n) first[++chunks] = i + 1;
    }
    first[++chunks] = n;

    std::vector<const float*> dev_out((size_t)n);
    for (int i = 0; i < n; ++i) dev_out[(size_t)i] = all[(size_t)i].out;   // launch rounds advance Item::out
    auto copy_in = [&](int c) -> int {
        for (int i = first[c]; i < first[c + 1]; ++i) {
            const size_t ni = (size_t)nframes[i] * streams[i]->f->ninp;
            if (ni) HIP_TRY(hipMemcpyAsync(const_cast<float*>(all[(size_t)i].in), in[i], ni * sizeof(float), hipMemcpyHostToDevice, e->cp_in));
        }
        HIP_TRY(hipEventRecord(e->ev_in[c], e->cp_in));
        return FE_OK;
    };
    auto copy_out = [&](int c) -> int {
        HIP_TRY(hipStreamWaitEvent(e->cp_out, e->ev_k[c], 0));
        for (int i = first[c]; i < first[c + 1]; ++i) {
            const size_t no = (size_t)nframes[i] * streams[i]->f->nout;
            if (no) HIP_TRY(hipMemcpyAsync(out[i], dev_out[(size_t)i], no * sizeof(float), hipMemcpyDeviceToHost, e->cp_out));
        }
        return FE_OK;
    };
    // the staging buffers may still be read by earlier work on the engine's stream
    HIP_TRY(hipEventRecord(e->ev_fork, e->stream));
    HIP_TRY(hipStreamWaitEvent(e->cp_in, e->ev_fork, 0));
    int rc = copy_in(0);
    if (rc) return rc;
    for (int c = 0; c < chunks; ++c) {
        HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_in[c], 0));
        rc = run_groups(e, streams, all, first[c], first[c + 1]);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e->ev_k[c], e->stream));
        if (c + 1 < chunks) { rc = copy_in(c + 1); if (rc) return rc; }
        rc = copy_out(c);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->ev_join, e->cp_out));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    return FE_OK;
}

// A large zero-copy call (the combined run-ahead chunks of many files: tens of megabytes of PCM each way) as a
// duplex pipeline.  Measured on MI355X (profiles/r03*_dropin_*): kernels READ page-locked host memory at 28 - 35 GB/s
// only (64-byte read requests, a bounded number in flight) and WRITE it at 41 GB/s while they hold the CUs; the DMA
// engines move 45 - 50 GB/s each way at once and leave the CUs alone (64 file threads: 6.0 Gsamples/s with kernel reads
// and writes, 8.2 with DMA in and kernel stores out, 9.0 - 9.5 with DMA both ways).  So the streams are cut into chunks;
// every chunk's PCM is copied into device memory by one copy stream (inbound), its K1/K2/K3 run from and to HBM on one
// of the two lanes, and its results are copied into the callers' buffers by another copy stream (outbound): chunk
// c + 1 rides in and chunk c - 1 rides out while chunk c computes — both directions of the bus are busy for the whole
// call, whatever the host threads' timing is.  The staging buffers alternate between consecutive batches, so the next
// batch's copies start while this one still computes.  (FE_TUNE_DUPLEX_OUT = 1: K3 stores straight into the callers'
// buffers instead of the outbound copies.)  At the end `lane` (the one the caller records its completion event on)
// waits for the other lane and for the last copy out.
constexpr int kDuplexChunks = 16;
constexpr size_t kDuplexStageCap = (size_t)8 << 30;      // per direction and parity: beyond it a batch keeps the zero-copy kernels
struct DuplexPlan {
    int nc = 0;
    int first[kDuplexChunks + 1] = {};
};
// chunk boundaries: whole streams, about equal byte counts; lane_of[i] = the lane stream i's chunk runs on
void plan_duplex(fe_stream* const* streams, int n, const long long* nframes, int lane, int chunks, DuplexPlan* p, int* lane_of) {
    long long total = 0;
    for (int i = 0; i < n; ++i) total += nframes[i] * (streams[i]->f->ninp + streams[i]->f->nout);
    long long acc = 0;
    p->nc = 0;
    p->first[0] = 0;
    for (int i = 0; i < n; ++i) {
        acc += nframes[i] * (streams[i]->f->ninp + streams[i]->f->nout);
        lane_of[i] = (lane + p->nc) & 1;
        if (p->nc + 1 < chunks && acc * chunks >= total * (p->nc + 1) && i + 1 < n) p->first[++p->nc] = i + 1;
    }
    p->first[++p->nc] = n;
}
// After a burst of big batches the traffic may consist of small ones only (single blocks, short run-ahead chunks): those
// never reach run_duplex, whose own shrink logic would therefore never run, and up to 1.25 x the cap per direction and
// parity would stay allocated for the life of the process.  Counted on the submit path; the staging goes once 256 batches
// in a row have passed it by and the batches that last used it have finished.
void release_idle_duplex_staging(fe_engine* e) {
    if (++e->dx_quiet < 256) return;
    e->dx_quiet = 0;
    for (int i = 0; i < 2; ++i) {
        if (!e->dx_stage[i] && !e->dx_stage_out[i]) continue;
        if (e->dx_free_pending[i]) {
            if (hipEventQuery(e->dx_free[i]) != hipSuccess) { (void)hipGetLastError(); continue; }   // still in use: next time
            e->dx_free_pending[i] = false;
        }
        if (e->dx_stage[i]) { (void)hipFree(e->dx_stage[i]); e->dx_stage[i] = nullptr; e->dx_stage_bytes[i] = 0; e->dx_stage_idle[i] = 0; }
        if (e->dx_stage_out[i]) { (void)hipFree(e->dx_stage_out[i]); e->dx_stage_out[i] = nullptr; e->dx_stage_out_bytes[i] = 0; e->dx_stage_out_idle[i] = 0; }
    }
}

int run_duplex(fe_engine* e, fe_stream* const* streams, std::vector<Item>& all, int n, const float* const* host_in,
               float* const* host_out, const long long* nframes, int lane, const DuplexPlan& p,
               unsigned int* pk_dev = nullptr, unsigned int* pk_host = nullptr, size_t pk_bytes = 0) {
    const bool dma_out = e->duplex_out != 1;
    if (!e->dx_ev[0]) {
        for (int i = 0; i < kDuplexChunks; ++i) HIP_TRY(hipEventCreateWithFlags(&e->dx_ev[i], hipEventDisableTiming));
        for (int i = 0; i < kDuplexChunks; ++i) HIP_TRY(hipEventCreateWithFlags(&e->dx_k3[i], hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&e->dx_free[i], hipEventDisableTiming));
    }
    if (!e->cp_in) HIP_TRY(hipStreamCreateWithFlags(&e->cp_in, hipStreamNonBlocking));
    if (dma_out && !e->cp_out) HIP_TRY(hipStreamCreateWithFlags(&e->cp_out, hipStreamNonBlocking));
    const int par = e->dx_parity;
    e->dx_parity ^= 1;
    // the batch before last used this staging buffer: it has long finished (at most two batches are in flight)
    if (e->dx_free_pending[par]) {
        HIP_TRY(hipEventSynchronize(e->dx_free[par]));
        e->dx_free_pending[par] = false;
    }
    size_t need = 0;
    std::vector<size_t> off((size_t)n);
    for (int i = 0; i < n; ++i) {
        off[(size_t)i] = need;
        need += (((size_t)nframes[i] * streams[i]->f->ninp + 3) & ~(size_t)3) * sizeof(float);
    }
    // Staging follows the batches: it grows to 1.25 x the largest batch and is let go again when batches have become much
    // smaller (a burst of 64 files x 64 stereo blocks leaves 0.6 GB per direction pair behind otherwise, for the life of the
    // process).  A batch the staging cannot be had for does not fail: FE_ERR_UNSUPPORTED sends it down the zero-copy path.
    auto fit_stage = [&](float*& buf, size_t& have, size_t want, int& idle) -> int {
        const bool too_small = have < want;
        // (let go only after 64 batches in a row that needed less than a quarter of it: a file's ramp 1, 2, 4 .. blocks
        // alternates small batches with big ones, and hipFree / hipMalloc synchronise the device)
        idle = (have > ((size_t)256 << 20) && have > 4 * want) ? idle + 1 : 0;
        const bool far_too_big = idle >= 64;
        if (!too_small && !far_too_big) return FE_OK;
        idle = 0;
        if (buf) HIP_TRY(hipFree(buf));
        buf = nullptr;
        have = 0;
        if (hipMalloc((void**)&buf, want + want / 4) != hipSuccess) {
            (void)hipGetLastError();
            buf = nullptr;
            return fail(FE_ERR_UNSUPPORTED, "no device memory for %zu bytes of duplex staging: the batch runs zero-copy", want + want / 4);
        }
        have = want + want / 4;
        return FE_OK;
    };
    const size_t stage_cap = e->duplex_cap_mb > 0 ? (size_t)e->duplex_cap_mb << 20 : kDuplexStageCap;
    if (need > stage_cap) return fail(FE_ERR_UNSUPPORTED, "batch of %zu bytes exceeds the duplex staging cap: it runs zero-copy", need);
    if (int rc = fit_stage(e->dx_stage[par], e->dx_stage_bytes[par], need, e->dx_stage_idle[par])) return rc;
    size_t need_out = 0;
    std::vector<size_t> off_out((size_t)n);
    if (dma_out) {
        for (int i = 0; i < n; ++i) {
            off_out[(size_t)i] = need_out;
            need_out += (((size_t)nframes[i] * streams[i]->f->nout + 3) & ~(size_t)3) * sizeof(float);
        }
        if (need_out > stage_cap) return fail(FE_ERR_UNSUPPORTED, "batch of %zu bytes exceeds the duplex staging cap: it runs zero-copy", need_out);
        if (int rc = fit_stage(e->dx_stage_out[par], e->dx_stage_out_bytes[par], need_out, e->dx_stage_out_idle[par])) {
            // the output side was refused: the batch runs zero-copy, and the input staging grown for it a moment ago is of no use
            if (e->dx_stage[par]) { (void)hipFree(e->dx_stage[par]); e->dx_stage[par] = nullptr; e->dx_stage_bytes[par] = 0; e->dx_stage_idle[par] = 0; }
            return rc;
        }
    }
    struct HostOutScope {                // with the results leaving by DMA the kernels see device memory on both sides
        fe_engine* e; bool was;
        HostOutScope(fe_engine* e_, bool off) : e(e_), was(e_->host_io) { if (off) e->host_io = false; }
        ~HostOutScope() { e->host_io = was; }
    } host_out_scope(e, dma_out);
    struct ResidentScope {
        fe_engine* e;
        explicit ResidentScope(fe_engine* e_) : e(e_) { e->in_resident = true; }
        ~ResidentScope() { e->in_resident = false; }
    } resident(e);
    const int nc = p.nc;
    const int* first = p.first;
    if (pk_dev) {
        // zeroed on the compute lanes, in front of every K3 (the copy streams stay pure DMA: a fill kernel between their
        // copies cost a quarter of the pipeline's rate)
        HIP_TRY(hipMemsetAsync(pk_dev, 0, pk_bytes, e->lanes[lane].st));
        HIP_TRY(hipEventRecord(e->lanes[lane].xev, e->lanes[lane].st));
        HIP_TRY(hipStreamWaitEvent(e->lanes[lane ^ 1].st, e->lanes[lane].xev, 0));
    }
    for (int c = 0; c < nc; ++c) {
        const int l = (lane + c) & 1;
        ftrace::Range chunk_range("folve duplex chunk %d/%d: streams %d..%d, copy in -> lane %d -> copy out", c + 1, nc, first[c], first[c + 1] - 1, l);
        for (int i = first[c]; i < first[c + 1]; ++i) {
            const size_t bytes = (size_t)nframes[i] * streams[i]->f->ninp * sizeof(float);
            float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(e->dx_stage[par]) + off[(size_t)i]);
            if (bytes) HIP_TRY(hipMemcpyAsync(dst, host_in[i], bytes, hipMemcpyHostToDevice, e->cp_in));
            all[(size_t)i].in = dst;
            if (dma_out) all[(size_t)i].out = reinterpret_cast<float*>(reinterpret_cast<char*>(e->dx_stage_out[par]) + off_out[(size_t)i]);
        }
        HIP_TRY(hipEventRecord(e->dx_ev[c], e->cp_in));
        HIP_TRY(hipStreamWaitEvent(e->lanes[l].st, e->dx_ev[c], 0));
        int rc = run_groups(e, streams, all, first[c], first[c + 1], l);
        if (rc) {
            // earlier chunks still write into the callers' buffers — on BOTH lanes (chunks c - 2, c - 4, .. ran on this
            // one, and a round that is refused before it enqueues anything does not drain its own lane)
            (void)hipStreamSynchronize(e->lanes[l].st);
            (void)hipStreamSynchronize(e->lanes[l ^ 1].st);
            (void)hipStreamSynchronize(e->cp_in);
            if (dma_out) (void)hipStreamSynchronize(e->cp_out);
            return rc;
        }
        if (dma_out) {
            HIP_TRY(hipEventRecord(e->dx_k3[c], e->lanes[l].st));
            HIP_TRY(hipStreamWaitEvent(e->cp_out, e->dx_k3[c], 0));
            for (int i = first[c]; i < first[c + 1]; ++i) {
                const size_t bytes = (size_t)nframes[i] * streams[i]->f->nout * sizeof(float);
                const char* src = reinterpret_cast<const char*>(e->dx_stage_out[par]) + off_out[(size_t)i];
                if (bytes) HIP_TRY(hipMemcpyAsync(host_out[i], src, bytes, hipMemcpyDeviceToHost, e->cp_out));
            }
        }
    }
    if (dma_out) {                       // the caller's completion event (recorded on `lane`) must follow the last copy out
        HIP_TRY(hipEventRecord(e->dx_k3[0], e->cp_out));
        HIP_TRY(hipStreamWaitEvent(e->lanes[lane].st, e->dx_k3[0], 0));
    }
    const int other = lane ^ 1;
    HIP_TRY(hipEventRecord(e->lanes[other].xev, e->lanes[other].st));
    HIP_TRY(hipStreamWaitEvent(e->lanes[lane].st, e->lanes[other].xev, 0));
    if (pk_dev) HIP_TRY(hipMemcpyAsync(pk_host, pk_dev, pk_bytes, hipMemcpyDeviceToHost, e->lanes[lane].st));   // behind every K3
    HIP_TRY(hipEventRecord(e->dx_free[par], e->lanes[lane].st));
    e->dx_free_pending[par] = true;
    return FE_OK;
}

#ifdef FOLVE_PHASE_TRACE
// TRACE build only: where the host's share of the one-block call goes (ns, summed; tools/phase_trace_single.py)
static unsigned long long g_host_ns[8];     // [0] entry -> launches enqueued, [1] -> completion seen, [2] calls, [3] polls, [4] exit -> next entry
static unsigned long long g_host_last_exit;
static unsigned long long host_now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec;
}
#define HOST_T(var) const unsigned long long var = host_now_ns()
#else
#define HOST_T(var) do {} while (0)
#endif

// peaks_out: optional [n][2] float bits fetched behind the outputs, under the same synchronisation.
int process_locked(fe_engine* e, fe_stream* const* streams, int n, const float* const* in,
                   const long long* nframes, float* const* out, int flags, unsigned int* peaks_out = nullptr,
                   hipEvent_t submit_event = nullptr, int lane = 0, long long* seq_out = nullptr,
                   float* const* block_peaks = nullptr, fe_ticket* ticket = nullptr, std::unique_lock<std::mutex>* lk = nullptr) {
    Lane& L = e->lanes[lane];
    const hipStream_t st = L.st;
    bool device_ptrs = (flags & FE_DEVICE_PTRS) != 0;
    const bool async = device_ptrs && (flags & FE_ASYNC);
    HOST_T(t_entry);
    HIP_TRY(hipSetDevice(e->device));

    // Host-pointer calls whose every buffer lies in page-locked memory bound to its stream run
    // zero-copy: the kernels read the PCM and write the result through the device's mapping of that
    // memory.  No staging copies (two DMA commands and their latency per call): this is the
    // single-block path of SoundProcessor::Process, where latency is everything.
    std::vector<const float*> zc_in;
    std::vector<float*> zc_out;
    bool zero_copy = false;
    const float* const* host_in = in;    // the callers' own pointers (run_duplex copies from them)
    float* const* host_out = out;
    if (!device_ptrs && n > 0) {
        bool all_bound = true;
        for (int i = 0; i < n && all_bound; ++i) {
            const fe_stream* s = streams[i];
            if (!s || !s->bound_host || nframes[i] < 0) { all_bound = false; break; }
            const char* lo = s->bound_host;
            const char* hi = lo + s->bound_bytes;
            const char* ib = reinterpret_cast<const char*>(in[i]);
            const char* ob = reinterpret_cast<const char*>(out[i]);
            const size_t nin = (size_t)nframes[i] * s->f->ninp * sizeof(float), nout = (size_t)nframes[i] * s->f->nout * sizeof(float);
            if (!(ib >= lo && ib + nin <= hi && ob >= lo && ob + nout <= hi)) all_bound = false;
        }
        if (all_bound) {
            zc_in.resize((size_t)n);
            zc_out.resize((size_t)n);
            for (int i = 0; i < n; ++i) {
                const fe_stream* s = streams[i];
                zc_in[(size_t)i] = reinterpret_cast<const float*>(s->bound_dev + (reinterpret_cast<const char*>(in[i]) - s->bound_host));
                zc_out[(size_t)i] = reinterpret_cast<float*>(s->bound_dev + (reinterpret_cast<char*>(out[i]) - s->bound_host));
            }
            in = zc_in.data();
            out = zc_out.data();
            device_ptrs = true;          // synchronous: the caller reads `out` when this returns
            zero_copy = true;
        }
    }
    if (submit_event && !zero_copy)
        return fail(FE_ERR_UNSUPPORTED, "fe_batch_submit needs every buffer inside page-locked memory bound to its stream");
    struct HostIoScope {                 // tells the launch rounds of THIS call where the PCM lives
        fe_engine* e;
        bool ended = false;
        HostIoScope(fe_engine* e_, bool on) : e(e_) { e->host_io = on; }
        void end() { if (!ended) { e->host_io = false; ended = true; } }
        ~HostIoScope() { end(); }
    } host_io_scope(e, zero_copy);

    std::vector<Item> all((size_t)n);
    std::vector<const float*> stage_out_of((size_t)n, nullptr);
    size_t in_floats = 0, out_floats = 0;
    for (int i = 0; i < n; ++i) {
        fe_stream* s = streams[i];
        if (!s || s->eng != e) return fail(FE_ERR_PARAM, "stream %d is null or on another engine", i);
        if (nframes[i] < 0) return fail(FE_ERR_PARAM, "negative frame count");
        if (nframes[i] > 0 && (!in[i] || !out[i])) return fail(FE_ERR_PARAM, "null buffer for stream %d", i);
        for (int k = 0; k < i; ++k)
            if (streams[k] == s) return fail(FE_ERR_PARAM, "stream listed twice in one batch");
        all[(size_t)i] = Item{s, in[i], out[i], nframes[i]};
        // staging offsets stay 16-byte aligned so that every stream takes the same kernel path
        // as it would alone (a batch is bit-identical to its streams run one by one)
        in_floats += ((size_t)nframes[i] * s->f->ninp + 3) & ~(size_t)3;
        out_floats += ((size_t)nframes[i] * s->f->nout + 3) & ~(size_t)3;
    }
    // A big submitted zero-copy batch runs as a duplex pipeline over both lanes (run_duplex).
    std::vector<int> lane_of((size_t)n, lane);
    DuplexPlan dplan;
    const size_t io_bytes = (in_floats + out_floats) * sizeof(float);
    const bool duplex = submit_event && zero_copy && !e->tuning_single_lane && !e->profiling && e->lanes[1].st && n >= 2 &&
                        io_bytes >= ((size_t)(e->duplex_min_mb > 0 ? e->duplex_min_mb : 32) << 20);
    if (duplex) {
        const size_t chunk_bytes = (size_t)(e->duplex_chunk_mb > 0 ? e->duplex_chunk_mb : 32) << 20;       // PCM in + out per chunk
        const int chunks = (int)std::max<size_t>(2, std::min<size_t>({(size_t)kDuplexChunks, (size_t)n, io_bytes / chunk_bytes + 1}));
        plan_duplex(streams, n, nframes, lane, chunks, &dplan, lane_of.data());
    }
    // A stream whose last call ran on another lane and is not known to have completed: its lane of this call
    // waits (on the device) for everything that lane holds so far.  One wait per pair of lanes covers all such streams.
    {
        bool waited[kLanes][kLanes] = {};
        for (int i = 0; i < n; ++i) {
            const fe_stream* s = streams[i];
            const int ol = s->last_lane, nl = lane_of[(size_t)i];
            if (ol < 0 || ol == nl || waited[nl][ol] || s->last_seq <= e->lanes[ol].done) continue;
            HIP_TRY(hipEventRecord(e->lanes[ol].xev, e->lanes[ol].st));
            HIP_TRY(hipStreamWaitEvent(e->lanes[nl].st, e->lanes[ol].xev, 0));
            waited[nl][ol] = true;
        }
    }
    const long long seq = L.submitted + 1;
    long long seqs[kLanes] = {0, 0};
    seqs[lane] = seq;
    if (duplex) seqs[lane ^ 1] = e->lanes[lane ^ 1].submitted + 1;
    if (seq_out) { seq_out[0] = seqs[0]; seq_out[1] = seqs[1]; }
    // (whatever happens below, kernels of this call may have been enqueued: the streams belong to their lanes now)
    struct LaneScope {
        fe_engine* e; fe_stream* const* streams; int n; const int* lane_of; const long long* seqs;
        bool ended = false;
        void end() {
            if (ended) return;
            ended = true;
            for (int l = 0; l < kLanes; ++l) if (seqs[l]) e->lanes[l].submitted = seqs[l];
            for (int i = 0; i < n; ++i) { streams[i]->last_lane = lane_of[i]; streams[i]->last_seq = seqs[lane_of[i]]; }
        }
        ~LaneScope() { end(); }
    } lane_scope{e, streams, n, lane_of.data(), seqs};
    if (!device_ptrs) {
        int rc = ensure_bytes(e, (void**)&e->stage_in, &e->stage_in_bytes, in_floats * sizeof(float));
        if (rc) return rc;
        rc = ensure_bytes(e, (void**)&e->stage_out, &e->stage_out_bytes, out_floats * sizeof(float));
        if (rc) return rc;
        size_t io = 0, oo = 0;
        for (int i = 0; i < n; ++i) {
            const size_t ni = (size_t)nframes[i] * streams[i]->f->ninp, no = (size_t)nframes[i] * streams[i]->f->nout;
            all[(size_t)i].in = e->stage_in + io;
            all[(size_t)i].out = e->stage_out + oo;
            stage_out_of[(size_t)i] = e->stage_out + oo;
            io += (ni + 3) & ~(size_t)3;
            oo += (no + 3) & ~(size_t)3;
        }
    }
    // per-block maxima (fe_batch_submit_peaks): one device array for the call, zeroed in front of the kernels, raised by K3,
    // copied into page-locked memory behind them; the ticket hands the blocks out to their streams' arrays
    unsigned int *pk_dev = nullptr, *pk_host = nullptr;
    size_t pk_bytes = 0;
    if (block_peaks && ticket && submit_event) {
        size_t blocks = 0;
        for (int i = 0; i < n; ++i) {
            if (!block_peaks[i] || nframes[i] <= 0) continue;
            const size_t nb = (size_t)((nframes[i] + streams[i]->f->P - 1) / streams[i]->f->P);
            ticket->peaks_dst.push_back(fe_ticket::PeakDst{block_peaks[i], blocks, nb});
            blocks += nb;
        }
        if (blocks) {
            int slot = -1;
            for (int k = 0; k < 4 && slot < 0; ++k) {
                const int c = (e->pkb_next + k) % 4;
                if (!e->pkb[c].in_use) slot = c;
            }
            if (slot < 0) return fail(FE_ERR_BUSY, "four batches with block maxima are outstanding: wait for their tickets");
            e->pkb_next = (slot + 1) % 4;
            fe_engine::PeakBuf& pb = e->pkb[slot];
            if (pb.cap < blocks) {
                if (pb.dev) HIP_TRY(hipFree(pb.dev));
                if (pb.host) HIP_TRY(hipHostFree(pb.host));
                pb.dev = nullptr; pb.host = nullptr; pb.cap = 0;
                const size_t cap = std::max<size_t>(blocks * 2, 4096);
                HIP_TRY(hipMalloc((void**)&pb.dev, cap * 2 * sizeof(unsigned int)));
                HIP_TRY(hipHostMalloc((void**)&pb.host, cap * 2 * sizeof(unsigned int), hipHostMallocDefault));
                pb.cap = cap;
            }
            pk_dev = pb.dev; pk_host = pb.host; pk_bytes = blocks * 2 * sizeof(unsigned int);
            size_t k = 0;
            for (int i = 0; i < n; ++i) {
                if (!block_peaks[i] || nframes[i] <= 0) continue;
                all[(size_t)i].blk_peaks = pk_dev + 2 * ticket->peaks_dst[k++].first;
            }
            pb.in_use = true;
            ticket->peaks_host = pk_host;
            ticket->peaks_slot = slot;
        }
    }
    // worth pipelining: several streams and enough bytes that the bus time dwarfs the extra events
    const bool pipelined = lane == 0 && !device_ptrs && !e->profiling && n >= 4 &&
                           (in_floats + out_floats) * sizeof(float) >= ((size_t)16 << 20);
    if (pipelined) {
        int rc = run_pipelined(e, streams, n, in, out, nframes, all);
        if (rc) return rc;
    } else {
        if (!device_ptrs) {
            for (int i = 0; i < n; ++i) {
                const size_t ni = (size_t)nframes[i] * streams[i]->f->ninp;
                if (ni) HIP_TRY(hipMemcpyAsync(const_cast<float*>(all[(size_t)i].in), in[i], ni * sizeof(float), hipMemcpyHostToDevice, st));
            }
        }
        if (pk_dev && !duplex) HIP_TRY(hipMemsetAsync(pk_dev, 0, pk_bytes, st));
        if (duplex) e->dx_quiet = 0;
        else release_idle_duplex_staging(e);
        int rc = duplex ? run_duplex(e, streams, all, n, host_in, host_out, nframes, lane, dplan, pk_dev, pk_host, pk_bytes)
                        : run_groups(e, streams, all, 0, n, lane, nullptr, /*spread=*/true);
        if (rc == FE_ERR_UNSUPPORTED && duplex) {
            // no staging to be had (device memory, the cap): nothing has been enqueued — the batch runs on `lane` with the
            // kernels reading and writing the callers' page-locked buffers, as small batches do
            for (int i = 0; i < n; ++i) lane_of[(size_t)i] = lane;     // (the bookkeeping of the plan that was not carried out)
            seqs[lane ^ 1] = 0;
            if (seq_out) seq_out[lane ^ 1] = 0;
            // (streams the plan had put on the other lane may have their last call there: everything it holds comes first)
            HIP_TRY(hipEventRecord(e->lanes[lane ^ 1].xev, e->lanes[lane ^ 1].st));
            HIP_TRY(hipStreamWaitEvent(st, e->lanes[lane ^ 1].xev, 0));
            if (pk_dev) HIP_TRY(hipMemsetAsync(pk_dev, 0, pk_bytes, st));
            rc = run_groups(e, streams, all, 0, n, lane, nullptr, /*spread=*/false);
            if (!rc && pk_dev) HIP_TRY(hipMemcpyAsync(pk_host, pk_dev, pk_bytes, hipMemcpyDeviceToHost, st));
        }
        if (rc) return rc;
        if (pk_dev && !duplex) HIP_TRY(hipMemcpyAsync(pk_host, pk_dev, pk_bytes, hipMemcpyDeviceToHost, st));
        if (!device_ptrs) {
            for (int i = 0; i < n; ++i) {
                const size_t no = (size_t)nframes[i] * streams[i]->f->nout;
                if (no) HIP_TRY(hipMemcpyAsync(out[i], stage_out_of[(size_t)i], no * sizeof(float), hipMemcpyDeviceToHost, st));
            }
        }
    }
    if (peaks_out) {
        for (int i = 0; i < n; ++i)
            HIP_TRY(hipMemcpyAsync(peaks_out + 2 * i, streams[i]->peaks, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost,
                                   st));
    }
    if (submit_event) {                  // fe_batch_submit: the caller waits on the ticket, not here
        HIP_TRY(hipEventRecord(submit_event, st));
        return FE_OK;
    }
    if (!async) {
        if (zero_copy) {
            HOST_T(t_launched);
            // The kernels are enqueued and nothing below touches the engine's shared buffers (a zero-copy call stages
            // nothing): the engine's lock is let go for the wait, so that another thread can submit a batch — it goes to the
            // other lane while this call is marked in flight on this one — instead of queueing behind a spinning caller.
            struct Unlocked {
                fe_engine* e; std::unique_lock<std::mutex>* lk;
                Unlocked(fe_engine* e_, std::unique_lock<std::mutex>* lk_) : e(e_), lk(lk_) { if (lk) { e->sync_in_flight++; lk->unlock(); } }
                ~Unlocked() { if (lk) { lk->lock(); e->sync_in_flight--; } }
            };
            host_io_scope.end();
            lane_scope.end();
            Unlocked unlocked(e, lk);
            // The latency path: poll for completion instead of sleeping on the runtime's interrupt
            // (the wake-up costs more than the kernels); bounded, then the ordinary wait.
            for (int spin = 0; spin < 4000; ++spin) {
                const hipError_t q = hipStreamQuery(st);
#ifdef FOLVE_PHASE_TRACE
                if (q == hipSuccess) {
                    const unsigned long long t_done = host_now_ns();
                    g_host_ns[0] += t_launched - t_entry; g_host_ns[1] += t_done - t_launched; g_host_ns[2] += 1; g_host_ns[3] += spin + 1;
                    if (g_host_last_exit) g_host_ns[4] += t_entry - g_host_last_exit;
                    g_host_last_exit = t_done;
                }
#endif
                if (q == hipSuccess) { (void)hipGetLastError(); if (lk) lk->lock(); L.done = std::max(L.done, seq); if (lk) lk->unlock(); return FE_OK; }   // (clears the sticky "not ready" of earlier polls)
                if (q != hipErrorNotReady) return fail(FE_ERR_DEVICE, "hipStreamQuery: %s", hipGetErrorString(q));
            }
            (void)hipGetLastError();
            HIP_TRY(hipStreamSynchronize(st));
            if (lk) lk->lock();
            L.done = std::max(L.done, seq);
            if (lk) lk->unlock();
            return FE_OK;
        }
        HIP_TRY(hipStreamSynchronize(st));
        L.done = std::max(L.done, seq);
    }
    return FE_OK;
}

// Before host-ordered work on lane 0 touches a stream's state (reset, peaks, close): order it behind the
// stream's last call if that ran on another lane and is not known to have completed.
int order_on_lane0(fe_engine* e, const fe_stream* s) {
    const int ol = s->last_lane;
    if (ol <= 0 || s->last_seq <= e->lanes[ol].done) return FE_OK;
    HIP_TRY(hipEventRecord(e->lanes[ol].xev, e->lanes[ol].st));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->lanes[ol].xev, 0));
    return FE_OK;
}

// The lane a submitted batch goes to: the one with fewer tickets outstanding, alternating on a tie.
// Lane 1 is created on first use.
int pick_lane(fe_engine* e, int* lane) {
    if (!e->lanes[1].st) {
        HIP_TRY(hipStreamCreateWithFlags(&e->lanes[1].st, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->lanes[1].xev, hipEventDisableTiming));
    }
    int l;
    if (e->sync_in_flight > 0) l = 1;    // lane 0 carries a synchronous call whose caller polls that stream: stay off it
    else if (e->lanes[0].outstanding != e->lanes[1].outstanding) l = e->lanes[0].outstanding < e->lanes[1].outstanding ? 0 : 1;
    else { l = e->lane_toggle; e->lane_toggle ^= 1; }
    *lane = e->tuning_single_lane ? 0 : l;
    return FE_OK;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

#ifdef FOLVE_PHASE_TRACE
int fe_debug_host_times(unsigned long long* out8, int reset) {
    for (int i = 0; i < 8; ++i) out8[i] = g_host_ns[i];
    if (reset) { for (int i = 0; i < 8; ++i) g_host_ns[i] = 0; g_host_last_exit = 0; }
    return 0;
}
#endif

const char* fe_last_error(void) { return g_last_error.c_str(); }

int fe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fe_fragm_for_size(unsigned int maxsize) {
    unsigned int fragm = FE_MAXQUANT;
#ifdef FOLVE_EXPERIMENT_P16
    // measurement build only (tools/build_p16.sh, tools/p16_experiment.py): every long filter at a 16 384-frame partition,
    // at the boundary too — what a 16 384-point internal partition would do to K1 / K2 / K3
    if (getenv("FOLVE_P16") && maxsize > 16384) return 16384;
#endif
    while (fragm > FE_MINPART && fragm >= 2 * maxsize) fragm /= 2;
    return (int)fragm;
}

int fe_engine_create(int device, void* hip_stream, fe_engine** out) {
    if (!out) return fail(FE_ERR_PARAM, "null out");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FE_ERR_DEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FE_ERR_PARAM, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    fe_engine* e = new (std::nothrow) fe_engine();
    if (!e) return fail(FE_ERR_ALLOC, "out of memory");
    e->device = device;
    if (hip_stream) {
        e->stream = (hipStream_t)hip_stream;
    } else {
        hipError_t r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (r != hipSuccess) { delete e; return fail(FE_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(r)); }
        e->own_stream = true;
    }
    e->lanes[0].st = e->stream;
    if (hipEventCreateWithFlags(&e->lanes[0].xev, hipEventDisableTiming) != hipSuccess) {
        delete e;
        return fail(FE_ERR_DEVICE, "hipEventCreate failed");
    }
    for (int i = 0; i < kJobSlots; ++i) {
        if (hipEventCreateWithFlags(&e->jobs_ev[i], hipEventDisableTiming) != hipSuccess) {
            delete e;
            return fail(FE_ERR_DEVICE, "hipEventCreate failed");
        }
    }
    // every descriptor slot's buffers up front, carved out of two slabs
    if (hipHostMalloc((void**)&e->jobs_slab_host, (size_t)kJobSlots * kSlabJobs * sizeof(fk::StreamJob), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&e->jobs_slab_dev, (size_t)kJobSlots * kSlabJobs * sizeof(fk::StreamJob)) != hipSuccess) {
        (void)hipGetLastError();
        delete e;
        return fail(FE_ERR_ALLOC, "descriptor buffers: allocation failed");
    }
    for (int i = 0; i < kJobSlots; ++i) {
        e->jobs_host[i] = e->jobs_slab_host + (size_t)i * kSlabJobs;
        e->jobs_dev[i] = e->jobs_slab_dev + (size_t)i * kSlabJobs;
        e->jobs_cap[i] = kSlabJobs * sizeof(fk::StreamJob);
    }
    for (int i = 0; i < 4; ++i) {
        if (hipEventCreate(&e->pev[i]) != hipSuccess) { delete e; return fail(FE_ERR_DEVICE, "hipEventCreate failed"); }
    }
    *out = e;
    return FE_OK;
}

static void engine_release(fe_engine* e) {
    if (!e) return;
    if (e->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(e->device);
    for (Lane& L : e->lanes) {
        if (L.st) (void)hipStreamSynchronize(L.st);
        if (L.Y) (void)hipFree(L.Y);
        if (L.xev) (void)hipEventDestroy(L.xev);
    }
    if (e->lanes[1].st) (void)hipStreamDestroy(e->lanes[1].st);
    for (auto& kv : e->tw) (void)hipFree(kv.second);
    if (e->cp_in) { (void)hipStreamSynchronize(e->cp_in); (void)hipStreamDestroy(e->cp_in); }
    if (e->cp_out) { (void)hipStreamSynchronize(e->cp_out); (void)hipStreamDestroy(e->cp_out); }
    for (int i = 0; i < kMaxChunks; ++i) {
        if (e->ev_in[i]) (void)hipEventDestroy(e->ev_in[i]);
        if (e->ev_k[i]) (void)hipEventDestroy(e->ev_k[i]);
    }
    for (hipEvent_t ev : e->dx_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->split_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->dx_k3) if (ev) (void)hipEventDestroy(ev);
    for (fe_engine::PeakBuf& pb : e->pkb) {
        if (pb.dev) (void)hipFree(pb.dev);
        if (pb.host) (void)hipHostFree(pb.host);
    }
    for (int i = 0; i < 2; ++i) {
        if (e->dx_free[i]) (void)hipEventDestroy(e->dx_free[i]);
        if (e->dx_stage[i]) (void)hipFree(e->dx_stage[i]);
        if (e->dx_stage_out[i]) (void)hipFree(e->dx_stage_out[i]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->stage_in) (void)hipFree(e->stage_in);
    if (e->stage_out) (void)hipFree(e->stage_out);
    for (int i = 0; i < kJobSlots; ++i) {
        if (e->jobs_own[i] && e->jobs_host[i]) (void)hipHostFree(e->jobs_host[i]);
        if (e->jobs_own[i] && e->jobs_dev[i]) (void)hipFree(e->jobs_dev[i]);
        if (e->jobs_ev[i]) (void)hipEventDestroy(e->jobs_ev[i]);
    }
    if (e->jobs_slab_host) (void)hipHostFree(e->jobs_slab_host);
    if (e->jobs_slab_dev) (void)hipFree(e->jobs_slab_dev);
    for (int i = 0; i < 4; ++i) if (e->pev[i]) (void)hipEventDestroy(e->pev[i]);
    for (auto& set : e->kev)
        for (auto& role : set)
            for (hipEvent_t ev : role) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ticket_events) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// Drops the creator's reference; the engine lives on until the last filter
// (and so the last stream) created on it is gone.
void fe_engine_destroy(fe_engine* e) { engine_release(e); }

int fe_engine_synchronize(fe_engine* e) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    for (Lane& L : e->lanes) {
        if (!L.st) continue;
        const long long upto = L.submitted;
        HIP_TRY(hipStreamSynchronize(L.st));
        L.done = upto;
    }
    return FE_OK;
}

int fe_engine_device(const fe_engine* e) { return e ? e->device : -1; }

int fe_device_local_cpulist(int device, char* buf, size_t size) {
    if (!buf || size < 2) return fail(FE_ERR_PARAM, "bad buffer");
    buf[0] = 0;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FE_ERR_DEVICE, "no PCI bus id for device %d", device);
    }
    for (char* p = bus; *p; ++p) *p = (char)tolower((unsigned char)*p);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE* fp = fopen(path, "r");
    if (!fp) return fail(FE_ERR_UNSUPPORTED, "%s not readable", path);
    const bool got = fgets(buf, (int)size, fp) != nullptr;
    fclose(fp);
    if (!got) return fail(FE_ERR_UNSUPPORTED, "%s is empty", path);
    buf[strcspn(buf, "\r\n")] = 0;
    if (!buf[0]) return fail(FE_ERR_UNSUPPORTED, "%s is empty", path);
    return FE_OK;
}

// ---- filter ---------------------------------------------------------------
int fe_filter_create(fe_engine* e, int ninp, int nout, int maxsize, float density, fe_filter** out) {
    if (!out) return fail(FE_ERR_PARAM, "null out");
    *out = nullptr;
    // e may be NULL: such a filter can be assembled and inspected but not committed
    if (ninp < 1 || ninp > FE_MAXINP) return fail(FE_ERR_PARAM, "inputs %d out of range", ninp);
    if (nout < 1 || nout > FE_MAXOUT) return fail(FE_ERR_PARAM, "outputs %d out of range", nout);
    if (maxsize < 1 || maxsize > FE_MAXSIZE) return fail(FE_ERR_PARAM, "size %d out of range", maxsize);
    if (!(density >= 0.0f && density <= 1.0f)) return fail(FE_ERR_PARAM, "density out of range");
    fe_filter* f = new (std::nothrow) fe_filter();
    if (!f) return fail(FE_ERR_ALLOC, "out of memory");
    f->eng = e;
    if (e) e->refs.fetch_add(1);
    f->ninp = ninp; f->nout = nout; f->size = maxsize; f->density = density;
    f->P = fe_fragm_for_size((unsigned)maxsize);
    f->log2P = 0;
    while ((1 << f->log2P) < f->P) f->log2P++;
    f->K = (maxsize + f->P - 1) / f->P;
    f->paths.resize((size_t)ninp * nout);
    *out = f;
    return FE_OK;
}

int fe_filter_add(fe_filter* f, int inp, int out, int step, const float* data, int ind0, int ind1) {
    if (!f) return fail(FE_ERR_PARAM, "null filter");
    if (f->committed) return fail(FE_ERR_STATE, "filter already committed");
    if (inp < 0 || inp >= f->ninp || out < 0 || out >= f->nout) return fail(FE_ERR_PARAM, "bad input/output");
    if (ind0 < 0 || ind1 < ind0 || step < 1 || (!data && ind1 > ind0)) return fail(FE_ERR_PARAM, "bad range");
    const int self = inp * f->nout + out;
    // A pair that is a link (/impulse/copy target) takes no data of its own: zita's impdata_create
    // returns without touching a linked node, and so does this.
    if (f->paths[(size_t)self].link >= 0) return FE_OK;
    PathHost& p = f->paths[(size_t)self];
    p.used = true;
    const long long cap = (long long)f->K * f->P;
    if (p.taps.empty()) {
        try { p.taps.assign((size_t)cap, 0.0f); } catch (...) { return fail(FE_ERR_ALLOC, "out of memory"); }
    }
    const long long hi = std::min<long long>(ind1, cap);
    for (long long t = ind0; t < hi; ++t) p.taps[(size_t)t] += data[(size_t)(t - ind0) * (size_t)step];
    if (hi > ind0) {
        for (int k = ind0 / f->P; k <= (int)((hi - 1) / f->P); ++k) p.mask[k >> 5] |= 1u << (k & 31);
    }
    return FE_OK;
}

int fe_filter_link(fe_filter* f, int inp1, int out1, int inp2, int out2) {
    if (!f) return fail(FE_ERR_PARAM, "null filter");
    if (f->committed) return fail(FE_ERR_STATE, "filter already committed");
    if (inp1 < 0 || inp1 >= f->ninp || out1 < 0 || out1 >= f->nout) return fail(FE_ERR_PARAM, "bad source pair");
    if (inp2 < 0 || inp2 >= f->ninp || out2 < 0 || out2 >= f->nout) return fail(FE_ERR_PARAM, "bad target pair");
    const int src = inp1 * f->nout + out1, dst = inp2 * f->nout + out2;
    if (src == dst) return fail(FE_ERR_PARAM, "cannot link a pair to itself");
    // As zita's impdata_copy: nothing happens when the source pair has no node yet (no data was
    // ever added to it, nor was it linked) or when the target already holds data of its own.
    const PathHost& sp = f->paths[(size_t)src];
    PathHost& d = f->paths[(size_t)dst];
    if (!sp.used) return FE_OK;
    if (!d.taps.empty()) return FE_OK;
    for (int at = src, guard = 0; at >= 0 && guard < 8192; at = f->paths[(size_t)at].link, ++guard)
        if (at == dst) return fail(FE_ERR_PARAM, "link would form a cycle");   // (undefined behaviour in zita)
    d.link = src;
    d.used = true;
    return FE_OK;
}

int fe_filter_commit(fe_filter* f) {
    if (!f) return fail(FE_ERR_PARAM, "null filter");
    if (f->committed) return FE_OK;
    fe_engine* e = f->eng;
    if (!e) return fail(FE_ERR_DEVICE, "filter has no engine: cannot commit (no CPU fallback)");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    const int P = f->P, K = f->K, np = f->ninp * f->nout;
    std::vector<int> data_of((size_t)np, -1);
    std::vector<int> owners;
    for (int i = 0; i < np; ++i) {
        const PathHost& p = f->paths[(size_t)i];
        if (p.used && p.link < 0 && !p.taps.empty()) { data_of[(size_t)i] = (int)owners.size(); owners.push_back(i); }
    }
    f->ndata = (int)owners.size();
    std::vector<fk::PathEntry> entries;
    std::vector<int> out_first((size_t)f->nout + 1, 0);
    for (int o = 0; o < f->nout; ++o) {
        out_first[(size_t)o] = (int)entries.size();
        for (int i = 0; i < f->ninp; ++i) {
            const int idx = i * f->nout + o;
            if (!f->paths[(size_t)idx].used) continue;
            const int d = data_of[(size_t)resolve(f, idx)];
            if (d >= 0) entries.push_back(fk::PathEntry{i, d});
        }
    }
    out_first[(size_t)f->nout] = (int)entries.size();

    // every output with exactly one input path, and how dense the populated-row bitmaps are (K2's form choice)
    f->mac_shape.single_path = true;
    f->mac_shape.max_paths = 0;
    f->mac_shape.ndata = f->ndata;
    for (int o = 0; o < f->nout; ++o) {
        const int n = out_first[(size_t)o + 1] - out_first[(size_t)o];
        if (n > 1) f->mac_shape.single_path = false;
        f->mac_shape.max_paths = std::max(f->mac_shape.max_paths, n);
    }
    {
        long long rows = 0, pop = 0;
        for (int d : owners) {
            const PathHost& p = f->paths[(size_t)d];
            rows += K;
            for (int w = 0; w < 4; ++w) pop += __builtin_popcount(p.mask[w]);
        }
        f->mac_shape.dense = rows > 0 && pop * 10 >= rows * 6;
    }

    fk::FftTables tabs{};
    int rc = get_twiddles(e, f->log2P, &tabs);
    if (rc) return rc;
    // a retry after a partial failure must not leak the earlier attempt's buffers
    if (f->out_first_dev) { (void)hipFree(f->out_first_dev); f->out_first_dev = nullptr; }
    if (f->paths_dev) { (void)hipFree(f->paths_dev); f->paths_dev = nullptr; }
    if (f->H) { (void)hipFree(f->H); f->H = nullptr; }
    if (f->mask_dev) { (void)hipFree(f->mask_dev); f->mask_dev = nullptr; }
    HIP_TRY(hipMalloc((void**)&f->out_first_dev, out_first.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(f->out_first_dev, out_first.data(), out_first.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!entries.empty()) {
        HIP_TRY(hipMalloc((void**)&f->paths_dev, entries.size() * sizeof(fk::PathEntry)));
        HIP_TRY(hipMemcpy(f->paths_dev, entries.data(), entries.size() * sizeof(fk::PathEntry), hipMemcpyHostToDevice));
    }
    if (f->ndata > 0) {
        const size_t per = (size_t)K * P;              // taps / H per data path
        const size_t gper = (size_t)(K + 1) * P;       // G rows per data path
        // populated rows of G(j) = s*H(j) + H(j-1): wherever H(j) or H(j-1) is populated
        std::vector<uint64_t> masks((size_t)f->ndata * 4, 0);
        DevTmp taps_tmp, spectra_tmp;          // taps and the H rows live only until G exists
        HIP_TRY(hipMalloc(&taps_tmp.p, per * f->ndata * sizeof(float)));
        float* taps_dev = static_cast<float*>(taps_tmp.p);
        for (int d = 0; d < f->ndata; ++d) {
            const PathHost& p = f->paths[(size_t)owners[(size_t)d]];
            for (int j = 0; j <= K; ++j) {
                const bool hj = j < K && ((p.mask[j >> 5] >> (j & 31)) & 1u);
                const bool hp = j >= 1 && ((p.mask[(j - 1) >> 5] >> ((j - 1) & 31)) & 1u);
                if (hj || hp) masks[(size_t)d * 4 + (size_t)(j >> 6)] |= 1ull << (j & 63);
            }
            hipError_t r = hipMemcpy(taps_dev + per * d, p.taps.data(), per * sizeof(float), hipMemcpyHostToDevice);
            if (r != hipSuccess) return fail(FE_ERR_DEVICE, "tap upload: %s", hipGetErrorString(r));
        }
        HIP_TRY(hipMalloc(&spectra_tmp.p, per * f->ndata * sizeof(float2)));
        float2* h_tmp = static_cast<float2*>(spectra_tmp.p);
        HIP_TRY(hipMalloc((void**)&f->H, gper * f->ndata * sizeof(float2)));
        HIP_TRY(hipMalloc((void**)&f->mask_dev, masks.size() * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(f->mask_dev, masks.data(), masks.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(fk::launch_filter_transform(taps_dev, h_tmp, f->H, f->ndata, K, f->log2P, tabs, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    f->dev.cin = f->ninp; f->dev.cout = f->nout; f->dev.P = P; f->dev.log2P = f->log2P; f->dev.K = K + 1;   // rows of G
    f->dev.H = f->H; f->dev.mask = f->mask_dev; f->dev.paths = f->paths_dev; f->dev.out_first = f->out_first_dev;
    f->dev.tw = tabs.tw;
    f->dev.twa = tabs.twa;
    f->dev.twb = tabs.twb;
    f->dev.twa2 = tabs.twa2;
    f->dev.twb2 = tabs.twb2;
    f->committed = true;
    return FE_OK;
}

void fe_filter_retain(fe_filter* f) { if (f) f->refs.fetch_add(1); }
int fe_filter_use_count(const fe_filter* f) { return f ? f->refs.load() : 0; }

void fe_filter_release(fe_filter* f) {
    if (!f) return;
    if (f->refs.fetch_sub(1) != 1) return;
    if (f->eng) {
        (void)hipSetDevice(f->eng->device);
        if (f->committed) (void)hipStreamSynchronize(f->eng->stream);
    }
    if (f->H) (void)hipFree(f->H);
    if (f->mask_dev) (void)hipFree(f->mask_dev);
    if (f->paths_dev) (void)hipFree(f->paths_dev);
    if (f->out_first_dev) (void)hipFree(f->out_first_dev);
    fe_engine* e = f->eng;
    delete f;
    engine_release(e);
}

int fe_filter_inputs(const fe_filter* f) { return f ? f->ninp : 0; }
int fe_filter_outputs(const fe_filter* f) { return f ? f->nout : 0; }
int fe_filter_block_size(const fe_filter* f) { return f ? f->P : 0; }
int fe_filter_partitions(const fe_filter* f) { return f ? f->K : 0; }
int fe_filter_maxsize(const fe_filter* f) { return f ? f->size : 0; }

int fe_filter_path_partitions(const fe_filter* f, int inp, int out) {
    if (!f || inp < 0 || inp >= f->ninp || out < 0 || out >= f->nout) return 0;
    const int idx = inp * f->nout + out;
    if (!f->paths[(size_t)idx].used) return 0;
    const PathHost& p = f->paths[(size_t)resolve(f, idx)];
    int n = 0;
    for (int w = 0; w < 4; ++w) n += __builtin_popcount(p.mask[w]);
    return n;
}

int fe_filter_get_taps(const fe_filter* f, int inp, int out, float* dst, int n) {
    if (!f || !dst || n < 0) return fail(FE_ERR_PARAM, "bad argument");
    if (inp < 0 || inp >= f->ninp || out < 0 || out >= f->nout) return fail(FE_ERR_PARAM, "bad input/output");
    memset(dst, 0, sizeof(float) * (size_t)n);
    const int idx = inp * f->nout + out;
    if (!f->paths[(size_t)idx].used) return FE_OK;
    const PathHost& p = f->paths[(size_t)resolve(f, idx)];
    const size_t m = std::min<size_t>((size_t)n, p.taps.size());
    if (m) memcpy(dst, p.taps.data(), m * sizeof(float));
    return FE_OK;
}

// ---- stream ---------------------------------------------------------------
int fe_stream_open(fe_filter* f, int max_blocks_per_call, fe_stream** out) {
    if (!out) return fail(FE_ERR_PARAM, "null out");
    *out = nullptr;
    if (!f) return fail(FE_ERR_PARAM, "null filter");
    if (!f->committed) return fail(FE_ERR_STATE, "filter not committed");
    if (max_blocks_per_call < 1 || max_blocks_per_call > 4096) return fail(FE_ERR_PARAM, "max_blocks_per_call out of range");
    fe_engine* e = f->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    fe_stream* s = new (std::nothrow) fe_stream();
    if (!s) return fail(FE_ERR_ALLOC, "out of memory");
    s->f = f; s->eng = e;
    s->max_blocks = max_blocks_per_call;
    s->ring = f->K + max_blocks_per_call;           // K + 1 rows of G reach K blocks back
    s->fdl_bytes = (size_t)f->ninp * s->ring * f->P * sizeof(float2);
    hipError_t r = hipMalloc((void**)&s->fdl, s->fdl_bytes);
    if (r == hipSuccess) r = hipMalloc((void**)&s->peaks, 2 * sizeof(unsigned int));
    if (r == hipSuccess) r = hipMemsetAsync(s->fdl, 0, s->fdl_bytes, e->stream);
    if (r == hipSuccess) r = hipMemsetAsync(s->peaks, 0, 2 * sizeof(unsigned int), e->stream);
    if (r != hipSuccess) {
        if (s->fdl) (void)hipFree(s->fdl);
        if (s->peaks) (void)hipFree(s->peaks);
        delete s;
        return fail(r == hipErrorOutOfMemory ? FE_ERR_ALLOC : FE_ERR_DEVICE, "stream allocation: %s", hipGetErrorString(r));
    }
    fe_filter_retain(f);
    *out = s;
    return FE_OK;
}

int fe_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(FE_ERR_PARAM, "null out");
    *out = nullptr;
    if (bytes == 0) return fail(FE_ERR_PARAM, "zero bytes");
    // portable + mapped: every GPU of the process can address it (a processor's buffer is allocated
    // before the router's choice of GPU is visible to the allocator)
    hipError_t r = hipHostMalloc(out, bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (r != hipSuccess) {
        *out = nullptr;
        return fail(r == hipErrorOutOfMemory ? FE_ERR_ALLOC : FE_ERR_DEVICE, "hipHostMalloc: %s", hipGetErrorString(r));
    }
    return FE_OK;
}

void fe_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int fe_stream_bind_host_buffer(fe_stream* s, void* buf, size_t bytes) {
    if (!s) return fail(FE_ERR_PARAM, "null stream");
    fe_engine* e = s->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    s->bound_host = nullptr; s->bound_dev = nullptr; s->bound_bytes = 0;
    if (!buf || bytes == 0) return FE_OK;            // unbind
    void* dev = nullptr;
    hipError_t r = hipHostGetDevicePointer(&dev, buf, 0);
    if (r != hipSuccess || !dev) {
        (void)hipGetLastError();
        return fail(FE_ERR_PARAM, "buffer is not page-locked memory the device can address (use fe_host_alloc)");
    }
    s->bound_host = static_cast<const char*>(buf);
    s->bound_dev = static_cast<char*>(dev);
    s->bound_bytes = bytes;
    return FE_OK;
}

int fe_stream_reset(fe_stream* s) {
    if (!s) return fail(FE_ERR_PARAM, "null stream");
    fe_engine* e = s->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    int rc = order_on_lane0(e, s);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(s->fdl, 0, s->fdl_bytes, e->stream));
    HIP_TRY(hipMemsetAsync(s->peaks, 0, 2 * sizeof(unsigned int), e->stream));
    s->slot0 = 0;
    s->blocks_done = 0;
    s->last_lane = 0;
    s->last_seq = ++e->lanes[0].submitted;
    return FE_OK;
}

void fe_stream_close(fe_stream* s) {
    if (!s) return;
    fe_engine* e = s->eng;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        (void)hipSetDevice(e->device);
        (void)hipStreamSynchronize(e->stream);
        if (s->last_lane > 0 && s->last_seq > e->lanes[s->last_lane].done) (void)hipStreamSynchronize(e->lanes[s->last_lane].st);
        (void)hipFree(s->fdl);
        (void)hipFree(s->peaks);
    }
    fe_filter_release(s->f);
    delete s;
}

int fe_stream_get_peaks(fe_stream* s, float* peak_signed, float* peak_abs) {
    if (!s) return fail(FE_ERR_PARAM, "null stream");
    fe_engine* e = s->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    unsigned int bits[2] = {0, 0};
    int orc = order_on_lane0(e, s);
    if (orc) return orc;
    HIP_TRY(hipMemcpyAsync(bits, s->peaks, sizeof(bits), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float v[2];
    memcpy(v, bits, sizeof(v));
    if (peak_signed) *peak_signed = v[0];
    if (peak_abs) *peak_abs = v[1];
    return FE_OK;
}

int fe_batch_get_peaks(fe_stream* const* streams, int n, float* peak_signed, float* peak_abs) {
    if (n < 0 || (n > 0 && !streams)) return fail(FE_ERR_PARAM, "bad arguments");
    if (n == 0) return FE_OK;
    if (!streams[0]) return fail(FE_ERR_PARAM, "stream 0 is null");
    fe_engine* e = streams[0]->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    std::vector<unsigned int> bits((size_t)n * 2, 0u);
    for (int i = 0; i < n; ++i) {
        if (!streams[i] || streams[i]->eng != e) return fail(FE_ERR_PARAM, "stream %d is null or on another engine", i);
        int orc = order_on_lane0(e, streams[i]);
        if (orc) return orc;
        HIP_TRY(hipMemcpyAsync(&bits[(size_t)i * 2], streams[i]->peaks, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost,
                               e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n; ++i) {
        float v[2];
        memcpy(v, &bits[(size_t)i * 2], sizeof(v));
        if (peak_signed) peak_signed[i] = v[0];
        if (peak_abs) peak_abs[i] = v[1];
    }
    return FE_OK;
}

int fe_stream_reset_peaks(fe_stream* s) {
    if (!s) return fail(FE_ERR_PARAM, "null stream");
    fe_engine* e = s->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    int orc = order_on_lane0(e, s);
    if (orc) return orc;
    HIP_TRY(hipMemsetAsync(s->peaks, 0, 2 * sizeof(unsigned int), e->stream));
    s->last_lane = 0;
    s->last_seq = ++e->lanes[0].submitted;
    return FE_OK;
}

long long fe_stream_blocks_done(const fe_stream* s) { return s ? s->blocks_done : 0; }
int fe_stream_block_size(const fe_stream* s) { return s ? s->f->P : 0; }
int fe_stream_max_blocks(const fe_stream* s) { return s ? s->max_blocks : 0; }

int fe_batch_process(fe_stream* const* streams, int n, const float* const* in, const long long* nframes,
                     float* const* out, int flags) {
    if (n < 0 || (n > 0 && (!streams || !in || !nframes || !out))) return fail(FE_ERR_PARAM, "bad batch arguments");
    if (n == 0) return FE_OK;
    if (!streams[0]) return fail(FE_ERR_PARAM, "null stream");
    fe_engine* e = streams[0]->eng;
    std::unique_lock<std::mutex> lk(e->mu);
    return process_locked(e, streams, n, in, nframes, out, flags, nullptr, nullptr, 0, nullptr, nullptr, nullptr, &lk);
}

int fe_batch_submit(fe_stream* const* streams, int n, const float* const* in, const long long* nframes,
                    float* const* out, fe_ticket** ticket) {
    return fe_batch_submit_peaks(streams, n, in, nframes, out, nullptr, ticket);
}

int fe_batch_submit_peaks(fe_stream* const* streams, int n, const float* const* in, const long long* nframes,
                          float* const* out, float* const* block_peaks, fe_ticket** ticket) {
    if (!ticket) return fail(FE_ERR_PARAM, "null ticket");
    *ticket = nullptr;
    if (n < 1 || !streams || !in || !nframes || !out) return fail(FE_ERR_PARAM, "bad batch arguments");
    if (!streams[0]) return fail(FE_ERR_PARAM, "null stream");
    fe_engine* e = streams[0]->eng;
    fe_ticket* t = new (std::nothrow) fe_ticket();
    if (!t) return fail(FE_ERR_ALLOC, "out of memory");
    std::lock_guard<std::mutex> lk(e->mu);
    if (hipSetDevice(e->device) != hipSuccess) {
        delete t;
        return fail(FE_ERR_DEVICE, "hipSetDevice(%d) failed", e->device);
    }
    int lane = 0;
    if (int lrc = pick_lane(e, &lane)) { delete t; return lrc; }
    if (!e->ticket_events.empty()) {
        t->ev = e->ticket_events.back();
        e->ticket_events.pop_back();
    } else if (hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        delete t;
        return fail(FE_ERR_DEVICE, "hipEventCreate failed");
    }
    const int rc = process_locked(e, streams, n, in, nframes, out, FE_HOST_PTRS, nullptr, t->ev, lane, t->seq, block_peaks, t);
    if (rc) {
        if (t->peaks_slot >= 0) e->pkb[t->peaks_slot].in_use = false;
        e->ticket_events.push_back(t->ev);      // (never recorded: nothing pends on it)
        delete t;
        return rc;
    }
    t->e = e;
    t->lane = lane;
    e->lanes[lane].outstanding++;
    e->refs.fetch_add(1);
    *ticket = t;
    return FE_OK;
}

int fe_ticket_done(fe_ticket* t) {
    if (!t) return fail(FE_ERR_PARAM, "null ticket");
    const hipError_t q = hipEventQuery(t->ev);
    (void)hipGetLastError();
    if (q == hipSuccess) return 1;
    if (q == hipErrorNotReady) return 0;
    return fail(FE_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(q));
}

int fe_ticket_wait(fe_ticket* t) {
    if (!t) return fail(FE_ERR_PARAM, "null ticket");
    fe_engine* e = t->e;
    int rc = FE_OK;
    // (events can be waited for from any device context: a failing hipSetDevice does not excuse the wait —
    //  the caller is about to be told that its buffers are its own again)
    (void)hipSetDevice(e->device);
    (void)hipGetLastError();
    // the latency path: poll (bounded), then the runtime's wait — as fe_stream_process does
    bool done = false;
    for (int spin = 0; rc == FE_OK && !done && spin < 4000; ++spin) {
        const hipError_t q = hipEventQuery(t->ev);
        if (q == hipSuccess) done = true;
        else if (q != hipErrorNotReady) rc = fail(FE_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(q));
    }
    (void)hipGetLastError();             // (clears the sticky "not ready" of the polls)
    if (!done) {
        const hipError_t w = hipEventSynchronize(t->ev);
        if (w != hipSuccess && rc == FE_OK) rc = fail(FE_ERR_DEVICE, "hipEventSynchronize: %s", hipGetErrorString(w));
        if (w == hipSuccess) done = true;
    }
    {
        std::lock_guard<std::mutex> lk(e->mu);
        Lane& L = e->lanes[t->lane];
        L.outstanding--;
        if (done) {
            for (int l = 0; l < kLanes; ++l)
                if (t->seq[l] > e->lanes[l].done) e->lanes[l].done = t->seq[l];
            if (rc == FE_OK)
                for (const fe_ticket::PeakDst& d : t->peaks_dst) memcpy(d.dst, t->peaks_host + 2 * d.first, d.blocks * 2 * sizeof(float));
            e->ticket_events.push_back(t->ev);
        } else {
            (void)hipEventDestroy(t->ev);        // may still be pending: never recycled
        }
        if (t->peaks_slot >= 0) e->pkb[t->peaks_slot].in_use = false;
    }
    delete t;
    engine_release(e);
    return rc;
}

int fe_stream_process_blocks(fe_stream* s, const float* in, long long nframes, float* out) {
    fe_stream* ss[1] = {s};
    const float* ii[1] = {in};
    float* oo[1] = {out};
    long long nn[1] = {nframes};
    return fe_batch_process(ss, 1, ii, nn, oo, FE_HOST_PTRS);
}

int fe_stream_process(fe_stream* s, const float* in, int valid_frames, float* out, float* peak_signed,
                      float* peak_abs) {
    if (!s) return fail(FE_ERR_PARAM, "null stream");
    if (valid_frames < 1 || valid_frames > s->f->P) return fail(FE_ERR_PARAM, "valid_frames must be in 1..block size");
    if (!in || !out) return fail(FE_ERR_PARAM, "null buffer");
    fe_stream* ss[1] = {s};
    const float* ii[1] = {in};
    float* oo[1] = {out};
    long long nn[1] = {valid_frames};
    unsigned int bits[2] = {0u, 0u};
    fe_engine* e = s->eng;
    std::unique_lock<std::mutex> lk(e->mu);
    // one synchronisation for the block and its peaks
    const int rc = process_locked(e, ss, 1, ii, nn, oo, FE_HOST_PTRS, (peak_signed || peak_abs) ? bits : nullptr, nullptr, 0, nullptr,
                                  nullptr, nullptr, &lk);
    if (rc) return rc;
    float v[2];
    memcpy(v, bits, sizeof(v));       // (zeros when no peak was asked for)
    if (peak_signed) *peak_signed = v[0];
    if (peak_abs) *peak_abs = v[1];
    return FE_OK;
}

// ---- launch-shape overrides and device self-tests -------------------------------------------
int fe_engine_set_tuning(fe_engine* e, int knob, int value) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    switch (knob) {
        case FE_TUNE_FWD_RUN:
        case FE_TUNE_INV_RUN:
            if (value < 0 || value > 4096) return fail(FE_ERR_PARAM, "run length out of range");
            (knob == FE_TUNE_FWD_RUN ? e->tuning.fwd_run : e->tuning.inv_run) = value;
            return FE_OK;
        case FE_TUNE_MAC_FORM:
            if (value != 0 && value != 1 && value != 4 && value != 8 && value != 16 && value != 100)
                return fail(FE_ERR_PARAM, "MAC form must be 0, 1, 4, 8, 16 or 100");
            e->tuning.mac_form = value;
            return FE_OK;
        case FE_TUNE_FFT_FORM:
            if (value < 0 || value > 4) return fail(FE_ERR_PARAM, "FFT form must be 0 .. 4");
            e->tuning.fft_form = value;
            return FE_OK;
        case FE_TUNE_WALK_LPB:
            if (value != 0 && value != 1 && value != 2 && value != 4) return fail(FE_ERR_PARAM, "lanes per bin must be 0, 1, 2 or 4");
            e->tuning.walk_lpb = value;
            return FE_OK;
        case FE_TUNE_WALK_TILES:
            if (value < 0 || value > 64) return fail(FE_ERR_PARAM, "time tiles must be 0 .. 64");
            e->tuning.walk_tiles = value;
            return FE_OK;
        case FE_TUNE_WALK_FMA:
            if (value != 0 && value != 3 && value != 4) return fail(FE_ERR_PARAM, "walk FMA form must be 0, 3 or 4");
            e->tuning.walk_fma = value;
            return FE_OK;
        case FE_TUNE_WALK_NT:
            if (value < 0 || value > 2) return fail(FE_ERR_PARAM, "walk nt must be 0, 1 or 2");
            e->tuning.walk_nt = value;
            return FE_OK;
        case FE_TUNE_DUPLEX_OUT:
            if (value < 0 || value > 2) return fail(FE_ERR_PARAM, "duplex out must be 0, 1 or 2");
            e->duplex_out = value;
            return FE_OK;
        case FE_TUNE_DUPLEX_MIN_MB:
            if (value < 0 || value > 4096) return fail(FE_ERR_PARAM, "threshold must be 0 .. 4096 MB");
            e->duplex_min_mb = value;
            return FE_OK;
        case FE_TUNE_DUPLEX_CHUNK_MB:
            if (value < 0 || value > 1024) return fail(FE_ERR_PARAM, "chunk size must be 0 .. 1024 MB");
            e->duplex_chunk_mb = value;
            return FE_OK;
        case FE_TUNE_LANES:
            if (value < 0 || value > 2) return fail(FE_ERR_PARAM, "lanes must be 0 (automatic), 1 or 2");
            e->tuning_single_lane = value == 1;
            return FE_OK;
        case FE_TUNE_FAIL_NEXT:
            e->fail_round_in = value < 0 ? -1 : value;
            return FE_OK;
        case FE_TUNE_DUPLEX_CAP_MB:
            if (value < 0 || value > (1 << 20)) return fail(FE_ERR_PARAM, "cap must be 0 .. 1048576 MB");
            e->duplex_cap_mb = value;
            return FE_OK;
        case FE_TUNE_SPLIT:
            if (value < 0 || value > 8) return fail(FE_ERR_PARAM, "split must be 0 (automatic), 1 (off) or 2 .. 8 time tiles");
            e->split_tiles = value;
            return FE_OK;
        default:
            return fail(FE_ERR_PARAM, "unknown tuning knob %d", knob);
    }
}

int fe_debug_xlane(fe_engine* e, float* out512) {
    if (!e || !out512) return fail(FE_ERR_PARAM, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    DevTmp buf;
    HIP_TRY(hipMalloc(&buf.p, 512 * sizeof(float)));
    HIP_TRY(fk::launch_xlane_selftest(static_cast<float*>(buf.p), e->stream));
    HIP_TRY(hipMemcpyAsync(out512, buf.p, 512 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return FE_OK;
}

// A small round trip through the engine: a kernel on its stream and its result back on the host.  What the host's
// GPU sharder asks a GPU that failed before it sends files there again (host/device_router.cpp).
int fe_engine_probe(fe_engine* e) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->fail_round_in < 0) return fail(FE_ERR_DEVICE, "injected device failure (test hook)");
    HIP_TRY(hipSetDevice(e->device));
    DevTmp buf;
    HIP_TRY(hipMalloc(&buf.p, 512 * sizeof(float)));
    float out[512];
    HIP_TRY(fk::launch_xlane_selftest(static_cast<float*>(buf.p), e->stream));
    HIP_TRY(hipMemcpyAsync(out, buf.p, sizeof(out), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return FE_OK;
}

// ---- measurement hooks ------------------------------------------------------
int fe_engine_set_profiling(fe_engine* e, int on) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (on == 2 && !e->kev[0][0][0]) {             // the event ring of mode 2, on first use
        HIP_TRY(hipSetDevice(e->device));
        for (auto& set : e->kev)
            for (auto& role : set)
                for (hipEvent_t& ev : role) HIP_TRY(hipEventCreate(&ev));
    }
    if (e->profiling == 2 && on != 2)
        for (int i = 0; i < fe_engine::kKevSets; ++i) { int rc = harvest_kernel_events(e, i); if (rc) return rc; }
    e->profiling = on == 2 ? 2 : on != 0;
    return FE_OK;
}

int fe_engine_last_kernels(fe_engine* e, char* forward, char* mac, char* inverse, size_t cap) {
    if (!e || cap == 0) return fail(FE_ERR_PARAM, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    char* dst[FE_K_COUNT] = {forward, mac, inverse};
    for (int k = 0; k < FE_K_COUNT; ++k)
        if (dst[k]) snprintf(dst[k], cap, "%s", e->last_names.k[k]);
    return FE_OK;
}

int fe_engine_get_profile(fe_engine* e, long long launches[FE_K_COUNT], double ms[FE_K_COUNT]) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    for (int k = 0; k < FE_K_COUNT; ++k) {
        if (launches) launches[k] = e->prof_launches[k];
        if (ms) ms[k] = e->prof_ms[k];
    }
    return FE_OK;
}

int fe_engine_get_kernel_profile(fe_engine* e, long long launches[FE_K_COUNT], double ms[FE_K_COUNT]) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    for (int i = 0; i < fe_engine::kKevSets; ++i) { int rc = harvest_kernel_events(e, i); if (rc) return rc; }
    for (int k = 0; k < FE_K_COUNT; ++k) {
        if (launches) launches[k] = e->prof_kernel_launches[k];
        if (ms) ms[k] = e->prof_kernel_ms[k];
    }
    return FE_OK;
}

static int hbm_rates_n(fe_engine* e, size_t bytes, int reps, double* gbs, int modes);
int fe_engine_hbm_rates(fe_engine* e, size_t bytes, int reps, double gbs[3]) { return hbm_rates_n(e, bytes, reps, gbs, 3); }
int fe_engine_hbm_rates2(fe_engine* e, size_t bytes, int reps, double gbs[5]) { return hbm_rates_n(e, bytes, reps, gbs, 5); }
int fe_engine_hbm_rates3(fe_engine* e, size_t bytes, int reps, double gbs[6]) { return hbm_rates_n(e, bytes, reps, gbs, 6); }
static int hbm_rates_n(fe_engine* e, size_t bytes, int reps, double* gbs, int modes) {
    if (!e || !gbs || reps < 1 || bytes < ((size_t)1 << 20)) return fail(FE_ERR_PARAM, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    bytes &= ~(size_t)15;
    DevTmp a, b;
    if (hipMalloc(&a.p, bytes) != hipSuccess || hipMalloc(&b.p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FE_ERR_ALLOC, "device allocation of 2 x %zu bytes failed", bytes);
    }
    HIP_TRY(hipMemsetAsync(a.p, 0x3c, bytes, e->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    hipError_t rc = hipEventCreate(&e1);
    for (int mode = 0; mode < modes && rc == hipSuccess; ++mode) {
        for (int w = 0; w < 2 && rc == hipSuccess; ++w) rc = fk::launch_hbm_probe(mode, a.p, b.p, bytes, e->stream);
        if (rc == hipSuccess) rc = hipEventRecord(e0, e->stream);
        for (int r = 0; r < reps && rc == hipSuccess; ++r) rc = fk::launch_hbm_probe(mode, a.p, b.p, bytes, e->stream);
        if (rc == hipSuccess) rc = hipEventRecord(e1, e->stream);
        if (rc == hipSuccess) rc = hipEventSynchronize(e1);
        float ms = 0.f;
        if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, e0, e1);
        if (rc == hipSuccess) gbs[mode] = (mode == 2 || mode == 4 || mode == 5 ? 2.0 : 1.0) * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc != hipSuccess) return fail(FE_ERR_DEVICE, "HBM probe: %s", hipGetErrorString(rc));
    return FE_OK;
}

int fe_engine_reset_profile(fe_engine* e) {
    if (!e) return fail(FE_ERR_PARAM, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    for (int i = 0; i < fe_engine::kKevSets; ++i) { int rc = harvest_kernel_events(e, i); if (rc) return rc; }
    for (int k = 0; k < FE_K_COUNT; ++k) {
        e->prof_launches[k] = 0; e->prof_ms[k] = 0.0;
        e->prof_kernel_launches[k] = 0; e->prof_kernel_ms[k] = 0.0;
    }
    return FE_OK;
}

}  // extern "C"
