// kernels.h — launch interface between the host engine and the gfx950 kernels.
//
// Data layout in HBM (all little-endian float32 / complex64 = float2):
//   spectrum row  : P complex bins of a 2P-point real FFT, bin 0 packed as
//                   (DC, Nyquist) — both are real — so a row is exactly P*8 B.
//   filter  G     : [data path][K + 1][P]   G(j) = s*H(j) + H(j-1), H scaled by 1/(2P)
//   stream  FDL   : [input channel][ring slots][P]    spectra Z(n) = FFT([x(n) | 0]) (delay line)
//   batch   Y     : [unit = (stream, output, block)][P] accumulated spectra
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fk {

// One stream's share of a batched call.  Device-resident array, one per stream.
struct StreamJob {
    const float* in;        // interleaved [nframes][cin]   (device)
    float* out;             // interleaved [nframes][cout]  (device)
    float2* fdl;            // [cin][ring][P]
    unsigned int* peaks;    // [2] float bits: max(0, signed max), max |.|
    unsigned int* blk_peaks;// NULL, or [nblocks][2]: the same two maxima per block of this call (zeroed by the host; K3 raises them)
    long long nframes;      // frames valid in this call (last block may be short)
    int nblocks;            // ceil(nframes / P)
    int slot0;              // ring slot of this call's first block
    int yunit0;             // first Y row of this stream in the batch
    int ring;               // FDL ring slots per input channel (>= K - 1 + blocks per call)
};

// One (input -> output) convolution path of a filter, as the MAC kernel sees it.
struct PathEntry {
    int in_ch;              // input channel feeding this output
    int data;               // index of the spectra set (links share one)
};

struct FilterDev {
    int cin, cout;
    int P, log2P;
    int K;                  // rows of G per data path = partitions + 1
    const float2* H;        // G: [ndata][K][P]
    const uint64_t* mask;   // [ndata][4] populated-row bitmap of G (K <= 129: words 0..2)
    const PathEntry* paths; // grouped by output channel
    const int* out_first;   // [cout + 1] prefix into paths
    const float2* tw;       // exp(-2*pi*i*k/(2P)), k in [0, 2P)   (real-FFT split / fold)
    const float2* twa;      // stage A rows k1 = 1,2,4: exp(-2*pi*i*n2*k1/P)  (fft_core.hpp WaveGeom)
    const float2* twb;      // stage B (per-wavefront N2-point FFT) pass tables
    const float2* twa2;     // the same two for the 2P-point transform of the stereo kernels (NULL if P < 512 or P = 8192)
    const float2* twb2;
};

// Which kernel each of the three launches of a round was (the instantiation as rocprofv3 prints it, without namespace and
// arguments): filled by the launchers when Tuning::names is set (fe_engine_last_kernels; bench.py matches its committed
// profiles against these).
struct LaunchNames { char k[3][96]; };

// Launch-shape choices of one engine (fe_engine_set_tuning; 0 = automatic everywhere).  The
// automatic choice depends on the batch shape, so tests pin a form to reach it with small batches.
struct Tuning {
    int fwd_run = 0;        // K1 walker: consecutive blocks per workgroup
    int inv_run = 0;        // K3 walker: consecutive blocks per workgroup
    int mac_form = 0;       // K2: 1 general, 4 / 8 / 16 sliding window of that many outputs, 100 whole-call walk
    int fft_form = 0;       // K1/K3: 1 general kernels only, 2 walkers whenever the shape allows (also small launches), 3 no channel-pair walkers (many channels: the one-block-per-workgroup pair kernels)
    int walk_lpb = 0;       // K2 whole-call walk: lanes per bin (1, 2, 4) instead of the automatic choice
    int walk_tiles = 0;     // K2 whole-call walk: time tiles per call
    int walk_fma = 0;       // K2 whole-call walk: 3 = the three-FMA form (mac_walk3.hip), 4 = the four-FMA form, 0 = by shape
    int walk_nt = 0;        // K2 three-FMA walk on one lane per bin: rows with the non-temporal hint — 0 by the launch's bytes, 1 never, 2 always
    LaunchNames* names = nullptr;   // set per call: where the launchers note the kernels they chose
    hipEvent_t (*kev)[2] = nullptr; // set per call while profiling: per role (0 = K1, 1 = K2, 2 = K3) a start / stop event to bind to the dispatch
    // set per call: the only descriptor of a one-stream launch, readable by the HOST.  Every kernel then receives it
    // by value among its arguments instead of fetching jobs[0] — a dependent read over the bus when the descriptors
    // sit in page-locked memory (2 us at the start of each of the three latency kernels), an upload in front of K1
    // when they sit in device memory (9.5 us per multi-block call of one stream)
    const struct StreamJob* one_job = nullptr;
    bool host_io = false;   // set per call: PCM in and out are page-locked HOST memory (zero-copy single-block path)
    int max_ring = 0;       // set per call: the longest FDL ring among the launch's streams (K2's per-lane offsets)
    bool in_resident = false;   // ... but the input has been copied into device memory already (big batches: DMA in, kernels write out)
};

// What the filter's populated-row bitmaps allow K2 to assume (computed once at commit).
struct MacShape {
    bool single_path = false;   // every output has exactly one input path
    int max_paths = 0;          // most input paths any output has
    int ndata = 0;              // spectra sets of the filter
    bool dense = false;         // most rows of G are populated (skipping rows would save < 40 %)
};

// K1: PCM -> spectra.  grid (max blocks, cin, jobs)
// pairs_ok: every stream's PCM pointer is 16-byte aligned (stereo frames are loaded as quads)
hipError_t launch_forward(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, bool pairs_ok,
                          const Tuning& tn, hipStream_t st);
// K2: Y = sum over paths and partitions of X * H.  time_tile: most blocks any stream carries
hipError_t launch_mac(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, float2* Y,
                      int time_tile, const MacShape& shape, const Tuning& tn, hipStream_t st);
// K3: spectra -> PCM (last P of each 2P window) + peaks.  grid (max blocks, cout, jobs)
hipError_t launch_inverse(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, const float2* Y,
                          bool walker_ok, const Tuning& tn, hipStream_t st);
// Device self-test of the cross-lane exchange fft_core.hpp relies on: out[0..63] = lane ids after
// xlane_swap32 on (lane, 100 + lane) first operand, out[64..127] second operand, out[128..255]
// the same for xlane_swap16, out[256..511] the four registers after xlane_transpose4 of (10*i + lane/16).
hipError_t launch_xlane_selftest(float* out512, hipStream_t st);
// Measurement hook (bench.py): plain streaming kernels over `bytes` of `a` (and `b`), 16 bytes per lane —
// mode 0 read a, 1 write b, 2 copy a -> b, 3 / 4 write / copy with every workgroup in its own region, 5 the best copy shape found
// (non-temporal both ways, 4 KiB bursts per wave).  What this GPU's HBM delivers to ANY kernel, read and written apart.
hipError_t launch_hbm_probe(int mode, const void* a, void* b, size_t bytes, hipStream_t st);
struct FftTables { const float2* tw; const float2* twa; const float2* twb; const float2* twa2; const float2* twb2; };
// K0: time-domain taps [ndata][K*P] -> Htmp [ndata][K][P] (scaled by 1/(2P)) -> G [ndata][K+1][P].
hipError_t launch_filter_transform(const float* taps, float2* Htmp, float2* Gs, int ndata, int K, int log2P,
                                   const FftTables& t, hipStream_t st);
// Host-side description of the twiddle buffer of a P-point engine.
int fft_table_count(int log2P);
void fill_fft_tables(int log2P, float2* dst, int off[4]);   // dst[fft_table_count(log2P)]; off: twa, twb, twa2, twb2

}  // namespace fk
