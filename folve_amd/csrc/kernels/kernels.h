// kernels.h — launch interface between the host engine and the gfx950 kernels.
//
// Data layout in HBM (all little-endian float32 / complex64 = float2):
//   spectrum row  : P complex bins of a 2P-point real FFT, bin 0 packed as
//                   (DC, Nyquist) — both are real — so a row is exactly P*8 B.
//   filter  H     : [data path][K partitions][P]      scaled by 1/(2P)
//   stream  FDL   : [input channel][ring slots][P]    frequency-domain delay line
//   stream  tail  : [2][input channel][P] float       previous input block
//                   (overlap-save window), ping-pong by call parity
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
    const float* tail_rd;   // [cin][P] input block preceding this call
    float* tail_wr;         // [cin][P] receives the last block of this call
    unsigned int* peaks;    // [2] float bits: max(0, signed max), max |.|
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
    int K;                  // partitions (ceil(size / P))
    const float2* H;        // [ndata][K][P]
    const uint32_t* mask;   // [ndata][4] populated-partition bitmap (K <= 128)
    const PathEntry* paths; // grouped by output channel
    const int* out_first;   // [cout + 1] prefix into paths
    const float2* tw;       // exp(-2*pi*i*k/(2P)), k in [0, 2P)   (real-FFT split / fold)
    const float2* twa;      // stage A rows k1 = 1,2,4: exp(-2*pi*i*n2*k1/P)  (fft_core.hpp WaveGeom)
    const float2* twb;      // stage B (per-wavefront N2-point FFT) pass tables
};

// K1: PCM -> spectra.  grid (max blocks, cin, jobs)
// walker_ok: every stream's PCM pointer is 16-byte aligned (lets mono / stereo streams take the
// pair-walker kernel); any_partial: some stream ends in a short block.
hipError_t launch_forward(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, bool walker_ok,
                          bool any_partial, hipStream_t st);
// K2: Y = sum over paths and partitions of X * H.  time_tile: outputs per thread (1,2,4,8,16)
hipError_t launch_mac(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, float2* Y,
                      int time_tile, hipStream_t st);
// K3: spectra -> PCM (last P of each 2P window) + peaks.  grid (max blocks, cout, jobs)
hipError_t launch_inverse(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, const float2* Y,
                          bool walker_ok, hipStream_t st);
struct FftTables { const float2* tw; const float2* twa; const float2* twb; };
// K0: time-domain taps [ndata][K*P] -> H [ndata][K][P] (scaled by 1/(2P)).
hipError_t launch_filter_transform(const float* taps, float2* H, int ndata, int K, int log2P, const FftTables& t,
                                   hipStream_t st);
// Host-side description of the twiddle buffer of a P-point engine.
int fft_table_count(int log2P);
void fill_fft_tables(int log2P, float2* dst, int* off_twa, int* off_twb);   // dst[fft_table_count(log2P)]

}  // namespace fk
