// kernels.hip — hand-written gfx950 kernels of the folve convolution hot path.
//
// Replaces what `Convproc::process()` does behind SoundProcessor::Process
// (/root/reference/sound-processor.cc:98-127) and what `impdata_create` does at
// filter set-up (/root/reference/zita-config.cc:163): uniformly partitioned FFT
// convolution, partition = block = P = `fragm` frames, evaluated here as
// overlap-save so that every block of a call is independent.  The overlap-save
// spectrum X(n) = FFT([x(n-1) | x(n)]) is never formed: with Z(n) = FFT([x(n) | 0]),
// X(n) = Z(n-1) + s*Z(n), s_k = (-1)^k, hence
//     Y(n) = sum_j X(n-j) H(j) = sum_{j=0..K} Z(n-j) G(j),  G(j) = s*H(j) + H(j-1),
// so each PCM block is read and transformed exactly once and a stream carries no
// time-domain state — only its ring of spectra.
//   K1 forward : block [x(n) | 0] -> Z(n) -> FDL ring row: a real FFT through a P-point complex FFT
//                per channel; stereo at P = 8192 by a workgroup that walks consecutive blocks (PCM read
//                once as quads, prefetched), stereo at smaller P as ONE 2P-point complex FFT of
//                z = L + i*R with the spectra separated by symmetry
//   K2 mac     : Y(n) = sum_paths sum_{j<=K} Z(n-j) * G(j): per bin a FIR along time — one thread walks a whole
//                call with the filter row and a window of spectra in registers (every row read once); a
//                16-output sliding window, a streaming form and a one-block latency form beside it
//   K3 inverse : Y(n) -> P-point complex IFFT in LDS -> last P samples of the
//                window, interleaved PCM store, per-stream peak
//   K0 filter  : taps -> H spectra, 1/(2P) folded in (as zita folds 0.5/parsize), then G
#include "kernels.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "fft_core.hpp"
#include "walk_common.hpp"

namespace fk {

#ifdef FOLVE_PHASE_TRACE
// TRACE build only (make TRACE=1): time per phase in 10 ns ticks (s_memrealtime), summed over workgroups by thread 0.
__device__ unsigned long long g_phase[2][8];
#define PH_INIT() unsigned long long ph_t = wall_clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PH(i)                                                     \
    do {                                                          \
        __builtin_amdgcn_sched_barrier(0);                        \
        const unsigned long long ph_n = wall_clock64();              \
        ph_acc[i] += ph_n - ph_t;                                 \
        ph_t = ph_n;                                              \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)
#define PH_FLUSH(kid)                                             \
    do {                                                          \
        if (threadIdx.x == 0)                                     \
            for (int ph_i = 0; ph_i < 8; ++ph_i) atomicAdd(&g_phase[kid][ph_i], ph_acc[ph_i]); \
    } while (0)
// time stamps (10 ns ticks, id in the top byte) of workgroup 0's thread 0, appended to a ring: the
// timeline of the one-block call as the GPU sees it (tools/phase_trace_single.py)
__device__ unsigned long long g_stamp[256];
__device__ unsigned int g_stamp_n;
#define STAMP(id)                                                                                     \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)               \
            g_stamp[atomicAdd(&g_stamp_n, 1u) & 255u] = ((unsigned long long)(id) << 56) | (wall_clock64() & 0xffffffffffffffull); \
    } while (0)
#else
#define PH_INIT() do {} while (0)
#define PH(i) do {} while (0)
#define PH_FLUSH(kid) do {} while (0)
#define STAMP(id) do {} while (0)
#endif

namespace {

// complex multiply-accumulate on two packed bins
__device__ __forceinline__ void cmac2(float4& acc, const float4& x, const float4& h) {
    acc.x = fmaf(x.x, h.x, acc.x); acc.x = fmaf(-x.y, h.y, acc.x);
    acc.y = fmaf(x.x, h.y, acc.y); acc.y = fmaf(x.y, h.x, acc.y);
    acc.z = fmaf(x.z, h.z, acc.z); acc.z = fmaf(-x.w, h.w, acc.z);
    acc.w = fmaf(x.z, h.w, acc.w); acc.w = fmaf(x.w, h.z, acc.w);
}

// populated-row bitmap of a data path: K + 1 <= 129 rows of G
__device__ __forceinline__ bool mask_bit(uint64_t lo, uint64_t hi, uint64_t top, int j) {
    return ((j < 64 ? lo >> j : j < 128 ? hi >> (j - 64) : top >> (j - 128)) & 1) != 0;
}

// Running maximum of a stream (non-negative floats order like their bit patterns).  Every wavefront of every block
// of a stream ends here, and atomics on ONE address serialise in L2 (12 - 16 ns each: a 256-block call of one
// 8-channel stream spent 0.4 ms in 32 768 of them).  A running maximum rises rarely, so look first — at L2, past the
// CU's own cache, which an atomic does not update — and leave the atomic to the few wavefronts that raise it.
__device__ __forceinline__ void peak_raise(unsigned int* p, float v) {
    const unsigned int bits = __float_as_uint(v);
    if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
}

// After stage B left Z (P-point FFT of z[m] = x[2m] + i x[2m+1]) in the LDS rows,
// form the 2P-point real spectrum and store it as a packed row.  The e^(-i*pi*k/P)
// factors are requested before the barrier that precedes this call (wsp).
template <int LOG2P>
struct SplitGeom {
    static constexpr int P = 1 << LOG2P;
    static constexpr int NT = WaveGeom<LOG2P>::NT;
    static constexpr int CNT = (P / 2 + NT - 1) / NT;
};

template <int LOG2P>
__device__ __forceinline__ void split_prefetch(float2 (&wsp)[SplitGeom<LOG2P>::CNT], const float2* __restrict__ tw,
                                               int tid) {
    using S = SplitGeom<LOG2P>;
#pragma unroll
    for (int c = 0; c < S::CNT; ++c) {
        const int k = tid + c * S::NT;
        wsp[c] = (k < S::P / 2) ? tw[k] : float2{1.f, 0.f};
    }
}

// Branch-free: bin k pairs with bin P - k; the thread that owns k = 0 (only in the first round)
// pairs the two self-paired bins instead — the packed (DC, Nyquist) bin 0 and bin P/2 — by
// selecting its second operand's address and its results, not by taking another path.
//   BATCH > 0: the scheduler may not interleave more than BATCH bin pairs (bounds the registers in flight)
template <int LOG2P, int BATCH = 0>
__device__ __forceinline__ void split_and_store(const float2* s, const float2 (&wsp)[SplitGeom<LOG2P>::CNT], int tid,
                                                float2* __restrict__ row, float scale) {
    using S = SplitGeom<LOG2P>;
    using G = WaveGeom<LOG2P>;
    constexpr int P = S::P;
    constexpr bool GUARD = (P / 2) % S::NT != 0;
#pragma unroll
    for (int c = 0; c < S::CNT; ++c) {
        const int k = tid + c * S::NT;
        if (GUARD && k >= P / 2) continue;
        const bool self = (c == 0) && (k == 0);
        const int k2 = self ? P / 2 : P - k;
        const float2 a = s[G::at(k)], b = s[G::at(k2)];
        const float2 e = float2{0.5f * (a.x + b.x), 0.5f * (a.y - b.y)};
        const float2 o = float2{0.5f * (a.y + b.y), -0.5f * (a.x - b.x)};
        const float2 t = cmul(o, wsp[c]);
        float2 r1 = float2{(e.x + t.x) * scale, (e.y + t.y) * scale};
        float2 r2 = float2{(e.x - t.x) * scale, -(e.y - t.y) * scale};
        if (c == 0) {
            r1 = self ? float2{(a.x + a.y) * scale, (a.x - a.y) * scale} : r1;   // (DC, Nyquist)
            r2 = self ? float2{b.x * scale, -b.y * scale} : r2;                   // bin P/2
        }
        // row is wave-uniform: the round's share of the address goes into the scalar base, and two
        // per-lane offsets (tid, NT - 1 - tid) serve all rounds
#ifdef FOLVE_K1_NT_STORE                                       // experiment: the spectrum rows leave with the non-temporal hint
#define FK_SPLIT_ST gst_u2_once
#else
#define FK_SPLIT_ST gst_u2
#endif
        if (c == 0) {
            FK_SPLIT_ST(row, (unsigned)k * 8u, r1);
            FK_SPLIT_ST(row, (unsigned)k2 * 8u, r2);
        } else {
            FK_SPLIT_ST(row + c * S::NT, (unsigned)tid * 8u, r1);
            FK_SPLIT_ST(row + (P - c * S::NT - (S::NT - 1)), (unsigned)(S::NT - 1 - tid) * 8u, r2);
        }
#undef FK_SPLIT_ST
        if constexpr (BATCH > 0) {
            // a compiler-level memory barrier: the next batch's LDS reads stay behind this batch's stores
            if ((c + 1) % BATCH == 0 && c + 1 < S::CNT) asm volatile("" ::: "memory");
        }
    }
}

// ---------------------------------------------------------------------------
// K1, general form: one workgroup per (block, input channel): Z(n) = FFT_2P([x(n) | 0]) of one
// channel through a P-point complex FFT.  grid (max blocks, input channels, streams)
// Any channel count, unaligned PCM, short last block, any P.
// ---------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT) void forward_kernel(JobRef jr,
                                                                      FilterDev f, int xl) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    // the stage-B pass tables ride into LDS beside the PCM loads (visible after the barrier that
    // follows stage A): read from global memory inside stage B they cost one cache round trip per
    // pass on the critical path of a lone transform
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    for (int i = threadIdx.x; i < G::TWB; i += G::NT) twb_l[i] = f.twb[i];
    const StreamJob job = fetch_job(jr);
    // xl: grid (8 * channels, blocks / 8, streams) — the channels of a block are dispatched together and, workgroup ids
    // going round the 8 XCDs, land on ONE XCD, so the strided reads of the same interleaved frames meet in its L2.
    // Calls of fewer than 8 blocks keep grid (blocks, channels, streams): that order would put them all on one XCD.
    const int b = xl ? blockIdx.y * 8 + (blockIdx.x & 7) : blockIdx.x;
    if (b >= job.nblocks) return;
    const int c = xl ? blockIdx.x >> 3 : blockIdx.y;
    const int tid = threadIdx.x;
    const int cin = f.cin;
    const long long f0 = (long long)b * P;               // first frame of the block
    const float* __restrict__ in = job.in;
    const bool wide2 = (cin == 2) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const bool wide1 = (cin == 1) && ((reinterpret_cast<uintptr_t>(in) & 7) == 0);

    auto load = [&](int m) -> float2 {                    // z[m] = (x[2m], x[2m+1]); the upper half is zero
        if (m >= P / 2) return float2{0.0f, 0.0f};
        float2 v;
        const long long fr = f0 + 2 * m;                  // even: the pair never straddles a block
        if (wide2 && fr + 1 < job.nframes) {              // stereo: one 16-byte load holds both frames
            const float4 q = gld(reinterpret_cast<const float4*>(in + fr * 2));
            v = (c == 0) ? float2{q.x, q.z} : float2{q.y, q.w};
        } else if (wide1 && fr + 1 < job.nframes) {       // mono: the pair is contiguous
            v = gld(reinterpret_cast<const float2*>(in + fr));
        } else {
            v.x = (fr < job.nframes) ? gld(in + fr * cin + c) : 0.0f;
            v.y = (fr + 1 < job.nframes) ? gld(in + (fr + 1) * cin + c) : 0.0f;
        }
        return v;
    };
    stage_a<LOG2P, false>(s, f.twa, tid, load);
    __syncthreads();
    stage_b<LOG2P, false>(s, twb_l, tid);
    float2 wsp[SplitGeom<LOG2P>::CNT];
    split_prefetch<LOG2P>(wsp, f.tw, tid);
    __syncthreads();
    const int slot = ring_slot(job.slot0, b, job.ring);
    float2* row = job.fdl + ((size_t)c * job.ring + slot) * P;
    split_and_store<LOG2P>(s, wsp, tid, row, 1.0f);
}

// ---------------------------------------------------------------------------
// K1, stereo form: ONE 2P-point complex FFT of z[t] = L[t] + i*R[t] (t < P, zero above) per block.
// grid (max blocks, 1, streams), P/8 threads (1024 at P = 8192: 16 wavefronts, one per LDS row).
// Interleaved stereo PCM *is* z: frames are loaded as plain (L, R) pairs, no deinterleave and no
// over-fetch; half of stage A's inputs are zero; the two real spectra come out of the symmetry
//     ZL[k] = (Z[k] + conj Z[2P-k]) / 2,   ZR[k] = (Z[k] - conj Z[2P-k]) / (2i).
// ---------------------------------------------------------------------------
// K1 for streams of four or more (an even number of) channels: one workgroup per (block, channel PAIR).  A frame of
// C interleaved channels holds the pair (2p, 2p+1) as 8 adjacent bytes: the workgroup loads those — half the load
// instructions of the per-channel kernel, each using 8 of every 32 bytes it touches instead of 4 — keeps both channels'
// samples in registers (only the lower half of z is data: 8 elements per thread and channel) and transforms one channel
// after the other in the same LDS image.  grid (8 * pairs, blocks / 8, streams), as forward_kernel.
template <int LOG2P>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT) void forward_chpair_kernel(JobRef jr, FilterDev f, int xl) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT, COLS = G::COLS;
    constexpr int H1 = (N1 + 1) / 2;                          // rows n1 < H1 hold data (m = n1*N2 + n2 < P/2); N1 == 1: guarded below
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    for (int i = threadIdx.x; i < G::TWB; i += NT) twb_l[i] = f.twb[i];
    const StreamJob job = fetch_job(jr);
    const int b = xl ? blockIdx.y * 8 + (blockIdx.x & 7) : blockIdx.x;
    if (b >= job.nblocks) return;
    const int c0 = (xl ? blockIdx.x >> 3 : blockIdx.y) * 2;
    const int tid = threadIdx.x;
    const int cin = f.cin;
    const long long f0 = (long long)b * P;
    const float* __restrict__ in = job.in + c0;
    float2 ev[COLS][H1], od[COLS][H1];                         // frames 2m and 2m + 1: (channel c0, channel c0 + 1)
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int n2 = tid + c * NT;
#pragma unroll
        for (int n1 = 0; n1 < H1; ++n1) {
            const int m = n1 * N2 + n2;
            const long long fr = f0 + 2 * m;
            const bool on = n2 < N2 && m < P / 2;
            ev[c][n1] = (on && fr < job.nframes) ? gld(reinterpret_cast<const float2*>(in + fr * cin)) : float2{0.f, 0.f};
            od[c][n1] = (on && fr + 1 < job.nframes) ? gld(reinterpret_cast<const float2*>(in + (fr + 1) * cin)) : float2{0.f, 0.f};
        }
    }
    const int slot = ring_slot(job.slot0, b, job.ring);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1) __syncthreads();                          // the first channel's split has read the image
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const int n2 = tid + c * NT;
            if ((N2 % NT) != 0 && n2 >= N2) continue;
            float2 v[N1];
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) {
                if (n1 < H1) v[n1] = h == 0 ? float2{ev[c][n1].x, od[c][n1].x} : float2{ev[c][n1].y, od[c][n1].y};
                else v[n1] = float2{0.f, 0.f};                // the zero padding of [x | 0]
            }
            stage_a_column<LOG2P, false>(s, f.twa, n2, v);
        }
        __syncthreads();
        stage_b<LOG2P, false>(s, twb_l, tid);
        float2 wsp[SplitGeom<LOG2P>::CNT];
        split_prefetch<LOG2P>(wsp, f.tw, tid);
        __syncthreads();
        float2* row = job.fdl + ((size_t)(c0 + h) * job.ring + slot) * P;
        split_and_store<LOG2P>(s, wsp, tid, row, 1.0f);
    }
}

template <int LOG2P>
__global__ __launch_bounds__(WaveGeom<LOG2P + 1>::NT) void forward_dual_kernel(JobRef jr,
                                                                               FilterDev f) {
    using G = WaveGeom<LOG2P + 1>;                        // geometry of the 2P-point transform
    constexpr int P = 1 << LOG2P, N = 2 * P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT, COLS = G::COLS;
    constexpr bool GUARD = (N2 % NT) != 0;
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;               // stage-B tables ride into LDS beside the PCM loads
    STAMP(1);
    for (int i = threadIdx.x; i < G::TWB; i += NT) twb_l[i] = f.twb2[i];
    const StreamJob job = fetch_job(jr);
    const int b = blockIdx.x;
    if (b >= job.nblocks) return;
    const int tid = threadIdx.x;
    const float2* __restrict__ pcm = reinterpret_cast<const float2*>(job.in) + (size_t)b * P;   // frame t: (L, R)
    const long long left = job.nframes - (long long)b * P;                                      // valid frames

    float2 v[COLS][N1];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int n2 = tid + c * NT;
        if (GUARD && n2 >= N2) continue;
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const int t = n1 * N2 + n2;
            const bool in_block = (N1 >= 2) ? (n1 < N1 / 2) : (t < P);
            v[c][n1] = (in_block && t < left) ? gld(pcm + t) : float2{0.0f, 0.0f};
        }
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int n2 = tid + c * NT;
        if (GUARD && n2 >= N2) continue;
        stage_a_column<LOG2P + 1, false>(s, f.twa2, n2, v[c]);
    }
    __syncthreads();
    stage_b<LOG2P + 1, false>(s, twb_l, tid);
    __syncthreads();
    const int slot = ring_slot(job.slot0, b, job.ring);
    float2* __restrict__ rowL = job.fdl + ((size_t)0 * job.ring + slot) * P;
    float2* __restrict__ rowR = job.fdl + ((size_t)1 * job.ring + slot) * P;
    constexpr int CNT = (P + NT - 1) / NT;
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int k = tid + c * NT;
        if (k >= P) continue;
        const float2 a = s[G::at(k)];
        if (k == 0) {
            const float2 ny = s[G::at(P)];
            gst(rowL, float2{a.x, ny.x});                 // packed (DC, Nyquist)
            gst(rowR, float2{a.y, ny.y});
        } else {
            const float2 bb = s[G::at(N - k)];
            gst(rowL + k, float2{0.5f * (a.x + bb.x), 0.5f * (a.y - bb.y)});
            gst(rowR + k, float2{0.5f * (a.y + bb.y), -0.5f * (a.x - bb.x)});
        }
    }
    STAMP(2);
}

// ---------------------------------------------------------------------------
// K1, latency form (P = 8192, stereo): the one-block call of SoundProcessor::Process.
// grid (blocks, 1, streams), 1024 threads = two halves of 512; half h transforms channel h in its own
// LDS image, both at once (the walker's workgroup does them one after the other: 14.7 us in the kernel
// for a lone stereo block; the one 2P-point transform of forward_dual_kernel 12.8 us).  Both halves
// request the block's (L0, R0, L1, R1) quads and keep their channel's two samples (fetching every quad
// once and sorting through LDS measured the same: the 64 KB take ~4.5 us over the bus either way).
// ---------------------------------------------------------------------------
template <int LOG2P, bool XL>
__global__ __launch_bounds__(2 * WaveGeom<LOG2P>::NT) void forward_pair_kernel(JobRef jr, FilterDev f) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT, COLS = G::COLS;
    static_assert(N1 == 8 && COLS == 2 && NT == 512, "pair form needs P = 8192");
    constexpr int HR = N1 / 2;
    __shared__ float2 s2[2][G::LDS_ELEMS + G::TWB];
    const int half = threadIdx.x / NT, tid = threadIdx.x - half * NT;   // half is wave-uniform
    float2* const s = s2[half];
    float2* const twb_l = s + G::LDS_ELEMS;
    STAMP(1);
    const StreamJob job = fetch_job(jr);      // one: the descriptor by value (Tuning::one_job)
    const int b = blockIdx.x;
    if (b >= job.nblocks) return;
    const float* __restrict__ in = job.in;
    const long long f0 = (long long)b * P;
    float2 x[COLS][HR];
    if (f0 + P <= job.nframes) {
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int h = 0; h < HR; ++h) {
                const float4 q = gld_u4_once(in + (size_t)f0 * 2 + (size_t)(h * N2 + c * NT) * 4, (unsigned)tid * 16u);
                x[c][h] = half ? float2{q.y, q.w} : float2{q.x, q.z};
            }
    } else {                                                  // a stream's short last block
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int h = 0; h < HR; ++h) {
                const long long fr = f0 + 2 * (h * N2 + tid + c * NT);
                float2 v{0.f, 0.f};
                if (fr < job.nframes) v.x = gld(in + fr * 2 + half);
                if (fr + 1 < job.nframes) v.y = gld(in + fr * 2 + 2 + half);
                x[c][h] = v;
            }
    }
    STAMP(10);                                                // descriptor read, PCM requested
    for (int i = tid; i < G::TWB; i += NT) twb_l[i] = f.twb[i];
    const float2 a1 = f.twa[0 * N2 + tid], a2 = f.twa[1 * N2 + tid], a4 = f.twa[2 * N2 + tid];
    const float2 w0 = f.tw[tid];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        float2 z[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) z[n1] = (n1 < HR) ? x[c][n1 < HR ? n1 : 0] : float2{0.0f, 0.0f};
        StageATw<LOG2P> tw;
        if (c == 0) {
            tw.w[0] = a1; tw.w[1] = a2; tw.w[2] = a4;
        } else {                                              // column tid + 512: W^(512*k1) = e^(-i*pi*k1/8)
            tw.w[0] = cmul_const(a1, kCos16[1], -kSin16[1]);
            tw.w[1] = cmul_const(a2, kCos16[2], -kSin16[2]);
            tw.w[2] = float2{a4.y, -a4.x};
        }
        tw.w[3] = float2{1.f, 0.f};
        stage_a_column<LOG2P, false>(s, tw, tid + c * NT, z);
    }
    STAMP(11);                                                // PCM arrived, stage A
    __syncthreads();
    STAMP(12);
    stage_b<LOG2P, false, XL>(s, twb_l, tid);
    STAMP(13);
    __syncthreads();
    STAMP(14);
    float2 wsp[SplitGeom<LOG2P>::CNT];                        // bins tid + 512*c: e^(-i*pi*512*c/P) = e^(-i*pi*c/16)
#pragma unroll
    for (int c = 0; c < SplitGeom<LOG2P>::CNT; ++c) wsp[c] = c ? cmul_const(w0, kCos32[c], -kSin32[c]) : w0;
    const int slot = ring_slot(job.slot0, b, job.ring);
    split_and_store<LOG2P>(s, wsp, tid, job.fdl + ((size_t)half * job.ring + slot) * P, 1.0f);
    STAMP(2);
}

// ---------------------------------------------------------------------------
// K1, stereo walker (P = 8192): the mirror image of the K3 walker.
// grid (runs of `run` consecutive blocks, 1, streams), 512 threads, TWO workgroups per CU.
//
// One workgroup owns a stream's channel pair and walks consecutive blocks.  A block's PCM is read
// once, as 16-byte (L0, R0, L1, R1) quads that hold z[m] = x[2m] + i*x[2m+1] of BOTH channels; the
// next block's quads are requested as soon as the second channel's stage A has consumed the
// current ones and fly during its stage B and split.  Everything loop-invariant costs 8 VGPRs: the
// stage-A twiddles of the thread's second column and the split twiddles of its eight bins are the
// table values of the FIRST column / bin rotated by compile-time constants (columns 512 apart,
// bins 512 apart), so one table read per kind serves the whole walk.  That keeps the kernel at
// <= 128 VGPRs: two workgroups (2 x 78 KB of LDS) share a CU, and one's memory phases overlap the
// other's LDS phases.  Every trip issues a fixed number of loads and stores (exact vmcnt waits).
//   XL: stage B's last two passes exchange through the cross-lane transpose (fft_core.hpp)
// ---------------------------------------------------------------------------
//   MC: the stream has four or more (an even number of) channels; the workgroup owns the pair (2p, 2p+1) and finds it as
//       8 adjacent bytes in every frame: two 8-byte loads per quad, plain ones (the other pairs' workgroups read the
//       same lines: they meet in the XCD's L2), grid (8 * pairs, runs / 8, streams) when xl (as forward_kernel), else
//       (runs, pairs, streams)
template <int LOG2P, bool XL, bool MC = false>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT, 4) void forward_walker_kernel(JobRef jr,
                                                                                FilterDev f, int run, int xl) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT, COLS = G::COLS;
    static_assert(N1 == 8 && COLS == 2 && NT == 512, "walker needs P = 8192");
    constexpr int HR = N1 / 2;                                // rows of a column that hold PCM (the rest is the zero padding)
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    PH_INIT();
    STAMP(1);
    const StreamJob job = fetch_job(jr);
    const int bx = (MC && xl) ? blockIdx.y * 8 + (blockIdx.x & 7) : blockIdx.x;
    const int c0 = !MC ? 0 : (xl ? blockIdx.x >> 3 : blockIdx.y) * 2;
    const int cin = MC ? f.cin : 2;
    const int b0 = bx * run;
    if (b0 >= job.nblocks) return;
    const int b1 = min(b0 + run, job.nblocks);
    const int tid = threadIdx.x;
    const float* __restrict__ in = job.in + c0;

    // stage A rows k1 = 1, 2, 4 at column tid; split twiddle e^(-i*pi*k/P) at k = tid
    const float2 a1_ = f.twa[0 * N2 + tid], a2_ = f.twa[1 * N2 + tid], a4_ = f.twa[2 * N2 + tid];
    const float2 w0_ = f.tw[tid];

    // quads of block b: frames (2m, 2m+1), m = h*N2 + n2 over the PCM rows of this thread's columns.
    // Registers across a stage B are what decides 128 VGPRs: the quads of a block are requested in
    // two halves (one column each), the first before the second channel's stage B, the second after
    // it, and the second channel's samples wait as pairs (16 VGPRs), not as quads.
    float4 q[COLS][HR];
    // this thread's frame pair inside a row of N2 pairs, in bytes
    auto lane_off = [&](int t) { return MC ? (unsigned)t * 8u * (unsigned)cin : (unsigned)t * 16u; };
    auto request_whole = [&](int c, const float* __restrict__ base, unsigned loff) {
#pragma unroll
        for (int h = 0; h < HR; ++h) {
            if constexpr (MC) {
                const float* __restrict__ r = base + (size_t)(h * N2 + c * NT) * 2 * cin;
                const float2 e = gld_u2(r, loff), o = gld_u2(r + cin, loff);
                q[c][h] = float4{e.x, e.y, o.x, o.y};
            } else {
                q[c][h] = gld_u4_once(base + (size_t)(h * N2 + c * NT) * 4, loff);
            }
        }
    };
    auto request_partial = [&](int b) {                       // a stream's short last block
        const long long f0 = (long long)b * P;
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int h = 0; h < HR; ++h) {
                const long long fr = f0 + 2 * (h * N2 + tid + c * NT);
                float4 v{0.f, 0.f, 0.f, 0.f};
                if (fr < job.nframes) { v.x = gld(in + fr * cin); v.y = gld(in + fr * cin + 1); }
                if (fr + 1 < job.nframes) { v.z = gld(in + (fr + 1) * cin); v.w = gld(in + (fr + 1) * cin + 1); }
                q[c][h] = v;
            }
    };
    const int bw = (int)min((long long)b1, max((long long)b0, job.nframes / P));   // blocks [b0, bw) are whole
    PH(7);                                                    // job descriptor
    if (b0 < bw) { request_whole(0, in + (size_t)b0 * P * cin, lane_off(tid)); request_whole(1, in + (size_t)b0 * P * cin, lane_off(tid)); }
    else request_partial(b0);
    // the stage-B pass tables into LDS behind the first block's requests (cache hits that ride in with the PCM instead of
    // ahead of it: a two-block walk of a lone stream is short enough to notice); first read after stage A's barrier
    for (int i = threadIdx.x; i < G::TWB; i += NT) twb_l[i] = f.twb[i];
    PH(6);                                                    // tables into LDS

    // Opaque copies, taken inside the loop: everything derived from the four table values (the
    // rotated twiddles, the products W^3, W^5 ..) is then recomputed per transform — a few dozen VALU
    // — instead of hoisted out of the loop into ~50 loop-invariant registers.
    auto opaque = [](float2 v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); return v; };
    auto stage_a_ch = [&](const float2 (&x)[COLS][HR], int t) {
        const float2 a1 = opaque(a1_), a2 = opaque(a2_), a4 = opaque(a4_);
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            float2 z[N1];
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) z[n1] = (n1 < HR) ? x[c][n1 < HR ? n1 : 0] : float2{0.0f, 0.0f};
            StageATw<LOG2P> tw;
            if (c == 0) {
                tw.w[0] = a1; tw.w[1] = a2; tw.w[2] = a4;
            } else {                                          // column tid + 512: W^(512*k1) = e^(-i*pi*k1/8)
                tw.w[0] = cmul_const(a1, kCos16[1], -kSin16[1]);
                tw.w[1] = cmul_const(a2, kCos16[2], -kSin16[2]);
                tw.w[2] = float2{a4.y, -a4.x};
            }
            tw.w[3] = float2{1.f, 0.f};
            stage_a_column<LOG2P, false>(s, tw, t + c * NT, z);
        }
    };
    auto split = [&](int b, int ch, int t) {                  // real-FFT split, FDL row store
        float2 wsp[SplitGeom<LOG2P>::CNT];                    // bins t + 512*c: e^(-i*pi*512*c/P) = e^(-i*pi*c/16)
        const float2 w0 = opaque(w0_);
#pragma unroll
        for (int c = 0; c < SplitGeom<LOG2P>::CNT; ++c) wsp[c] = c ? cmul_const(w0, kCos32[c], -kSin32[c]) : w0;
        const int slot = ring_slot(job.slot0, b, job.ring);
        split_and_store<LOG2P>(s, wsp, t, job.fdl + ((size_t)(c0 + ch) * job.ring + slot) * P, 1.0f);
    };
    // One block, both channels.  NEXT: the block `next` is requested on the way.
    auto do_block = [&]<bool NEXT>(std::bool_constant<NEXT>, int b, const float* __restrict__ next, int t, unsigned loff) {
        float2 x[COLS][HR], x1[COLS][HR];
#pragma unroll
        for (int c = 0; c < COLS; ++c)
#pragma unroll
            for (int h = 0; h < HR; ++h) {
                x[c][h] = float2{q[c][h].x, q[c][h].z};
                x1[c][h] = float2{q[c][h].y, q[c][h].w};
            }
        stage_a_ch(x, t);
        PH(0);                                                // PCM wait + stage A
        __syncthreads();
        PH(1);
        stage_b<LOG2P, false, XL>(s, twb_l, t);
        PH(2);
        __syncthreads();
        PH(3);
        split(b, 0, t);
        PH(4);
        __syncthreads();                                      // the image is rewritten by the next stage A
        PH(5);
        stage_a_ch(x1, t);
        __builtin_amdgcn_sched_barrier(0);                    // the loads are not hoisted into stage A (registers)
        if constexpr (NEXT) request_whole(0, next, loff);               // flies during this channel's stage B and split
        PH(0);
        __syncthreads();
        PH(1);
        stage_b<LOG2P, false, XL>(s, twb_l, t);
        PH(2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) request_whole(1, next, loff);               // flies during the split
        __syncthreads();
        PH(3);
        split(b, 1, t);
        PH(4);
        __syncthreads();
        PH(5);
    };

#pragma unroll 1
    for (int b = b0; b < bw; ++b) {
        int t = tid;
        asm volatile("" : "+v"(t));                           // keeps the address arithmetic inside the loop
        // Every trip issues the same number of memory operations.  After the walk's last block the loads go to the first
        // frames of this block with a lane offset of zero — a handful of lines, the values never used — not over the
        // whole block again: a short walk (two blocks of a lone many-channel stream) would read half its PCM twice.
        const bool more = b + 1 < bw;
        do_block(std::true_type{}, b, in + (size_t)(more ? b + 1 : b) * P * cin, t, more ? lane_off(t) : 0u);
    }
    if (bw < b1) {
        if (bw > b0) request_partial(bw);
        do_block(std::false_type{}, bw, nullptr, tid, 0u);
    }
    PH_FLUSH(0);
    STAMP(2);
}

// ---------------------------------------------------------------------------
// K0: filter partitions -> spectra.  grid (K, data paths)
// ---------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT) void filter_kernel(const float* __restrict__ taps,
                                                                     float2* __restrict__ H, int K,
                                                                     const float2* __restrict__ tw,
                                                                     const float2* __restrict__ twa,
                                                                     const float2* __restrict__ twb) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    for (int i = threadIdx.x; i < G::TWB; i += G::NT) twb_l[i] = twb[i];
    const int j = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const float2* __restrict__ part = reinterpret_cast<const float2*>(taps + ((size_t)d * K + j) * P);
    auto load = [&](int m) -> float2 { return (m < P / 2) ? part[m] : float2{0.0f, 0.0f}; };   // [h_j | 0]
    stage_a<LOG2P, false>(s, twa, tid, load);
    __syncthreads();
    stage_b<LOG2P, false>(s, twb_l, tid);
    float2 wsp[SplitGeom<LOG2P>::CNT];
    split_prefetch<LOG2P>(wsp, tw, tid);
    __syncthreads();
    split_and_store<LOG2P>(s, wsp, tid, H + ((size_t)d * K + j) * P, 0.5f / (float)P);
}

// G(j) = s*H(j) + H(j-1), j = 0..K (H(-1) = H(K) = 0), s_k = (-1)^k; the packed bin 0 holds
// (DC, Nyquist) and s = +1 for both (P is even).  grid (K + 1, data paths)
__global__ __launch_bounds__(256) void make_g_kernel(const float2* __restrict__ H, float2* __restrict__ Gs, int K,
                                                     int P) {
    const int j = blockIdx.x, d = blockIdx.y;
    const float2* __restrict__ h = H + ((size_t)d * K + j) * P;
    const float2* __restrict__ hp = H + ((size_t)d * K + (j - 1)) * P;
    float2* __restrict__ g = Gs + ((size_t)d * (K + 1) + j) * P;
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        const float sgn = (k & 1) ? -1.0f : 1.0f;
        float2 acc{0.0f, 0.0f};
        if (j < K) { const float2 x = h[k]; acc = float2{sgn * x.x, (k == 0 ? 1.0f : sgn) * x.y}; }
        if (j >= 1) { const float2 x = hp[k]; acc.x += x.x; acc.y += x.y; }
        g[k] = acc;
    }
}

// ---------------------------------------------------------------------------
// K3: inverse.  grid (max blocks per stream, output channels, streams)
//
// The Hermitian fold Z[k] = E[k] + i O[k], E = Y[k] + conj Y[P-k],
// O = (Y[k] - conj Y[P-k]) e^(+i*pi*k/P) pairs k with P-k.  A thread owns stage-A
// columns in pairs (p, N2-p) — slot 0 owns the self-paired columns 0 and N2/2 —
// so both members of every pair sit in its own registers: no LDS pass, no
// second read of Y.
// ---------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT) void inverse_kernel(JobRef jr,
                                                                      FilterDev f,
                                                                      const float2* __restrict__ Y, int xl) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT;
    constexpr int SLOTS = (N2 / 2 + NT - 1) / NT;            // column pairs per thread
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    for (int i = threadIdx.x; i < G::TWB; i += NT) twb_l[i] = f.twb[i];
    const StreamJob job = fetch_job(jr);
    // xl: grid (8 * channels, blocks / 8, streams), as forward_kernel: the 4-byte stores of a block's channels into the
    // same interleaved frames meet in one XCD's L2 instead of reaching HBM as partial lines at different times
    const int b = xl ? blockIdx.y * 8 + (blockIdx.x & 7) : blockIdx.x;
    if (b >= job.nblocks) return;
    const int o = xl ? blockIdx.x >> 3 : blockIdx.y;
    const int tid = threadIdx.x;
    const int cout = f.cout;
    const float2* __restrict__ y = Y + ((size_t)job.yunit0 + (size_t)o * job.nblocks + b) * P;
    const float2* __restrict__ tw = f.tw;

    // ---- loads: the two columns of each slot, and the twiddles of the first ----
    float2 ya[SLOTS][N1], yb[SLOTS][N1], wa[SLOTS][N1];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int p = tid + q * NT;
        if (p < N2 / 2) {
            const int ca = p, cb = (p == 0) ? N2 / 2 : N2 - p;
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) {
                ya[q][n1] = y[n1 * N2 + ca];
                yb[q][n1] = y[n1 * N2 + cb];
                wa[q][n1] = tw[n1 * N2 + ca];                 // e^(-i*pi*k/P), k = n1*N2 + ca
            }
        }
    }
    // ---- fold in registers, stage A, rows to LDS ----
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int p = tid + q * NT;
        if (p < N2 / 2) {
            float2 za[N1], zb[N1];
            if (p != 0) {
                // k = n1*N2 + p  <->  P - k = (N1-1-n1)*N2 + (N2 - p);  e^(-i*pi*(P-k)/P) = -conj(e^(-i*pi*k/P))
#pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) {
                    const float2 a = ya[q][n1], bb = yb[q][N1 - 1 - n1];
                    const float2 e = cadd_conj(a, bb);
                    const float2 dd = csub_conj(a, bb);
                    const float2 oo = cmulc(dd, wa[q][n1]);
                    za[n1] = cadd_i(e, oo);
                    zb[N1 - 1 - n1] = conj_csub_i(e, oo);
                }
            } else {
                // column 0: k = n1*N2 <-> (N1-n1)*N2 (k = 0 is the packed (DC, Nyquist) bin);
                // column N2/2: k = n1*N2 + N2/2 <-> (N1-1-n1)*N2 + N2/2.  Both pair inside the column.
#pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) {
                    if (n1 == 0) {
                        const float2 y0 = ya[q][0];
                        za[0] = float2{y0.x + y0.y, y0.x - y0.y};
                    } else {
                        const float2 a = ya[q][n1], bb = ya[q][N1 - n1];
                        const float2 e = cadd_conj(a, bb);
                        const float2 dd = csub_conj(a, bb);
                        const float2 oo = cmulc(dd, wa[q][n1]);
                        za[n1] = cadd_i(e, oo);
                    }
                    const float2 a = yb[q][n1], bb = yb[q][N1 - 1 - n1];
                    const float2 e = cadd_conj(a, bb);
                    const float2 dd = csub_conj(a, bb);
                    const float2 oo = cmulc(dd, tw[n1 * N2 + N2 / 2]);
                    zb[n1] = cadd_i(e, oo);
                }
            }
            const int ca = p, cb = (p == 0) ? N2 / 2 : N2 - p;
            stage_a_column<LOG2P, true>(s, f.twa, ca, za);
            stage_a_column<LOG2P, true>(s, f.twa, cb, zb);
        }
    }
    __syncthreads();
    stage_b<LOG2P, true>(s, twb_l, tid);
    __syncthreads();

    // ---- transposed read: consecutive lanes take consecutive output frames ----
    float* __restrict__ out = job.out;
    const bool wide1 = (cout == 1) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
    const long long fb = (long long)b * P;
    float pk_s = 0.0f, pk_a = 0.0f;
    constexpr int OUTS = (P / 2 + NT - 1) / NT;
    // Interleaved output of two or more channels, where the geometry allows: FRAME-MAJOR (see inverse_walker_kernel) — of
    // every run of 2 * NT frames thread t takes frames t and t + NT, so that a store instruction's 64 four-byte pieces lie
    // in 64 consecutive frames instead of every other frame of 128: half the write requests.
    constexpr bool FM = (P / 2) % NT == 0 && NT % 2 == 0;
    const float* sf = reinterpret_cast<const float*>(s);
#pragma unroll
    for (int c = 0; c < OUTS; ++c) {
        const int q = P / 2 + tid + c * NT;                  // z[q] = (y[2q], y[2q+1]); overlap-save keeps q >= P/2
        if (q >= P) continue;
        if (FM && !wide1) {
            const int q0 = P / 2 + c * NT + (tid >> 1);
            const float z0 = sf[2 * G::at(q0) + (tid & 1)], z1 = sf[2 * G::at(q0 + NT / 2) + (tid & 1)];
            const long long fr = fb + 2 * c * NT + tid;
            if (fr < job.nframes) {
                gst(out + fr * cout + o, z0);
                pk_s = fmaxf(pk_s, z0);
                pk_a = fmaxf(pk_a, fabsf(z0));
            }
            if (fr + NT < job.nframes) {
                gst(out + (fr + NT) * cout + o, z1);
                pk_s = fmaxf(pk_s, z1);
                pk_a = fmaxf(pk_a, fabsf(z1));
            }
            continue;
        }
        const float2 z = s[G::at(q)];
        const long long fr = fb + 2 * q - P;
        if (wide1 && fr + 1 < job.nframes) {                 // mono: the pair is contiguous
            *reinterpret_cast<float2*>(out + fr) = z;
            pk_s = fmaxf(pk_s, fmaxf(z.x, z.y));
            pk_a = fmaxf(pk_a, fmaxf(fabsf(z.x), fabsf(z.y)));
            continue;
        }
        if (fr < job.nframes) {
            gst(out + fr * cout + o, z.x);
            pk_s = fmaxf(pk_s, z.x);
            pk_a = fmaxf(pk_a, fabsf(z.x));
        }
        if (fr + 1 < job.nframes) {
            gst(out + (fr + 1) * cout + o, z.y);
            pk_s = fmaxf(pk_s, z.y);
            pk_a = fmaxf(pk_a, fabsf(z.y));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pk_s = fmaxf(pk_s, __shfl_xor(pk_s, off, 64));
        pk_a = fmaxf(pk_a, __shfl_xor(pk_a, off, 64));
    }
    if ((tid & 63) == 0) {
        // non-negative floats order like their bit patterns
        peak_raise(job.peaks + 0, pk_s);
        peak_raise(job.peaks + 1, pk_a);
        if (job.blk_peaks) {                                   // per-block maxima: this workgroup holds one block (the launcher sees to it)
            atomicMax(job.blk_peaks + 2 * b + 0, __float_as_uint(pk_s));
            atomicMax(job.blk_peaks + 2 * b + 1, __float_as_uint(pk_a));
        }
    }
}

// K3 for streams of four or more (an even number of) output channels: one workgroup per (block, channel PAIR), the
// counterpart of forward_chpair_kernel.  The two outputs are transformed one after the other in the same LDS image; the
// first one's samples wait in registers, and every frame's pair leaves as one 8-byte store (half the store instructions
// of the per-channel kernel, 8 of every 32 bytes of a line instead of 4).  grid (8 * pairs, blocks / 8, streams).
template <int LOG2P>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT) void inverse_chpair_kernel(JobRef jr, FilterDev f,
                                                                             const float2* __restrict__ Y, int xl) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT;
    constexpr int SLOTS = (N2 / 2 + NT - 1) / NT;            // column pairs per thread
    constexpr int OUTS = (P / 2 + NT - 1) / NT;
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    for (int i = threadIdx.x; i < G::TWB; i += NT) twb_l[i] = f.twb[i];
    const StreamJob job = fetch_job(jr);
    const int b = xl ? blockIdx.y * 8 + (blockIdx.x & 7) : blockIdx.x;
    if (b >= job.nblocks) return;
    const int o0 = (xl ? blockIdx.x >> 3 : blockIdx.y) * 2;
    const int tid = threadIdx.x;
    const int cout = f.cout;
    const float2* __restrict__ tw = f.tw;
    float2 first[OUTS];                                       // output o0's samples of this thread's frames
    float* __restrict__ out = job.out + o0;
    const long long fb = (long long)b * P;
    float pk_s = 0.0f, pk_a = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float2* __restrict__ y = Y + ((size_t)job.yunit0 + (size_t)(o0 + h) * job.nblocks + b) * P;
        if (h == 1) __syncthreads();                          // output o0's samples have been read from the image
        // ---- loads: the two columns of each slot, and the twiddles of the first ----
        float2 ya[SLOTS][N1], yb[SLOTS][N1], wa[SLOTS][N1];
    #pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int p = tid + q * NT;
            if (p < N2 / 2) {
                const int ca = p, cb = (p == 0) ? N2 / 2 : N2 - p;
    #pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) {
                    ya[q][n1] = y[n1 * N2 + ca];
                    yb[q][n1] = y[n1 * N2 + cb];
                    wa[q][n1] = tw[n1 * N2 + ca];                 // e^(-i*pi*k/P), k = n1*N2 + ca
                }
            }
        }
        // ---- fold in registers, stage A, rows to LDS ----
    #pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int p = tid + q * NT;
            if (p < N2 / 2) {
                float2 za[N1], zb[N1];
                if (p != 0) {
                    // k = n1*N2 + p  <->  P - k = (N1-1-n1)*N2 + (N2 - p);  e^(-i*pi*(P-k)/P) = -conj(e^(-i*pi*k/P))
    #pragma unroll
                    for (int n1 = 0; n1 < N1; ++n1) {
                        const float2 a = ya[q][n1], bb = yb[q][N1 - 1 - n1];
                        const float2 e = cadd_conj(a, bb);
                        const float2 dd = csub_conj(a, bb);
                        const float2 oo = cmulc(dd, wa[q][n1]);
                        za[n1] = cadd_i(e, oo);
                        zb[N1 - 1 - n1] = conj_csub_i(e, oo);
                    }
                } else {
                    // column 0: k = n1*N2 <-> (N1-n1)*N2 (k = 0 is the packed (DC, Nyquist) bin);
                    // column N2/2: k = n1*N2 + N2/2 <-> (N1-1-n1)*N2 + N2/2.  Both pair inside the column.
    #pragma unroll
                    for (int n1 = 0; n1 < N1; ++n1) {
                        if (n1 == 0) {
                            const float2 y0 = ya[q][0];
                            za[0] = float2{y0.x + y0.y, y0.x - y0.y};
                        } else {
                            const float2 a = ya[q][n1], bb = ya[q][N1 - n1];
                            const float2 e = cadd_conj(a, bb);
                            const float2 dd = csub_conj(a, bb);
                            const float2 oo = cmulc(dd, wa[q][n1]);
                            za[n1] = cadd_i(e, oo);
                        }
                        const float2 a = yb[q][n1], bb = yb[q][N1 - 1 - n1];
                        const float2 e = cadd_conj(a, bb);
                        const float2 dd = csub_conj(a, bb);
                        const float2 oo = cmulc(dd, tw[n1 * N2 + N2 / 2]);
                        zb[n1] = cadd_i(e, oo);
                    }
                }
                const int ca = p, cb = (p == 0) ? N2 / 2 : N2 - p;
                stage_a_column<LOG2P, true>(s, f.twa, ca, za);
                stage_a_column<LOG2P, true>(s, f.twa, cb, zb);
            }
        }
        __syncthreads();
        stage_b<LOG2P, true>(s, twb_l, tid);
        __syncthreads();
        // ---- transposed read: consecutive lanes take consecutive output frames ----
        // FRAME-MAJOR where the geometry allows (see inverse_walker_kernel): of every run of 2 * NT frames thread t takes
        // frames t and t + NT — one float of element t / 2 and one of element t / 2 + NT / 2 of the run — so that a store
        // instruction's 64 eight-byte pieces lie in 64 consecutive frames: half the write requests of the element-major
        // order (frames 2t and 2t + 1), in which every instruction touches every other frame of 128.
        constexpr bool FM = (P / 2) % NT == 0 && NT % 2 == 0;
        const float* sf = reinterpret_cast<const float*>(s);
#pragma unroll
        for (int c = 0; c < OUTS; ++c) {
            const int q = P / 2 + tid + c * NT;              // z[q] = (y[2q], y[2q+1]); overlap-save keeps q >= P/2
            if (q >= P) continue;
            const int q0 = P / 2 + c * NT + (tid >> 1);
            const float2 z = FM ? float2{sf[2 * G::at(q0) + (tid & 1)], sf[2 * G::at(q0 + NT / 2) + (tid & 1)]} : s[G::at(q)];
            if (h == 0) { first[c] = z; continue; }
            const long long fr = FM ? fb + 2 * c * NT + tid : fb + 2 * q - P;
            const long long fr2 = FM ? fr + NT : fr + 1;
            const float2 e = float2{first[c].x, z.x}, o = float2{first[c].y, z.y};   // frames fr and fr2: (o0, o0 + 1)
            if (fr < job.nframes) {
                gst(reinterpret_cast<float2*>(out + fr * cout), e);
                pk_s = fmaxf(pk_s, fmaxf(e.x, e.y));
                pk_a = fmaxf(pk_a, fmaxf(fabsf(e.x), fabsf(e.y)));
            }
            if (fr2 < job.nframes) {
                gst(reinterpret_cast<float2*>(out + fr2 * cout), o);
                pk_s = fmaxf(pk_s, fmaxf(o.x, o.y));
                pk_a = fmaxf(pk_a, fmaxf(fabsf(o.x), fabsf(o.y)));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pk_s = fmaxf(pk_s, __shfl_xor(pk_s, off, 64));
        pk_a = fmaxf(pk_a, __shfl_xor(pk_a, off, 64));
    }
    if ((tid & 63) == 0) {
        peak_raise(job.peaks + 0, pk_s);
        peak_raise(job.peaks + 1, pk_a);
        if (job.blk_peaks) {                                   // per-block maxima: this workgroup holds one block (the launcher sees to it)
            atomicMax(job.blk_peaks + 2 * b + 0, __float_as_uint(pk_s));
            atomicMax(job.blk_peaks + 2 * b + 1, __float_as_uint(pk_a));
        }
    }
}

// e^(+i*pi*k/P) at k = n1*N2 + N2/2 for P = 8192 (N1 = 8): the fold twiddles of the self-paired middle
// column are the odd 32nd roots of unity — constants, so the walkers' loops hold no table load.
constexpr float kMidCos8[8] = {0.98078528040323043f, 0.83146961230254524f, 0.55557023301960229f, 0.19509032201612833f, -0.19509032201612819f, -0.55557023301960196f, -0.83146961230254535f, -0.98078528040323043f};
constexpr float kMidSin8[8] = {0.19509032201612825f, 0.55557023301960218f, 0.83146961230254524f, 0.98078528040323043f, 0.98078528040323043f, 0.83146961230254546f, 0.55557023301960218f, 0.19509032201612861f};

// ---------------------------------------------------------------------------
// K3, fast form ("pair-walker"): mono / stereo output, P = 8192.
// grid (runs of `run` consecutive blocks, 1, streams), 512 threads, TWO workgroups per CU.
//
// One workgroup owns the stream's output channel pair and walks `run` consecutive
// blocks.  The fold twiddles e^(-i*pi*k/P) of a thread's eight bins are 1024 bins apart:
// one table value rotated by the 16th roots of unity (constants), so they cost 2 VGPRs
// for the whole walk; the Y row of the next FFT is requested as soon as the fold has
// consumed the current one; the first channel's samples wait in registers for the
// second's, and the block leaves as whole (L0, R0, L1, R1) quads — 16 bytes per lane,
// full lines — instead of two workgroups interleaving 4-byte stores.  Held to 128 VGPRs:
// two workgroups share a CU and cover each other's barriers and memory waits.
// ---------------------------------------------------------------------------
//   MC (with COUT = 2): the stream has four or more (an even number of) outputs; the workgroup owns the pair
//       (2p, 2p+1) and every frame's pair leaves as one 8-byte store — plain stores: the pairs of a block meet in the
//       XCD's L2 and reach HBM as whole lines.  grid (8 * pairs, runs / 8, streams) when xl, else (runs, pairs, streams)
template <int LOG2P, int COUT, bool XL, bool MC = false>
__global__ __launch_bounds__(WaveGeom<LOG2P>::NT, 4) void inverse_walker_kernel(JobRef jr,
                                                                                FilterDev f,
                                                                                const float2* __restrict__ Y, int run, int xl) {
    static_assert(!MC || COUT == 2, "a channel pair");
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT;
    static_assert(N1 == 8 && NT == N2 / 2, "walker needs P = 8192 (one column pair per thread)");
    constexpr int OUTS = (P / 2) / NT;                        // output complex samples per thread (8)
    // LDS: the FFT image, then the stage-B pass tables: the loop waits on vmcnt only for the
    // prefetched Y row — vmcnt returns in order, so a table load inside the loop would also wait
    // for the prefetch and the previous block's stores.
    __shared__ float2 s[G::LDS_ELEMS + G::TWB];
    float2* const twb_l = s + G::LDS_ELEMS;
    PH_INIT();
    STAMP(5);
    const StreamJob job = fetch_job(jr);
    const int bx = (MC && xl) ? blockIdx.y * 8 + (blockIdx.x & 7) : blockIdx.x;
    const int o0 = !MC ? 0 : (xl ? blockIdx.x >> 3 : blockIdx.y) * 2;
    const int cout = MC ? f.cout : COUT;
    const int b0 = bx * run;
    if (b0 >= job.nblocks) return;
    const int b1 = min(b0 + run, job.nblocks);
    const int tid = threadIdx.x;
    const float2* __restrict__ tw = f.tw;
    // column pair of this thread: (p, N2 - p); thread 0 owns the self-paired columns 0 and N2/2
    const int ca = tid, cb = (tid == 0) ? N2 / 2 : N2 - tid;

    const float2 wb_ = tw[ca];                                // e^(-i*pi*k/P) at k = ca; k = n1*N2 + ca is a rotation by e^(-i*pi*n1/8)
    StageATw<LOG2P> atw_a_ = load_stage_a_tw<LOG2P>(f.twa, ca);  // loop-invariant: kept in registers (rows k1 = 1, 2, 4)
    StageATw<LOG2P> atw_b_ = load_stage_a_tw<LOG2P>(f.twa, cb);
    atw_a_.w[3] = atw_b_.w[3] = float2{1.f, 0.f};

    auto row_of = [&](int b, int o) { return Y + ((size_t)job.yunit0 + (size_t)(o0 + o) * job.nblocks + b) * P; };
    // The next Y row is requested in two halves (one column of the pair each): the first as soon as
    // the fold has consumed the current row — it flies during stages A and B —, the second after
    // stage B: 16 instead of 32 VGPRs in flight across the transform, which is what lets two
    // workgroups share a CU.
    float2 ya[N1], yb[N1];
    auto request_a = [&](const float2* __restrict__ y) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) ya[n1] = gld_u2_once(y + n1 * N2, (unsigned)ca * 8u);
    };
    auto request_b = [&](const float2* __restrict__ y) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) yb[n1] = gld_u2_once(y + n1 * N2, (unsigned)cb * 8u);
    };
    request_a(row_of(b0, 0));
    request_b(row_of(b0, 0));
    for (int i = threadIdx.x; i < G::TWB; i += NT) twb_l[i] = f.twb[i];   // behind the first row's requests (see forward_walker_kernel)
    float* __restrict__ out = job.out + o0;
    float pk_s = 0.0f, pk_a = 0.0f;
    PH(7);                                                    // tables into LDS, job descriptor, first Y row requested

    // One block, all of its channels.  WHOLE: every frame of the block exists — true for all
    // blocks but a stream's last.  The loop over whole blocks issues a FIXED number of loads and
    // stores per trip, so the wait for a prefetched row is an exact vmcnt(N) that leaves the
    // block's stores in flight; with a data-dependent store count hipcc falls back to vmcnt(0)
    // and every wave idles for a store round trip per block.
    auto do_block = [&]<bool WHOLE>(std::bool_constant<WHOLE>, int b) {
        float2 zl[OUTS];                                      // first channel's samples of the block
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            // Opaque copy of the thread index: keeps the (cheap) LDS / table address arithmetic of the
            // FFT inside the loop instead of hoisted into ~100 loop-invariant registers.
            int t = tid;
            asm volatile("" : "+v"(t));
            // ---- Hermitian fold in registers (see inverse_kernel) ----
            float2 za[N1], zb[N1];
            float2 wb = wb_;                                  // opaque: the rotated twiddles are recomputed, not hoisted
            asm volatile("" : "+v"(wb.x), "+v"(wb.y));
            StageATw<LOG2P> atw_a = atw_a_, atw_b = atw_b_;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                asm volatile("" : "+v"(atw_a.w[r].x), "+v"(atw_a.w[r].y));
                asm volatile("" : "+v"(atw_b.w[r].x), "+v"(atw_b.w[r].y));
            }
            if (t != 0) {
#pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) {
                    const float2 a = ya[n1], bb = yb[N1 - 1 - n1];
                    const float2 e = cadd_conj(a, bb);
                    const float2 dd = csub_conj(a, bb);
                    const float2 wa = n1 ? cmul_const(wb, kCos16[n1], -kSin16[n1]) : wb;
                    const float2 oo = cmulc(dd, wa);
                    za[n1] = cadd_i(e, oo);
                    zb[N1 - 1 - n1] = conj_csub_i(e, oo);
                }
            } else {
#pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) {
                    if (n1 == 0) {
                        const float2 y0 = ya[0];
                        za[0] = float2{y0.x + y0.y, y0.x - y0.y};
                    } else {
                        const float2 a = ya[n1], bb = ya[N1 - n1];
                        const float2 e = cadd_conj(a, bb);
                        const float2 dd = csub_conj(a, bb);
                        const float2 oo = cmul_const(dd, kCos16[n1], kSin16[n1]);   // thread 0: k = n1*N2, conj(e^(-i*pi*n1/8))
                        za[n1] = cadd_i(e, oo);
                    }
                    const float2 a = yb[n1], bb = yb[N1 - 1 - n1];
                    const float2 e = cadd_conj(a, bb);
                    const float2 dd = csub_conj(a, bb);
                    const float2 oo = cmul_const(dd, kMidCos8[n1], kMidSin8[n1]);
                    zb[n1] = cadd_i(e, oo);
                }
            }
            PH(0);                                            // Y row wait + fold
            // The next row flies during this FFT.  The loads are unconditional so that the trip's
            // memory-operation count is fixed; after the walk's last row they read the (cache-resident,
            // 2P-entry) twiddle table instead of a row, and the values are never used.
            const float2* __restrict__ nrow = (o + 1 < COUT) ? row_of(b, o + 1) : (b + 1 < b1 ? row_of(b + 1, 0) : tw);
            request_a(nrow);
            stage_a_column<LOG2P, true>(s, atw_a, t, za);
            stage_a_column<LOG2P, true>(s, atw_b, (t == 0) ? N2 / 2 : N2 - t, zb);
            PH(1);
            __syncthreads();
            PH(2);
            stage_b<LOG2P, true, XL, XL>(s, twb_l, t);              // XL: only the upper half of every row (the kept samples) is written
            request_b(nrow);
            PH(3);
            __syncthreads();
            PH(4);
            // ---- transposed read: consecutive lanes take consecutive output frames ----
            const long long fb = (long long)b * P;
            // This thread's two frames of the c-th run of 2 * NT frames: frames 2t and 2t + 1 (one complex element of the
            // image), or — a whole block of a many-channel stream — FRAME-MAJOR: thread t takes frames t and t + NT, so
            // that a store instruction's 64 eight-byte pieces lie in 64 CONSECUTIVE frames (2 KB, each 64-byte segment
            // written twice by neighbouring lanes) instead of every other frame of 128 (4 KB, every segment once per
            // instruction): half the write requests for the same bytes.  (Element q + 512 c + 256 j of the image lies
            // 68 c + 34 j elements behind element q — rows of 8, 16 + 1 padding: constant offsets for the LDS reads;
            // tests/host_fft_check.cpp holds the identity against WaveGeom<13>::at.)
            auto take = [&](int c) -> float2 {
                if constexpr (MC && WHOLE) {
                    const float* sf = reinterpret_cast<const float*>(s) + (2 * G::at(P / 2 + (t >> 1)) + (t & 1));
                    return float2{sf[2 * (68 * c)], sf[2 * (68 * c + 34)]};
                } else {
                    return s[G::at(P / 2 + t + c * NT)];
                }
            };
            if (COUT == 2 && o == 0) {
#pragma unroll
                for (int c = 0; c < OUTS; ++c) zl[c] = take(c);
            } else {
#pragma unroll
                for (int c = 0; c < OUTS; ++c) {
                    const int q = P / 2 + t + c * NT;         // z[q] = (y[2q], y[2q+1]); overlap-save keeps q >= P/2
                    const float2 z = take(c);
                    const long long fr = fb + 2 * q - P;
                    if constexpr (COUT == 2) {
                        const float2 l = zl[c];
                        if constexpr (WHOLE) {
                            if constexpr (MC) {
                                float* __restrict__ r = out + (size_t)(fb + 2 * c * NT) * cout;
                                const unsigned toff = (unsigned)t * 4u * (unsigned)cout;
#ifdef FOLVE_WHATIF_QUAD_STORES      // what-if (wrong output, same bytes, half the write requests): every other pair stores 16 bytes per frame for
                                     // itself and its neighbour — what a four-channel workgroup would buy: cfg4 K3 67 -> 62.5 us, 1 x 8 ch x 1 024 blocks 226 -> 198
                                     // (tools/build_variant.sh quadst -DFOLVE_WHATIF_QUAD_STORES; tools/mc_walker_probe.py)
                                if ((o0 & 2) == 0) {
                                    *(FK_GLOBAL v4f*)((FK_GLOBAL char*)r + toff) = v4f{l.x, z.x, l.x, z.x};
                                    *(FK_GLOBAL v4f*)((FK_GLOBAL char*)(r + NT * cout) + toff) = v4f{l.y, z.y, l.y, z.y};
                                }
#else
                                gst_u2(r, toff, float2{l.x, z.x});                      // frame fb + 2 c NT + t
                                gst_u2(r + NT * cout, toff, float2{l.y, z.y});          // and the one NT frames on
#endif
                            } else {
                                gst_u4_once(out + (fb + 2 * c * NT) * 2, (unsigned)t * 16u, float4{l.x, z.x, l.y, z.y});   // frame fb + 2q - P
                            }
                            pk_s = fmaxf(pk_s, fmaxf(fmaxf(l.x, l.y), fmaxf(z.x, z.y)));
                            pk_a = fmaxf(pk_a, fmaxf(fmaxf(fabsf(l.x), fabsf(l.y)), fmaxf(fabsf(z.x), fabsf(z.y))));
                        } else {
                            if (fr < job.nframes) {
                                gst(out + fr * cout, l.x); gst(out + fr * cout + 1, z.x);
                                pk_s = fmaxf(pk_s, fmaxf(l.x, z.x));
                                pk_a = fmaxf(pk_a, fmaxf(fabsf(l.x), fabsf(z.x)));
                            }
                            if (fr + 1 < job.nframes) {
                                gst(out + (fr + 1) * cout, l.y); gst(out + (fr + 1) * cout + 1, z.y);
                                pk_s = fmaxf(pk_s, fmaxf(l.y, z.y));
                                pk_a = fmaxf(pk_a, fmaxf(fabsf(l.y), fabsf(z.y)));
                            }
                        }
                    } else {
                        if constexpr (WHOLE) {
                            gst_u2(out + (fb + 2 * c * NT), (unsigned)t * 8u, z);
                            pk_s = fmaxf(pk_s, fmaxf(z.x, z.y));
                            pk_a = fmaxf(pk_a, fmaxf(fabsf(z.x), fabsf(z.y)));
                        } else {
                            if (fr < job.nframes) { gst(out + fr, z.x); pk_s = fmaxf(pk_s, z.x); pk_a = fmaxf(pk_a, fabsf(z.x)); }
                            if (fr + 1 < job.nframes) { gst(out + fr + 1, z.y); pk_s = fmaxf(pk_s, z.y); pk_a = fmaxf(pk_a, fabsf(z.y)); }
                        }
                    }
                }
            }
            PH(5);
            __syncthreads();                                  // the image is rewritten by the next stage A
            PH(6);
        }
    };

    const int bw = (int)min((long long)b1, max((long long)b0, job.nframes / P));   // blocks [b0, bw) are whole
#pragma unroll 1
    for (int b = b0; b < bw; ++b) do_block(std::true_type{}, b);
    if (bw < b1) do_block(std::false_type{}, bw);             // a stream's short last block
    PH_FLUSH(1);
    STAMP(6);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pk_s = fmaxf(pk_s, __shfl_xor(pk_s, off, 64));
        pk_a = fmaxf(pk_a, __shfl_xor(pk_a, off, 64));
    }
    if ((tid & 63) == 0) {
        peak_raise(job.peaks + 0, pk_s);
        peak_raise(job.peaks + 1, pk_a);
        if (job.blk_peaks) {                                   // per-block maxima: this workgroup holds one block (the launcher sees to it)
            atomicMax(job.blk_peaks + 2 * b0 + 0, __float_as_uint(pk_s));
            atomicMax(job.blk_peaks + 2 * b0 + 1, __float_as_uint(pk_a));
        }
    }
}

// ---------------------------------------------------------------------------
// K3, latency form (P = 8192, stereo): the one-block call of SoundProcessor::Process.
// grid (blocks, 1, streams), 1024 threads = two halves of 512; half h folds and transforms output h in
// its own LDS image, both at once (the walker's workgroup: one after the other, 11.4 us in the kernel
// for a lone stereo block).  When both images stand, every thread reads the L samples from one and the
// R samples from the other and stores whole (L0, R0, L1, R1) quads — half h the even / odd ones.
// ---------------------------------------------------------------------------
template <int LOG2P, bool XL>
__global__ __launch_bounds__(2 * WaveGeom<LOG2P>::NT) void inverse_pair_kernel(JobRef jr, FilterDev f,
                                                                               const float2* __restrict__ Y) {
    using G = WaveGeom<LOG2P>;
    constexpr int P = 1 << LOG2P;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT;
    static_assert(N1 == 8 && NT == N2 / 2, "pair form needs P = 8192 (one column pair per thread)");
    constexpr int OUTS = (P / 2) / NT;                        // output complex samples per thread and channel (8)
    __shared__ float2 s2[2][G::LDS_ELEMS + G::TWB];
    const int half = threadIdx.x / NT, t = threadIdx.x - half * NT;     // half is wave-uniform
    float2* const s = s2[half];
    float2* const twb_l = s + G::LDS_ELEMS;
    STAMP(5);
    const StreamJob job = fetch_job(jr);
    const int b = blockIdx.x;
    if (b >= job.nblocks) return;
    const int ca = t, cb = (t == 0) ? N2 / 2 : N2 - t;
    const float2* __restrict__ y = Y + ((size_t)job.yunit0 + (size_t)half * job.nblocks + b) * P;
    float2 ya[N1], yb[N1];
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) ya[n1] = gld_u2_once(y + n1 * N2, (unsigned)ca * 8u);
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) yb[n1] = gld_u2_once(y + n1 * N2, (unsigned)cb * 8u);
    for (int i = t; i < G::TWB; i += NT) twb_l[i] = f.twb[i];
    const float2 wb = f.tw[ca];                               // e^(-i*pi*k/P) at k = ca; k = n1*N2 + ca is a rotation by e^(-i*pi*n1/8)
    StageATw<LOG2P> atw_a = load_stage_a_tw<LOG2P>(f.twa, ca);
    StageATw<LOG2P> atw_b = load_stage_a_tw<LOG2P>(f.twa, cb);
    atw_a.w[3] = atw_b.w[3] = float2{1.f, 0.f};
    // ---- Hermitian fold in registers (see inverse_kernel / inverse_walker_kernel) ----
    float2 za[N1], zb[N1];
    if (t != 0) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const float2 a = ya[n1], bb = yb[N1 - 1 - n1];
            const float2 e = cadd_conj(a, bb);
            const float2 dd = csub_conj(a, bb);
            const float2 wa = n1 ? cmul_const(wb, kCos16[n1], -kSin16[n1]) : wb;
            const float2 oo = cmulc(dd, wa);
            za[n1] = cadd_i(e, oo);
            zb[N1 - 1 - n1] = conj_csub_i(e, oo);
        }
    } else {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            if (n1 == 0) {
                const float2 y0 = ya[0];
                za[0] = float2{y0.x + y0.y, y0.x - y0.y};
            } else {
                const float2 a = ya[n1], bb = ya[N1 - n1];
                const float2 e = cadd_conj(a, bb);
                const float2 dd = csub_conj(a, bb);
                const float2 oo = cmul_const(dd, kCos16[n1], kSin16[n1]);   // thread 0: k = n1*N2, conj(e^(-i*pi*n1/8))
                za[n1] = cadd_i(e, oo);
            }
            const float2 a = yb[n1], bb = yb[N1 - 1 - n1];
            const float2 e = cadd_conj(a, bb);
            const float2 dd = csub_conj(a, bb);
            const float2 oo = cmul_const(dd, kMidCos8[n1], kMidSin8[n1]);
            zb[n1] = cadd_i(e, oo);
        }
    }
    stage_a_column<LOG2P, true>(s, atw_a, t, za);
    stage_a_column<LOG2P, true>(s, atw_b, cb, zb);
    __syncthreads();
    stage_b<LOG2P, true, XL, XL>(s, twb_l, t);                // XL: only the upper half of every row (the kept samples) is written
    __syncthreads();
    // ---- both images stand: quads c = half, half + 2, .. of this thread's frames ----
    float* __restrict__ out = job.out;
    const long long fb = (long long)b * P;
    const bool whole = fb + P <= job.nframes;
    float pk_s = 0.0f, pk_a = 0.0f;
#pragma unroll
    for (int cc = 0; cc < OUTS / 2; ++cc) {
        const int c = 2 * cc + half;
        const int q = P / 2 + t + c * NT;                     // z[q] = (y[2q], y[2q+1]); overlap-save keeps q >= P/2
        const float2 l = s2[0][G::at(q)], r = s2[1][G::at(q)];
        const long long fr = fb + 2 * q - P;
        if (whole) {
            gst_u4_once(out + (fb + 2 * c * NT) * 2, (unsigned)t * 16u, float4{l.x, r.x, l.y, r.y});
            pk_s = fmaxf(pk_s, fmaxf(fmaxf(l.x, l.y), fmaxf(r.x, r.y)));
            pk_a = fmaxf(pk_a, fmaxf(fmaxf(fabsf(l.x), fabsf(l.y)), fmaxf(fabsf(r.x), fabsf(r.y))));
        } else {
            if (fr < job.nframes) {
                gst(out + fr * 2, l.x); gst(out + fr * 2 + 1, r.x);
                pk_s = fmaxf(pk_s, fmaxf(l.x, r.x));
                pk_a = fmaxf(pk_a, fmaxf(fabsf(l.x), fabsf(r.x)));
            }
            if (fr + 1 < job.nframes) {
                gst(out + fr * 2 + 2, l.y); gst(out + fr * 2 + 3, r.y);
                pk_s = fmaxf(pk_s, fmaxf(l.y, r.y));
                pk_a = fmaxf(pk_a, fmaxf(fabsf(l.y), fabsf(r.y)));
            }
        }
    }
    STAMP(6);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pk_s = fmaxf(pk_s, __shfl_xor(pk_s, off, 64));
        pk_a = fmaxf(pk_a, __shfl_xor(pk_a, off, 64));
    }
    if ((t & 63) == 0) {
        peak_raise(job.peaks + 0, pk_s);
        peak_raise(job.peaks + 1, pk_a);
        if (job.blk_peaks) {                                   // per-block maxima: this workgroup holds one block (the launcher sees to it)
            atomicMax(job.blk_peaks + 2 * b + 0, __float_as_uint(pk_s));
            atomicMax(job.blk_peaks + 2 * b + 1, __float_as_uint(pk_a));
        }
    }
}

// Packed bin 0 = (DC, Nyquist): two real products, not a complex one.  Done at the end by the
// whole workgroup that owns bin 0: thread t takes output block t % TT and every (threads / TT)-th
// partition, so each thread has one to three independent load pairs in flight; the groups are
// summed through LDS.  (As a serial loop in the workgroup's first wavefront this tail took 7 %
// of the whole kernel: 14 of 205 us at cfg3.)
template <int TT>
__device__ __forceinline__ void mac_packed_bin0(const StreamJob& job, const FilterDev& f, float2* __restrict__ Y,
                                                int t0, int pe0, int pe1, size_t yrow0) {
    if (blockIdx.x != 0) return;                              // uniform over the workgroup
    __shared__ float2 part[256];
    const int P = f.P, K = f.K, ring = job.ring;
    const int G = blockDim.x / TT;                            // partition groups (threads: 64..256, TT <= 32)
    const int tt = threadIdx.x % TT, g = threadIdx.x / TT;
    float re = 0.f, im = 0.f;
    if (g < G) {
        for (int pe = pe0; pe < pe1; ++pe) {
            const PathEntry pth = f.paths[pe];
            const float2* __restrict__ Hd = f.H + (size_t)pth.data * K * P;
            const float2* __restrict__ X = job.fdl + (size_t)pth.in_ch * ring * P;
            const uint64_t mlo = f.mask[pth.data * 4 + 0];
            const uint64_t mhi = f.mask[pth.data * 4 + 1], mtop = f.mask[pth.data * 4 + 2];
#pragma unroll 4
            for (int j = g; j < K; j += G) {
                const bool on = mask_bit(mlo, mhi, mtop, j);
                const int slot = ring_slot(job.slot0, t0 + tt - j, ring);
                const float2 x = on ? gld(X + (size_t)slot * P) : float2{0.f, 0.f};
                const float2 h = on ? gld(Hd + (size_t)j * P) : float2{0.f, 0.f};
                re = fmaf(x.x, h.x, re);
                im = fmaf(x.y, h.y, im);
            }
        }
    }
    part[threadIdx.x] = float2{re, im};
    __syncthreads();
    if (threadIdx.x < TT && t0 + tt < job.nblocks) {
        float2 sum = part[tt];
        for (int k = 1; k < G; ++k) {
            const float2 v = part[tt + k * TT];
            sum.x += v.x;
            sum.y += v.y;
        }
        gst(Y + (yrow0 + tt) * P, sum);
    }
}

// ---------------------------------------------------------------------------
// K2: multiply-accumulate.  grid (bin-pair tiles, outputs * time tiles, streams)
// Each thread owns two adjacent bins (16 B) and TT consecutive output blocks of
// one (stream, output channel); X rows are streamed once per time tile.
// ---------------------------------------------------------------------------
template <int TT>
__global__ __launch_bounds__(256) void mac_kernel(JobRef jr, FilterDev f,
                                                  float2* __restrict__ Y, int tiles) {
    const StreamJob job = fetch_job(jr);
    const int o = blockIdx.y / tiles;
    const int t0 = (blockIdx.y - o * tiles) * TT;
    if (t0 >= job.nblocks) return;
    const int P = f.P, K = f.K, ring = job.ring;
    const int P2 = P >> 1;
    const int bp = blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[tt] = float4{0.f, 0.f, 0.f, 0.f};

    const int pe0 = f.out_first[o], pe1 = f.out_first[o + 1];
    for (int pe = pe0; pe < pe1; ++pe) {
        const PathEntry pth = f.paths[pe];
        const float4* __restrict__ Hd = reinterpret_cast<const float4*>(f.H + (size_t)pth.data * K * P) + bp;
        const float4* __restrict__ X = reinterpret_cast<const float4*>(job.fdl + (size_t)pth.in_ch * ring * P) + bp;
        const uint64_t mlo = f.mask[pth.data * 4 + 0];
        const uint64_t mhi = f.mask[pth.data * 4 + 1], mtop = f.mask[pth.data * 4 + 2];
        for (int u = 0; u < K - 1 + TT; ++u) {
            const int rel = t0 - (K - 1) + u;             // input block, relative to the call's first
            const int slot = ring_slot(job.slot0, rel, ring);
            const float4 x = gld(X + (size_t)slot * P2);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int j = tt + K - 1 - u;
                if (j >= 0 && j < K) {
                    const bool on = mask_bit(mlo, mhi, mtop, j);
                    if (on) {
                        const float4 h = Hd[(size_t)j * P2];
                        cmac2(acc[tt], x, h);
                    }
                }
            }
        }
    }
    const size_t yrow0 = (size_t)job.yunit0 + (size_t)o * job.nblocks + t0;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        if (t0 + tt < job.nblocks) {
            float2* row = Y + (yrow0 + tt) * P;
            if (bp == 0) row[1] = float2{acc[tt].z, acc[tt].w};   // bin 0 is packed: written below
            else reinterpret_cast<float4*>(row)[bp] = acc[tt];
        }
    }
    mac_packed_bin0<TT>(job, f, Y, t0, pe0, pe1, yrow0);
}

// ---------------------------------------------------------------------------
// K2, dense fast path: sliding-window MAC.  Per bin the work is a length-K FIR
// along time, Y(t) = sum_j X(t-j) H(j).  Each thread keeps TT accumulators and a
// TT-deep window of X in registers; per step j it loads ONE H row element and
// ONE new X row element and issues TT complex MACs.  The window is a circular
// buffer whose slot index (tt - j) mod TT is static because the j loop is
// unrolled by TT — no register moves, no re-loads of H.
//   HBM bytes per output block and bin: 8*(K+TT)/TT (X) + 8 (Y) + H via L2.
// ---------------------------------------------------------------------------
// One bin per thread uses the native 2-float vector (a 64-bit VGPR pair) so that a complex
// multiply-accumulate is exactly two v_pk_fma_f32: the op_sel / neg_lo modifiers broadcast
// x.re / x.im, swap h and negate in the instruction itself.  (hipcc's own lowering of the
// scalar form spends two more VALU per MAC building the (-x.im, x.im) and (h.im, h.re) pairs.)
template <int NB> struct BinVec;
template <> struct BinVec<1> { using type = v2f; };
template <> struct BinVec<2> { using type = float4; };

__device__ __forceinline__ void cmacv(v2f& acc, const v2f& x, const v2f& h) {
    // acc.lo += x.lo*h.lo ; acc.hi += x.lo*h.hi
    // acc.lo -= x.hi*h.hi ; acc.hi += x.hi*h.lo
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(acc)
        : "v"(x), "v"(h));
}
__device__ __forceinline__ void cmacv(float4& acc, const float4& x, const float4& h) { cmac2(acc, x, h); }
// The two halves of a complex MAC into two accumulators (their sum is the MAC): no instruction of
// a long MAC chain then depends on its predecessor.
__device__ __forceinline__ void cmac_re(v2f& acc, const v2f& x, const v2f& h) {   // acc += x.re * (h.re, h.im)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(h));
}
__device__ __forceinline__ void cmul_re(v2f& acc, const v2f& x, const v2f& h) {   // acc = x.re * (h.re, h.im)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(x), "v"(h));
}
__device__ __forceinline__ void cmul_im(v2f& acc, const v2f& x, const v2f& h) {   // acc = x.im * (-h.im, h.re)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(acc) : "v"(x), "v"(h));
}
__device__ __forceinline__ void cmac_im(v2f& acc, const v2f& x, const v2f& h) {   // acc += x.im * (-h.im, h.re)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(x), "v"(h));
}
__device__ __forceinline__ void vzero(v2f& v) { v = v2f{0.f, 0.f}; }
__device__ __forceinline__ void vzero(float4& v) { v = float4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ v2f vload(const v2f* p) { return *(const FK_GLOBAL v2f*)p; }
__device__ __forceinline__ float4 vload(const float4* p) { return gld(p); }

// Tiles with <= 32 accumulator/window floats per array are held to 128 VGPRs (four waves per
// SIMD): the streamed loads need that much parallelism in flight.
template <int TT, int NB, int D>
__global__ __launch_bounds__(256, (TT * NB <= 16) ? 4 : 1) void mac_slide_kernel(JobRef jr, FilterDev f,
                                                        float2* __restrict__ Y, int tiles) {
    using V = typename BinVec<NB>::type;
    const StreamJob job = fetch_job(jr);
    const int o = blockIdx.y / tiles;
    const int tile = blockIdx.y - o * tiles;
    const int t0 = tile * TT;
    if (t0 >= job.nblocks) return;
    const int P = f.P, K = f.K, ring = job.ring;
    const size_t PV = (size_t)(P / NB);                     // vectors per spectrum row
    const int bv = blockIdx.x * blockDim.x + threadIdx.x;   // this thread's vector within a row
    // Even time tiles walk the partitions up (j = 0 .. K-1: the window takes ever older blocks), odd
    // tiles DOWN from the oldest partition (the window takes ever newer blocks).  Neighbouring tiles
    // share all but TT of their K-1+TT input rows; walking them in opposite directions halves the
    // distance in time between the two reads of a shared row, so more of the second reads still
    // find it in the XCD's L2.  Both directions are ONE instruction stream: walking down, the
    // window and the accumulators are simply held in mirrored order (slot q <-> TT-1-q).
    const bool down = (tile & 1) != 0;
    const int first = down ? t0 - (K - 1) : t0;             // first block of the initial window
    const int xdir = down ? 1 : -1;                         // ring direction of the block entering next
    const ptrdiff_t hstep = down ? -(ptrdiff_t)PV : (ptrdiff_t)PV;
    V acc[TT];                                              // acc[a]: output block t0 + (down ? TT-1-a : a)
#pragma unroll
    for (int a = 0; a < TT; ++a) vzero(acc[a]);

    const int pe0 = f.out_first[o], pe1 = f.out_first[o + 1];
    for (int pe = pe0; pe < pe1; ++pe) {
        const PathEntry pth = f.paths[pe];
        const V* __restrict__ Hd = reinterpret_cast<const V*>(f.H + (size_t)pth.data * K * P) + bv;
        const V* __restrict__ X = reinterpret_cast<const V*>(job.fdl + (size_t)pth.in_ch * ring * P) + bv;
        const uint64_t mlo = f.mask[pth.data * 4 + 0];
        const uint64_t mhi = f.mask[pth.data * 4 + 1], mtop = f.mask[pth.data * 4 + 2];
        V xw[TT];                                           // slot q: the block at distance d from `first`, d == (down ? TT-1-q : q) (mod TT)
#pragma unroll
        for (int q = 0; q < TT; ++q)
            xw[q] = vload(X + (size_t)ring_slot(job.slot0, first + (down ? TT - 1 - q : q), ring) * PV);
        // software pipeline, D steps deep: the H and X elements of step i+D are
        // requested at step i (keeps >= 40 KB per CU in flight at 2-4 waves/SIMD)
        static_assert(TT % D == 0, "prefetch ring must divide the unroll");
        V hq[D], xq[D];
        // Row cursors advance by one row per step (no per-step modulo / multiply).
        int xs = ring_slot(job.slot0, down ? first + TT : t0 - 1, ring);   // ring slot of the block entering next
        const V* hp = down ? Hd + (size_t)(K - 1) * PV : Hd;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < K) {
                hq[d] = vload(hp);
                xq[d] = vload(X + (size_t)xs * PV);
                hp += hstep;
                xs += xdir;
                xs = (xs < 0) ? xs + ring : (xs >= ring) ? xs - ring : xs;
            }
        }
        for (int i0 = 0; i0 < K; i0 += TT) {
#pragma unroll
            for (int ii = 0; ii < TT; ++ii) {
                const int i = i0 + ii;                      // step; partition j = i (up) or K-1-i (down)
                if (i < K) {
                    const V h = hq[ii % D];
                    const V xnew = xq[ii % D];
                    if (i + D < K) {
                        hq[ii % D] = vload(hp);
                        xq[ii % D] = vload(X + (size_t)xs * PV);
                        hp += hstep;
                        xs += xdir;
                        xs = (xs < 0) ? xs + ring : (xs >= ring) ? xs - ring : xs;
                    }
                    const bool on = mask_bit(mlo, mhi, mtop, down ? K - 1 - i : i);
                    if (on) {
#pragma unroll
                        for (int a = 0; a < TT; ++a) cmacv(acc[a], xw[(a - ii) & (TT - 1)], h);
                    }
                    xw[(TT - 1 - ii) & (TT - 1)] = xnew;    // replaces the block that has just left the window
                }
            }
        }
    }
    const size_t yrow0 = (size_t)job.yunit0 + (size_t)o * job.nblocks + t0;
#pragma unroll
    for (int a = 0; a < TT; ++a) {
        const int tt = down ? TT - 1 - a : a;
        if (t0 + tt < job.nblocks) {
            float2* row = Y + (yrow0 + tt) * P;
            if constexpr (NB == 2) {
                if (bv == 0) row[1] = float2{acc[a].z, acc[a].w};   // bin 0 is packed: written below
                else reinterpret_cast<float4*>(row)[bv] = acc[a];
            } else {
                if (bv != 0) gst_v2(row + bv, acc[a]);
            }
        }
    }
    mac_packed_bin0<TT>(job, f, Y, t0, pe0, pe1, yrow0);
}

// ---------------------------------------------------------------------------
// K2, whole-call walk: one thread owns one bin of one (stream, output) for the WHOLE call.
// Per bin the MAC is a length-K FIR along time; the thread keeps the filter's K rows of that bin
// (KR registers pairs) and a window of the last K input spectra of that bin in registers and walks
// the call's blocks in order: per output block ONE new X element is loaded, ONE Y element stored,
// K complex MACs issued.  Every X row is read from HBM exactly once per call (plus the K - 1 rows
// of history at the start) and every Y row written once: HBM bytes per output bin
// 8*(K - 1 + T)/T + 8 against 8*(K + 16)/16 + 8 of the 16-output sliding window.
//
// The window is a ring of W = KR + D register pairs: block t lives in slot t mod W, and the D
// slots beyond the K rows still needed receive the loads of blocks t+1 .. t+D while they are in
// flight — the window IS the prefetch buffer.  The t loop is unrolled by W, so every slot index
// (u - j) mod W is static: no register moves, no indexing.
//   one path per output (the launcher checks), K <= KR; bin 0 (packed DC / Nyquist: two real spectra) falls out of
//   the same instructions, the re-part and im-part accumulators being kept apart until the store.
// ---------------------------------------------------------------------------
// Several lanes per bin (LPB = 2 or 4) carry filters of more than KR rows: lane `sub` of a bin's group holds rows
// sub*KR .. sub*KR + KR - 1 of G and a window of the spectra those rows meet — the blocks KR*sub further back in time.
// Only lane 0 of a group loads from memory.  The element that leaves a lane's window at a step (block t - KR in the
// lane's own frame) is exactly what the next lane needs as ITS newest element at that step: it is handed down with one
// DPP row shift per register before its slot is re-used, so every X element is still read from HBM once per call
// however long the filter.  The partial sums of a group are added with two quad-permute DPP steps and every lane of
// the group stores the (same) result.  K + 1 <= 33 * LPB rows: 65 (512 k taps) with two lanes, 129 (MAXSIZE) with four.
//
// Time tiles (`tiles` > 1): a one-stream call does not have enough (bin, output) pairs to fill the chip; its blocks are
// cut into `tiles` runs of `tile_len`, each walked by its own workgroups (and re-reading the rows of history before it).
//
// Several PATHS per output (NP = 2 or 4: full filter matrices, e.g. a true-stereo reverb's four paths): the LPB lanes of
// a bin's group are NP sets of LPP = LPB / NP lanes, one set per input path of the output — its own rows of G, its own
// input channel's spectra (a per-lane byte offset on the uniform row base), its own head lane that loads; the hand-down
// stays inside a set, and the group's partial sums add up to the output's spectrum exactly as above.  An output with
// fewer than NP paths leaves the spare sets' G at zero.
template <int KR, int D, bool PIN = false, int NACC = 6, int LPB = 1, int NP = 1>
__global__ __launch_bounds__(256, (2 * (2 * KR + D) + 24 + (LPB > 1 ? 16 : 0) <= 128) ? 4 : (2 * (2 * KR + D) + 12 + (LPB > 1 ? 16 : 0) <= 168) ? 3 : 2) void mac_walk_kernel(
    JobRef jr, FilterDev f, float2* __restrict__ Y, int tiles, int tile_len) {
    constexpr int W = KR + D;
    static_assert(LPB == 1 || LPB == 2 || LPB == 4, "lanes per bin");
    static_assert((NP == 1 || NP == 2 || NP == 4) && NP <= LPB, "paths per output");
    constexpr int LPP = LPB / NP;                           // lanes per path
    const StreamJob job = fetch_job(jr);
    const int o = blockIdx.y / tiles;
    const int tb = (blockIdx.y - o * tiles) * tile_len;     // first block of this workgroup's time tile
    if (tb >= job.nblocks) return;
    const int nb = min(tile_len, job.nblocks - tb);
    const int P = f.P, K = f.K, ring = job.ring;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int bin = tid / LPB, sub = tid % LPB;
    const int pi = sub / LPP;                               // which of the output's paths this lane works for
    const int jb = (sub % LPP) * KR;                        // this lane's first row of G
    const bool head = sub % LPP == 0;                       // the lane of its set that takes the loaded element
    const unsigned voff = (unsigned)bin * 8u;               // this thread's bin inside any spectrum row
    const bool packed = bin == 0;
    const int pe0 = f.out_first[o], pe1 = f.out_first[o + 1];
    const size_t yrow0 = (size_t)job.yunit0 + (size_t)o * job.nblocks + tb;
    if (pe1 > pe0) {
        const bool path_on = NP == 1 || pi < pe1 - pe0;
        const PathEntry pth = f.paths[NP == 1 ? pe0 : pe0 + (path_on ? pi : 0)];
        // row bases are wave-uniform (scalar registers); the per-lane part of every address is `voff` — plus, with
        // several paths, the path's spectra set and input channel (the launcher checks that 32 bits hold them)
        const float2* __restrict__ Hd = NP == 1 ? f.H + (size_t)pth.data * K * P : f.H;
        const float2* __restrict__ X = NP == 1 ? job.fdl + (size_t)pth.in_ch * ring * P : job.fdl;
        const unsigned voff_g = NP == 1 ? voff : voff + (unsigned)pth.data * (unsigned)K * (unsigned)P * 8u;
        const unsigned voff_x = NP == 1 ? voff : voff + (unsigned)pth.in_ch * (unsigned)ring * (unsigned)P * 8u;
        auto ldrow_g = [&](const float2* rowbase) -> v2f {
            return *(const FK_GLOBAL v2f*)((const FK_GLOBAL char*)rowbase + voff_g);
        };
        auto ldrow = [&](const float2* rowbase) -> v2f {
            return *(const FK_GLOBAL v2f*)((const FK_GLOBAL char*)rowbase + voff_x);
        };
        v2f g[KR], w[W];
        // Every load below is issued unconditionally (rows that do not exist are replaced by a valid
        // row and zeroed by a select): with a fixed sequence of memory operations the compiler's
        // s_waitcnt vmcnt(N) for block t leaves exactly the D younger loads and the stores in flight;
        // behind a conditional load it would have to assume the worst and drain the prefetch.
        // history: block -j in slot W - j (j = 1 .. K-1); the call's first D blocks in slots 0 .. D-1
        if constexpr (LPB == 1) {
            if (K == KR) {                                  // the common case (K = 32 partitions + 1): no selects
#pragma unroll
                for (int j = 0; j < KR; ++j) g[j] = ldrow_g(Hd + (size_t)j * P);
#pragma unroll
                for (int j = 1; j < KR; ++j) w[W - j] = ldrow(X + (size_t)ring_slot(job.slot0, tb - j, ring) * P);
            } else {
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const v2f v = ldrow_g(Hd + (size_t)(j < K ? j : 0) * P);
                    g[j] = (j < K) ? v : v2f{0.f, 0.f};
                }
#pragma unroll
                for (int j = 1; j < KR; ++j) {
                    const v2f v = ldrow(X + (size_t)ring_slot(job.slot0, j < K ? tb - j : tb, ring) * P);
                    w[W - j] = (j < K) ? v : v2f{0.f, 0.f};
                }
            }
            w[D] = v2f{0.f, 0.f};
        } else {
            // per-lane rows: G row jb + j; the window starts KR * sub blocks back.  Slot D (block -KR of the lane's
            // frame) is loaded too: it is the first element handed down to the next lane.
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const bool on = path_on && jb + j < K;
                const v2f v = ldrow_g(Hd + (size_t)(on ? jb + j : 0) * P);
                g[j] = on ? v : v2f{0.f, 0.f};
            }
#pragma unroll
            for (int j = 1; j <= KR; ++j) {
                const bool on = path_on && jb + j < K;      // (an element no row will ever meet is a zero)
                const v2f v = ldrow(X + (size_t)ring_slot(job.slot0, on ? tb - j - jb : tb, ring) * P);
                w[W - j] = on ? v : v2f{0.f, 0.f};
            }
        }
        // PIN: the in-loop loads are issued by inline asm at the step they belong to and awaited by an
        // explicit s_waitcnt with the exact count of younger memory operations.  Left to itself the
        // compiler sinks each load to the step that first uses it (it shortens live ranges at the
        // register limit this kernel runs at), which turns a D-deep prefetch into none.  The window
        // register is the asm's output and the wait's in/out operand, so every use is ordered behind
        // its wait; the kernel must stay free of spills and copies of window registers (checked in the
        // disassembly: no v_mov of a window register inside the loop; `make check-isa` looks for scratch).
        auto issue = [&](v2f& dst, const float2* rowbase) {
            if constexpr (PIN) asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(dst) : "v"(voff_x), "s"(rowbase) : "memory");
            else dst = ldrow(rowbase);
        };
        // the next row to request, as a pointer that wraps at the ring's end (a handful of scalar
        // instructions per step instead of a slot * row-size product)
        const float2* xrow = X + (size_t)ring_slot(job.slot0, tb, ring) * P;
        const float2* const xend = X + (size_t)ring * P;
        int left = __builtin_amdgcn_readfirstlane(nb);      // blocks not yet requested (a scalar, and the compiler is told so)
        auto advance = [&]() {                              // past the tile's last block: stay on it (re-read, never used)
            const bool more_rows = left > 1;
            const float2* nx = xrow + P;
            nx = (nx == xend) ? X : nx;
            xrow = more_rows ? nx : xrow;
            left = max(left - 1, 1);
        };
        if constexpr (PIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // G and the history have arrived: the count starts here
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue(w[d], xrow);
            advance();
        }
        // Every step waits for the element of the NEXT step (below); the first step's own element is awaited here.
        // A wait is an asm statement that "writes" the window register (that is what orders its uses behind it), and
        // hipcc pads ten to thirteen s_nop between such a statement and an inline-asm MAC that reads the register
        // right after it (it cannot see what the asm does): with the wait one step ahead of the use the padding is gone.
        if constexpr (PIN) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(D - 1) : "memory");
        if constexpr (LPP > 1) {
            // (the hand-down of block tb - KR: slot D of the neighbour, read before the loop's first issue re-uses it)
            const float hx = dpp_row_shr1(w[D].x), hy = dpp_row_shr1(w[D].y);
            w[0].x = head ? w[0].x : hx;
            w[0].y = head ? w[0].y : hy;
        }
        float2* __restrict__ yrow = Y + yrow0 * P;          // uniform: advances one row per step
        for (int t0 = 0; t0 < nb; t0 += W) {
            // unrolled by W through a fold expression (every window index a compile-time constant; the
            // loop unroller gives up on a body of this size), with an exit after the tile's last block
            const bool more = static_all<W>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                // block t + D rides in while blocks t .. t+D-1 are used: its slot held block t - KR, no longer needed
                issue(w[(u + D) % W], xrow);
                advance();
                // the NEXT step's element.  For the lanes of a group behind the head it is the element that leaves the
                // lane before it at that step (block t + 1 - KR of that lane's frame, in the slot the next issue re-uses):
                // handed down by one DPP row shift per register; the head lanes take what was loaded.
                if constexpr (PIN) {
                    // younger than the load of block t + 1: the D - 1 loads after it and the stores of the steps since it
                    // was issued — D - 1 of them once the walk is that old (fewer before: the count below is exact in the
                    // first round and merely stricter at the start of later ones)
                    constexpr int N = (D - 1) + (u < D - 1 ? u : D - 1);
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[(u + 1) % W]) : "n"(N) : "memory");
                }
                if constexpr (LPP > 1) {
                    const float hx = dpp_row_shr1(w[(u + 1 + D) % W].x), hy = dpp_row_shr1(w[(u + 1 + D) % W].y);
                    w[(u + 1) % W].x = head ? w[(u + 1) % W].x : hx;
                    w[(u + 1) % W].y = head ? w[(u + 1) % W].y : hy;
                }
                // NACC accumulators — the real-part and the imaginary-part products of every (NACC/2)-th partition —
                // so that every v_pk_fma_f32 is NACC instructions away from the one it depends on (a MAC's own
                // two halves back to back stall each other), and so that hipcc's hazard recognizer, which pads
                // with s_nop when two inline-asm statements touching one register are fewer than five apart
                // (it cannot see that they are plain VALU), finds nothing to pad
                v2f acc[NACC];
                static_for<KR>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j < NACC / 2) {            // an accumulator's first product: a multiply, no zeroing
                        cmul_re(acc[2 * j], w[(u - j + 2 * W) % W], g[j]);
                        cmul_im(acc[2 * j + 1], w[(u - j + 2 * W) % W], g[j]);
                    } else {
                        cmac_re(acc[2 * (j % (NACC / 2))], w[(u - j + 2 * W) % W], g[j]);
#ifdef FOLVE_EXPERIMENT_CUT
                        if constexpr (j % 2 == 0)           // what-if: three packed FMAs per two rows (wrong results)
#endif
                        cmac_im(acc[2 * (j % (NACC / 2)) + 1], w[(u - j + 2 * W) % W], g[j]);
                    }
                });
                // the re-part and the im-part accumulators are summed apart: the packed bin 0 holds (DC, Nyquist),
                // two REAL spectra, whose products are (sum x.re g.re, sum x.im g.im) = (re-part.lo, -im-part.lo)
                v2f sre = acc[0], sim = acc[1];
#pragma unroll
                for (int a = 2; a < NACC; a += 2) { sre += acc[a]; sim += acc[a + 1]; }
                v2f sum = sre + sim;
                sum.x = packed ? sre.x : sum.x;
                sum.y = packed ? -sim.x : sum.y;
                if constexpr (LPB >= 2) {                    // the group's partial sums: every lane ends up with the total
                    sum.x += dpp_quad<0xB1>(sum.x); sum.y += dpp_quad<0xB1>(sum.y);
                }
                if constexpr (LPB >= 4) {
                    sum.x += dpp_quad<0x4E>(sum.x); sum.y += dpp_quad<0x4E>(sum.y);
                }
                // unconditional: a store under a branch would not count in the compiler's vmcnt arithmetic
                // and halve the prefetch depth
                if constexpr (PIN) asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(voff), "v"(sum), "s"(yrow) : "memory");   // (scalar row base + lane offset: no per-lane 64-bit pointer to advance)
                else *(FK_GLOBAL v2f*)((FK_GLOBAL char*)yrow + voff) = sum;
                yrow += P;
                return t0 + u + 1 < nb;
            });
            if (!more) break;
        }
    } else {
        // an output without an input path: silence
        for (int t = 0; t < nb; ++t) gst_v2(Y + (yrow0 + t) * P + bin, v2f{0.f, 0.f});
    }
}

// ---------------------------------------------------------------------------
// K2, latency form: the one-block call of SoundProcessor::Process (one stream, one block).
// grid (P/128, outputs, streams), 64 threads, two bins per thread.  With so little work the time
// is the length of one thread's dependent chain, so the K rows are requested U at a time before
// the first multiply (mac_kernel<1> walks them one load round trip after the other: 13.6 us for
// K = 32 against ~4 here), rows of G that hold no taps are zeros in memory and are simply read,
// and the grid is cut small enough to put a wavefront on every other CU.
//   one block per stream (nblocks == 1); bin 0 (packed DC / Nyquist) by wave reduction.
// ---------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(64) void mac_small_kernel(JobRef jr, FilterDev f,
                                                       float2* __restrict__ Y) {
    STAMP(3);
    const StreamJob job = fetch_job(jr);   // one: the descriptor by value (Tuning::one_job)
    const int o = blockIdx.y;
    const int P = f.P, K = f.K, ring = job.ring;
    const int P2 = P >> 1;
    const int bp = blockIdx.x * 64 + threadIdx.x;          // bin pair
    // The sums run in mac_kernel's order (oldest block first, i.e. partition K-1 down to 0; the packed
    // bin's products summed in ascending partition order), so a stream computes the same bits whether
    // its block travels alone through this kernel or inside a batch through mac_kernel.
    float4 acc{0.f, 0.f, 0.f, 0.f};
    float pr[3] = {0.f, 0.f, 0.f}, pi[3] = {0.f, 0.f, 0.f};   // packed bin: this lane's partitions j = lane + 64*m
    const int pe0 = f.out_first[o], pe1 = f.out_first[o + 1];
    for (int pe = pe0; pe < pe1; ++pe) {
        const PathEntry pth = f.paths[pe];
        const float4* __restrict__ Hd = reinterpret_cast<const float4*>(f.H + (size_t)pth.data * K * P) + bp;
        const float4* __restrict__ X = reinterpret_cast<const float4*>(job.fdl + (size_t)pth.in_ch * ring * P) + bp;
        for (int j1 = K - 1; j1 >= 0; j1 -= U) {
            float4 x[U], h[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int j = (j1 - i >= 0) ? j1 - i : 0;                         // clamped: the surplus is discarded below
                x[i] = gld(X + (size_t)ring_slot(job.slot0, -j, ring) * P2);
                h[i] = gld(Hd + (size_t)j * P2);
            }
#pragma unroll
            for (int i = 0; i < U; ++i)
                if (j1 - i >= 0) cmac2(acc, x[i], h[i]);
        }
        if (blockIdx.x == 0) {
            const float2* __restrict__ H0 = f.H + (size_t)pth.data * K * P;
            const float2* __restrict__ X0 = job.fdl + (size_t)pth.in_ch * ring * P;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int j = threadIdx.x + 64 * m;
                if (j < K) {
                    const float2 x = gld(X0 + (size_t)ring_slot(job.slot0, -j, ring) * P);
                    const float2 h = gld(H0 + (size_t)j * P);
                    pr[m] = fmaf(x.x, h.x, pr[m]);
                    pi[m] = fmaf(x.y, h.y, pi[m]);
                }
            }
        }
    }
    float2* row = Y + ((size_t)job.yunit0 + o) * P;         // nblocks == 1: one row per (stream, output)
    if (blockIdx.x == 0) {
        float dc = 0.f, ny = 0.f;
        for (int j = 0; j < K; ++j) {                       // K <= 129 wave-uniform steps
            const int m = j >> 6;
            const float vr = __shfl(m == 0 ? pr[0] : m == 1 ? pr[1] : pr[2], j & 63, 64);
            const float vi = __shfl(m == 0 ? pi[0] : m == 1 ? pi[1] : pi[2], j & 63, 64);
            dc = (j == 0) ? vr : dc + vr;
            ny = (j == 0) ? vi : ny + vi;
        }
        if (bp == 0) { acc.x = dc; acc.y = ny; }
    }
    reinterpret_cast<float4*>(row)[bp] = acc;
    STAMP(4);
}

// the cross-lane exchanges of fft_core.hpp on lane ids (tests/test_forms_gpu.py::test_xlane_exchange_semantics)
__global__ __launch_bounds__(64) void xlane_selftest_kernel(float* __restrict__ out) {
    const int lane = threadIdx.x;
    float a = (float)lane, b = 100.f + (float)lane;
    xlane_swap32(a, b);
    out[lane] = a; out[64 + lane] = b;
    a = (float)lane; b = 100.f + (float)lane;
    xlane_swap16(a, b);
    out[128 + lane] = a; out[192 + lane] = b;
    float r0 = (float)(lane / 16), r1 = 10.f + (float)(lane / 16), r2 = 20.f + (float)(lane / 16), r3 = 30.f + (float)(lane / 16);
    xlane_transpose4(r0, r1, r2, r3);
    out[256 + lane] = r0; out[320 + lane] = r1; out[384 + lane] = r2; out[448 + lane] = r3;
}

template <template <int> class Fn, class... A>
hipError_t dispatch_log2p(int log2P, A&&... a) {
    switch (log2P) {
        case 6: return Fn<6>::run(a...);
        case 7: return Fn<7>::run(a...);
        case 8: return Fn<8>::run(a...);
        case 9: return Fn<9>::run(a...);
        case 10: return Fn<10>::run(a...);
        case 11: return Fn<11>::run(a...);
        case 12: return Fn<12>::run(a...);
        case 13: return Fn<13>::run(a...);
#ifdef FOLVE_EXPERIMENT_P16
        case 14: return Fn<14>::run(a...);
#endif
        default: return hipErrorInvalidValue;
    }
}

// Walker run length: the longest walk (<= 32 blocks) that still gives every CU two workgroups.
static int auto_run(int njobs, int max_blocks) {
    int runlen = 32;
    while (runlen > 1 && (long long)njobs * ((max_blocks + runlen - 1) / runlen) < 512) runlen >>= 1;
    return runlen;
}

template <int L>
struct FwdLaunch {
    // pairs_ok: every stream's PCM pointer is 16-byte aligned (stereo frames loaded as quads / pairs)
    static hipError_t run(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, bool pairs_ok,
                          const Tuning& tn, hipStream_t st) {
        const JobRef jr = make_job_ref(jobs, tn);
        // Stereo fast forms, for launches that give every CU two workgroups (below that the per-channel kernel's
        // twice as many, half as long workgroups finish sooner: a lone stream's 256-block call, K1 21.4 -> 19.9 us and
        // K3 27.2 -> 24.0 us, 384 blocks 28.7 -> 26.8 / 33.2 -> 30.7; from 512 (block, stream) units on the walkers win:
        // tools/crossover_fft_forms.py).
        const bool fast = tn.fft_form != 1 && f.cin == 2 && pairs_ok && !(tn.fft_form == 4 && (long long)njobs * max_blocks <= 512) &&
                          (tn.fft_form == 2 || (long long)njobs * max_blocks >= (L == 13 && !(tn.host_io && !tn.in_resident) ? 512 : 256));   // (PCM over the bus: whole quads from 256 on, as before)
        if (fast) {
            if constexpr (L == 13) {
                // P = 8192: walk consecutive blocks
                const int runlen = tn.fwd_run > 0 ? tn.fwd_run : auto_run(njobs, max_blocks);
                dim3 grid((max_blocks + runlen - 1) / runlen, 1, njobs), block(WaveGeom<L>::NT);
                FK_LAUNCH(0, (forward_walker_kernel<L, true>), grid, block, st, jr, f, runlen, 0);
                return hipGetLastError();
            } else if constexpr (L >= 9 && L < 13) {      // 2P >= 1024: the one-transform stereo form exists
                if (f.twa2) {
                    dim3 grid(max_blocks, 1, njobs), block(WaveGeom<L + 1>::NT);
                    FK_LAUNCH(0, forward_dual_kernel<L>, grid, block, st, jr, f);
                    return hipGetLastError();
                }
            }
        }
        if constexpr (L == 13) {
            // the one-block call from host memory: ONE 1024-thread workgroup, a half per channel, both
            // transforms at once (one workgroup per CU: latency is all that counts here)
            if (((tn.host_io && !tn.in_resident && tn.fft_form == 0 && (long long)njobs * max_blocks <= 64) ||
                 (tn.fft_form == 4 && (long long)njobs * max_blocks <= 512)) && f.cin == 2 && pairs_ok) {
                dim3 grid(max_blocks, 1, njobs), block(2 * WaveGeom<L>::NT);
                FK_LAUNCH(0, (forward_pair_kernel<L, true>), grid, block, st, jr, f);
                return hipGetLastError();
            }
        }
        const int xl = max_blocks >= 8 ? 1 : 0;               // XCD-local order of a block's channels (see forward_kernel)
        if constexpr (L == 13) {
            // many channels, enough (block, pair) units for every workgroup to walk at least two: the walker per channel
            // pair — the next block's PCM flies during this one's transforms, two workgroups per CU
            const int pairs = f.cin / 2;
            if (tn.fft_form != 1 && tn.fft_form != 3 && pairs_ok && f.cin >= 4 && f.cin % 2 == 0 &&
                (tn.fft_form == 2 || (long long)njobs * pairs * max_blocks >= 1024)) {
                const int runlen = tn.fwd_run > 0 ? tn.fwd_run : auto_run(njobs * pairs, max_blocks);
                const int runs = (max_blocks + runlen - 1) / runlen;
                const int wxl = runs >= 8 ? 1 : 0;
                const dim3 grid = wxl ? dim3(8 * pairs, (runs + 7) / 8, njobs) : dim3(runs, pairs, njobs);
                FK_LAUNCH(0, (forward_walker_kernel<L, true, true>), grid, dim3(WaveGeom<L>::NT), st, jr, f, runlen, wxl);
                return hipGetLastError();
            }
        }
        if (tn.fft_form != 1 && pairs_ok && f.cin >= 4 && f.cin % 2 == 0) {   // many channels: a workgroup per channel pair
            const dim3 grid = xl ? dim3(8 * (f.cin / 2), (max_blocks + 7) / 8, njobs) : dim3(max_blocks, f.cin / 2, njobs);
            FK_LAUNCH(0, forward_chpair_kernel<L>, grid, dim3(WaveGeom<L>::NT), st, jr, f, xl);
            return hipGetLastError();
        }
        const dim3 grid = xl ? dim3(8 * f.cin, (max_blocks + 7) / 8, njobs) : dim3(max_blocks, f.cin, njobs);
        FK_LAUNCH(0, forward_kernel<L>, grid, dim3(WaveGeom<L>::NT), st, jr, f, xl);
        return hipGetLastError();
    }
};
template <int L>
struct InvLaunch {
    // pairs_ok: every stream's output pointer is 16-byte aligned
    static hipError_t run(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, const float2* Y,
                          bool pairs_ok, const Tuning& tn, hipStream_t st) {
        const JobRef jr = make_job_ref(jobs, tn);
        constexpr int NT = WaveGeom<L>::NT;
        if constexpr (L == 13) {      // P = 8192 (every filter longer than 4096 taps): one column pair per thread
            // The walker halves the workgroup count; keep the general kernel while that would leave CUs idle.
            // host_io: the output goes over the bus — the walker's whole 16-byte quads in full lines, not
            // the general kernel's interleaved 4-byte stores (27 us against ~12 for one stereo block)
            // the one-block call from host memory, stereo: both outputs at once in one 1024-thread workgroup
            if (((tn.host_io && tn.fft_form == 0 && (long long)njobs * max_blocks <= 64) ||
                 (tn.fft_form == 4 && (long long)njobs * max_blocks <= 512)) && pairs_ok && f.cout == 2) {
                dim3 grid(max_blocks, 1, njobs), block(2 * NT);
                FK_LAUNCH(2, (inverse_pair_kernel<L, true>), grid, block, st, jr, f, Y);
                return hipGetLastError();
            }
            const bool fast = tn.fft_form != 1 && pairs_ok && (f.cout == 1 || f.cout == 2) &&
                              (tn.fft_form == 2 || tn.host_io || (long long)njobs * max_blocks >= (f.cout == 2 ? 512 : 256));
            if (fast) {
                const int runlen = tn.inv_run > 0 ? tn.inv_run : auto_run(njobs, max_blocks);
                dim3 grid((max_blocks + runlen - 1) / runlen, 1, njobs), block(NT);
                if (f.cout == 2) FK_LAUNCH(2, (inverse_walker_kernel<L, 2, true>), grid, block, st, jr, f, Y, runlen, 0);
                else FK_LAUNCH(2, (inverse_walker_kernel<L, 1, true>), grid, block, st, jr, f, Y, runlen, 0);
                return hipGetLastError();
            }
            // many outputs: the walker per output pair (see FwdLaunch)
            const int pairs = f.cout / 2;
            if (tn.fft_form != 1 && tn.fft_form != 3 && pairs_ok && f.cout >= 4 && f.cout % 2 == 0 &&
                (tn.fft_form == 2 || (long long)njobs * pairs * max_blocks >= 1024)) {
                const int runlen = tn.inv_run > 0 ? tn.inv_run : auto_run(njobs * pairs, max_blocks);
                const int runs = (max_blocks + runlen - 1) / runlen;
                const int wxl = runs >= 8 ? 1 : 0;
                const dim3 grid = wxl ? dim3(8 * pairs, (runs + 7) / 8, njobs) : dim3(runs, pairs, njobs);
                FK_LAUNCH(2, (inverse_walker_kernel<L, 2, true, true>), grid, dim3(NT), st, jr, f, Y, runlen, wxl);
                return hipGetLastError();
            }
        }
        const int xl = max_blocks >= 8 ? 1 : 0;
        if (tn.fft_form != 1 && pairs_ok && f.cout >= 4 && f.cout % 2 == 0) {
            const dim3 grid = xl ? dim3(8 * (f.cout / 2), (max_blocks + 7) / 8, njobs) : dim3(max_blocks, f.cout / 2, njobs);
            FK_LAUNCH(2, inverse_chpair_kernel<L>, grid, dim3(NT), st, jr, f, Y, xl);
            return hipGetLastError();
        }
        const dim3 grid = xl ? dim3(8 * f.cout, (max_blocks + 7) / 8, njobs) : dim3(max_blocks, f.cout, njobs);
        FK_LAUNCH(2, inverse_kernel<L>, grid, dim3(NT), st, jr, f, Y, xl);
        return hipGetLastError();
    }
};
template <int L>
struct FilterLaunch {
    static hipError_t run(const float* taps, float2* Htmp, float2* Gs, int ndata, int K, const FftTables& t,
                          hipStream_t st) {
        dim3 grid(K, ndata), block(WaveGeom<L>::NT);
        hipLaunchKernelGGL(filter_kernel<L>, grid, block, 0, st, taps, Htmp, K, t.tw, t.twa, t.twb);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        dim3 grid2(K + 1, ndata), block2(256);
        hipLaunchKernelGGL(make_g_kernel, grid2, block2, 0, st, Htmp, Gs, K, 1 << L);
        return hipGetLastError();
    }
};

}  // namespace

hipError_t launch_forward(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, bool pairs_ok,
                          const Tuning& tn, hipStream_t st) {
    return dispatch_log2p<FwdLaunch>(f.log2P, jobs, njobs, max_blocks, f, pairs_ok, tn, st);
}

hipError_t launch_inverse(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, const float2* Y,
                          bool walker_ok, const Tuning& tn, hipStream_t st) {
    return dispatch_log2p<InvLaunch>(f.log2P, jobs, njobs, max_blocks, f, Y, walker_ok, tn, st);
}

namespace {
__global__ __launch_bounds__(256) void hbm_read_kernel(const v4f* __restrict__ a, v4f* __restrict__ sink, size_t n) {
    v4f acc{0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
    if (acc.x == 12345.678f) sink[0] = acc;                   // never true for the probe's data: keeps the loads
}
__global__ __launch_bounds__(256) void hbm_write_kernel(v4f* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v4f{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void hbm_copy_kernel(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// The same stores, every workgroup into its OWN contiguous region (what K1, K3 and a tiled K2 do: a workgroup walks its
// rows) instead of one front of consecutive kilobytes moving through the buffer: 5.6 - 6.1 TB/s against 4.0 - 5.1
// (tools/micro/write_rate.hip; reads do not care: 6.0 - 6.4 TB/s either way).
__global__ __launch_bounds__(256) void hbm_write_regions_kernel(v4f* __restrict__ b, size_t n) {
    const size_t per = n / gridDim.x;
    v4f* __restrict__ p = b + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += 256) p[i] = v4f{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void hbm_copy_regions_kernel(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    const size_t per = n / gridDim.x;
    const v4f* __restrict__ pa = a + (size_t)blockIdx.x * per;
    v4f* __restrict__ pb = b + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += 256) pb[i] = pa[i];
}
// The best float4 copy found on this pool's MI355X boxes (tools/micro/copy_rate.hip, profiles/r06_copy_rate.txt: 5.87 - 5.99 TB/s
// counting both directions; /opt/skills/guides/MI355X_MICROARCH.md:36 quotes 6.29): non-temporal loads AND stores, a read burst of
// 4 KiB per wave followed by its write burst, grid-stride tiles.  The yardstick bench.py holds K1 / K2 / K3 against.
__global__ __launch_bounds__(256) void hbm_copy_best_kernel(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    constexpr int U = 4;
    const size_t tile = (size_t)256 * U, tiles = n / tile;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const v4f* p = a + t * tile + threadIdx.x;
        v4f* q = b + t * tile + threadIdx.x;
        v4f r[U];
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = __builtin_nontemporal_load(p + j * 256);
#pragma unroll
        for (int j = 0; j < U; ++j) __builtin_nontemporal_store(r[j], q + j * 256);
    }
}
}  // namespace

hipError_t launch_hbm_probe(int mode, const void* a, void* b, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 16;
    dim3 grid(2048), block(256);                              // 8 workgroups per CU, grid-stride
    if (mode == 5) {
        hipLaunchKernelGGL(hbm_copy_best_kernel, dim3(256 * 12), block, 0, st, (const v4f*)a, (v4f*)b, n);
        return hipGetLastError();
    }
    if (mode == 0) hipLaunchKernelGGL(hbm_read_kernel, grid, block, 0, st, (const v4f*)a, (v4f*)b, n);
    else if (mode == 1) hipLaunchKernelGGL(hbm_write_kernel, grid, block, 0, st, (v4f*)b, n);
    else if (mode == 2) hipLaunchKernelGGL(hbm_copy_kernel, grid, block, 0, st, (const v4f*)a, (v4f*)b, n);
    else if (mode == 3) hipLaunchKernelGGL(hbm_write_regions_kernel, dim3(8192), block, 0, st, (v4f*)b, n);
    else hipLaunchKernelGGL(hbm_copy_regions_kernel, dim3(8192), block, 0, st, (const v4f*)a, (v4f*)b, n);
    return hipGetLastError();
}

hipError_t launch_xlane_selftest(float* out512, hipStream_t st) {
    hipLaunchKernelGGL(xlane_selftest_kernel, dim3(1), dim3(64), 0, st, out512);
    return hipGetLastError();
}

hipError_t launch_filter_transform(const float* taps, float2* Htmp, float2* Gs, int ndata, int K, int log2P,
                                   const FftTables& t, hipStream_t st) {
    return dispatch_log2p<FilterLaunch>(log2P, taps, Htmp, Gs, ndata, K, t, st);
}

// ---- twiddle buffer of a P-point engine -----------------------------------------------------------
//   [ tw2P (2P) | stage A rows of the P-point FFT (4*N2) | stage B pass tables |
//     stage A rows of the 2P-point FFT (stereo form) | its stage B pass tables ]
template <int L>
struct TableLayout {
    static hipError_t run(int* n) {      // n[0..4]: tw, twa, twb, twa2, twb2 entries; n[5], n[6]: N2, log2N2; n[7], n[8]: of 2P
        n[0] = 2 << L;
        n[1] = WaveGeom<L>::TWA;
        n[2] = WaveGeom<L>::TWB;
        n[5] = WaveGeom<L>::N2;
        n[6] = WaveGeom<L>::LOG2N2;
        if constexpr (L >= 9 && L < 13) {                  // P = 8192 has its own stereo forms (walker, pair)
            n[3] = WaveGeom<L + 1>::TWA;
            n[4] = WaveGeom<L + 1>::TWB;
            n[7] = WaveGeom<L + 1>::N2;
            n[8] = WaveGeom<L + 1>::LOG2N2;
        } else {
            n[3] = n[4] = n[7] = n[8] = 0;
        }
        return hipSuccess;
    }
};

int fft_table_count(int log2P) {
    int n[9];
    if (dispatch_log2p<TableLayout>(log2P, n) != hipSuccess) return 0;
    return n[0] + n[1] + n[2] + n[3] + n[4];
}

namespace {
void fill_stage_a(float2* dst, int N2, double N) {     // rows k1 = 1, 2, 4, 8: exp(-2*pi*i*n2*k1/N)
    for (int row = 0, k1 = 1; row < 4; ++row, k1 *= 2)
        for (int n2 = 0; n2 < N2; ++n2) {
            const double a = -2.0 * M_PI * (double)n2 * (double)k1 / N;
            dst[row * N2 + n2] = float2{(float)cos(a), (float)sin(a)};
        }
}
void fill_stage_b(float2* dst, int log2n2) {
    const Plan pl = make_plan(log2n2);
    for (int p = 1; p < pl.n; ++p) {
        const int R = pl.r[p], NS = pl.ns[p];
        for (int r = 1; r < R; ++r)
            for (int k = 0; k < NS; ++k) {
                const double a = -2.0 * M_PI * (double)k * (double)r / ((double)NS * (double)R);
                dst[pl.off[p] + (r - 1) * NS + k] = float2{(float)cos(a), (float)sin(a)};
            }
    }
}
}  // namespace

void fill_fft_tables(int log2P, float2* dst, int off[4]) {
    int n[9];
    (void)dispatch_log2p<TableLayout>(log2P, n);
    const double P = (double)(1 << log2P);
    for (int k = 0; k < n[0]; ++k) {
        const double a = -M_PI * (double)k / P;                       // exp(-2*pi*i*k/(2P))
        dst[k] = float2{(float)cos(a), (float)sin(a)};
    }
    off[0] = n[0];                        // twa
    off[1] = off[0] + n[1];               // twb
    off[2] = off[1] + n[2];               // twa2 (or the end)
    off[3] = off[2] + n[3];               // twb2
    if (n[1]) fill_stage_a(dst + off[0], n[5], P);
    fill_stage_b(dst + off[1], n[6]);
    if (n[3]) fill_stage_a(dst + off[2], n[7], 2.0 * P);
    if (n[4]) fill_stage_b(dst + off[3], n[8]);
}

// Form choice.  A single block per call is a pure stream over K rows (mac_kernel<1>, HBM-bound).
// Run-ahead calls (>= 12 blocks): the whole-call walk when every output has one mostly populated path of at most
// 132 rows — every X row read once; otherwise the 16-output sliding window, which also skips unpopulated rows of
// sparse filters.  The walk's shape (rows per lane, lanes per bin, time tiles) is chosen so that the launch has
// at least two wavefronts per SIMD: a batch of many streams takes one lane per bin, a lone stream spreads its filter
// over the lanes of a group and its blocks over time tiles.
namespace {
// rows of G per lane for `lpb` lanes per bin: the smallest instantiated window that holds ceil(rows / lpb); 0: none
// One lane per bin has a finer ladder (13 / 21 / 26 / 29 rows beside 9 / 17 / 33): a window wider than the filter multiplies
// zeros — SantaLucia's 26 rows (K = 25) in the 33-row window were a fifth of a lone stream's K2 arithmetic.
// (the finer ladder exists for ONE lane per bin only — `lpb` is the whole group, `np` its path sets: a 2 x 2 matrix's
// lpb = np = 2 has one lane per path but only the 17 / 33-row instantiations)
int walk_rows_per_lane(int rows, int lpb, int np) {
    const int lpp = lpb / np;                                  // lanes per path
    const int need = (rows + lpp - 1) / lpp;
    const bool fine = lpb == 1;
    if (need <= 9 && (lpb == 1 || (lpb == 4 && np != 2))) return 9;   // (9 rows: one lane, or four lanes in one or four sets)
    if (fine && need <= 13) return 13;
    if (need <= 17) return 17;
    if (fine && need <= 21) return 21;
    if (fine && need <= 26) return 26;
    if (fine && need <= 29) return 29;
    if (need <= 33) return 33;
    return 0;
}
// np: lanes sets per group, one per path of an output (1, 2 or 4)
bool choose_walk(const FilterDev& f, int njobs, int max_blocks, int np, WalkShape* out) {
    const int rows = f.K;
    if (rows > 132 || f.P < 256) return false;
    const long long want = 2048;                               // wavefronts: two per SIMD
    // a time tile re-reads `rows` rows of history: no shorter than 32 blocks, nor than half the filter
    const int min_tile = rows / 2 > 32 ? rows / 2 : 32;
    out->kr = 0;
    out->np = np;
    for (int lpb = np; lpb <= 4; lpb *= 2) {                   // fewest lanes per bin first: least arithmetic overhead,
        int kr = walk_rows_per_lane(rows, lpb, np);                // widest rows per wavefront
        if (!kr) continue;
        const long long waves = (long long)njobs * f.cout * (f.P / 64) * lpb;
        int tiles = 1;
        while (waves * tiles < want && max_blocks / (tiles * 2) >= min_tile) tiles *= 2;
        out->kr = kr;
        out->lpb = lpb;
        out->tiles = tiles;
        out->tile_len = (max_blocks + tiles - 1) / tiles;
        if (waves * tiles >= want) break;
    }
    return out->kr != 0;
}
template <int KR, int D, int LPB, int NP = 1>
void launch_walk(const JobRef& jr, int njobs, const FilterDev& f, float2* Y, const WalkShape& w, const Tuning& tn, hipStream_t st) {
    dim3 grid(f.P * LPB / 256, f.cout * w.tiles, njobs), block(256);
    // PIN: the window loads issued by inline asm at their step, awaited by hand-counted s_waitcnt (tools/check_isa.py
    // simulates every such loop on the built code object).  `make NO_PIN=1` builds the same walk with compiler-scheduled
    // loads and the compiler's own waits — slower (the loads sink to their first use: K2 0.30 -> 0.37 ms at 64 blocks)
    // but correct by construction: the build to fall back on when a toolchain change makes the check fail.
#ifdef FOLVE_WALK_NO_PIN
    constexpr bool kPin = false;
#else
    constexpr bool kPin = true;
#endif
    // accumulators per lane: four (the real- and imaginary-part products of every second partition).  Six was round 2's
    // number (the hazard padding between dependent inline-asm FMAs); with the wait a step ahead of the use four is enough
    // and saves two packed adds per step: K2 of cfg3's diagonal batch 0.975 -> 0.963 ms, of its 2 x 2 matrix 1.805 -> 1.752,
    // of cfg4 120.7 -> 118.0 us (A/B/A, `-DFOLVE_WALK_NACC=n` variants; two: the same as four, eight: 2 % slower).
#ifndef FOLVE_WALK_NACC
#define FOLVE_WALK_NACC 4
#endif
    FK_LAUNCH(1, (mac_walk_kernel<KR, D, kPin, FOLVE_WALK_NACC, LPB, NP>), grid, block, st, jr, f, Y, w.tiles, w.tile_len);
}
}  // namespace

hipError_t launch_mac(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, float2* Y, int time_tile,
                      const MacShape& shape, const Tuning& tn, hipStream_t st) {
    const JobRef jr = make_job_ref(jobs, tn);
    const int P2 = f.P / 2;
    int form = tn.mac_form;
    WalkShape ws{};
    // one path per output, or up to four with a lane set each (their per-lane offsets must fit 32 bits)
    const int np = shape.max_paths <= 1 ? 1 : shape.max_paths == 2 ? 2 : 4;
    const bool paths_ok = shape.max_paths <= 1 ||
                          (shape.max_paths <= 4 && (unsigned long long)f.cin * (unsigned)tn.max_ring * f.P * 8ull < (1ull << 32) &&
                           (unsigned long long)shape.ndata * f.K * f.P * 8ull < (1ull << 32));
    // (the three-FMA walk addresses a stream's ring and a tile's rows of Y by 32-bit byte offsets)
    const bool offsets_ok = ((unsigned long long)tn.max_ring + 1) * f.P * 8ull < (1ull << 32) && ((unsigned long long)max_blocks + 8) * f.P * 8ull < (1ull << 32);
    // (a limit of the three-FMA form alone: the four-FMA walk addresses by 64-bit pointers)
    const bool walk_ok = paths_ok && choose_walk(f, njobs, max_blocks, np, &ws);
    if (tn.walk_lpb > 0 && walk_ok) {                           // tests: pin the lanes per bin / the time tiles
        int kr = (tn.walk_lpb == 1 || tn.walk_lpb == 2 || tn.walk_lpb == 4) && tn.walk_lpb >= np ? walk_rows_per_lane(f.K, tn.walk_lpb, np) : 0;
        if (kr) { ws.lpb = tn.walk_lpb; ws.kr = kr; }
    }
    if (tn.walk_tiles > 0 && walk_ok) {
        ws.tiles = tn.walk_tiles;
        ws.tile_len = (max_blocks + ws.tiles - 1) / ws.tiles;
    }
    if (form == 100 && !walk_ok) form = 0;
    if (form == 0) {
        if (time_tile >= 12) form = (walk_ok && shape.dense) ? 100 : 16;
        else if (time_tile >= 6) form = 8;
        else if (time_tile >= 4) form = 4;
        else form = 1;
    }
    if (form == 100) {
        // Three FMAs per complex multiply-add (mac_walk3.hip) wherever that form is instantiated: a quarter less arithmetic
        // where the walk's time is arithmetic (cfg4's two lanes per bin: K2 121 -> 110 us; a 2 x 2 matrix: 1.74 -> 1.53 ms),
        // and fewer joules where it is memory (cfg3's batch at the power cap: the whole call 2.25 -> 2.19 ms, K1 and K3
        // included).  tn.walk_fma pins either form.
        const bool fma3 = tn.walk_fma == 3 || tn.walk_fma == 0;
        if (fma3 && offsets_ok && walk3_has(ws.kr, ws.lpb, ws.np)) return launch_walk3(jobs, njobs, f, Y, ws, tn, st);
        // 256-thread workgroups: one wavefront per workgroup ran 11 % slower, two 3 % (a workgroup's four
        // waves start together and read 2 KB of a row between them: DRAM locality)
        if (ws.np == 2 && ws.lpb == 2) {
            if (ws.kr == 17) launch_walk<17, 15, 2, 2>(jr, njobs, f, Y, ws, tn, st);
            else launch_walk<33, 7, 2, 2>(jr, njobs, f, Y, ws, tn, st);
        } else if (ws.np == 2) {
            if (ws.kr == 17) launch_walk<17, 15, 4, 2>(jr, njobs, f, Y, ws, tn, st);
            else launch_walk<33, 7, 4, 2>(jr, njobs, f, Y, ws, tn, st);
        } else if (ws.np == 4) {
            if (ws.kr == 9) launch_walk<9, 15, 4, 4>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 17) launch_walk<17, 15, 4, 4>(jr, njobs, f, Y, ws, tn, st);
            else launch_walk<33, 7, 4, 4>(jr, njobs, f, Y, ws, tn, st);
        } else if (ws.lpb == 1) {
            if (ws.kr == 9) launch_walk<9, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 13) launch_walk<13, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 17) launch_walk<17, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 21) launch_walk<21, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 26) launch_walk<26, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 29) launch_walk<29, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            else launch_walk<33, 7, 1>(jr, njobs, f, Y, ws, tn, st);
        } else if (ws.lpb == 2) {
            if (ws.kr == 17) launch_walk<17, 15, 2>(jr, njobs, f, Y, ws, tn, st);
            else launch_walk<33, 7, 2>(jr, njobs, f, Y, ws, tn, st);
        } else {
            if (ws.kr == 9) launch_walk<9, 15, 4>(jr, njobs, f, Y, ws, tn, st);
            else if (ws.kr == 17) launch_walk<17, 15, 4>(jr, njobs, f, Y, ws, tn, st);
            else launch_walk<33, 7, 4>(jr, njobs, f, Y, ws, tn, st);
        }
        return hipGetLastError();
    }
    if (form == 4 || form == 8 || form == 16) {
        const int nb = form == 16 ? 1 : 2;
        if (f.P / nb >= 64) {
            const int pv = f.P / nb;
            const int nt = pv < 256 ? pv : 256;
            const int tiles = (max_blocks + form - 1) / form;
            dim3 grid(pv / nt, f.cout * tiles, njobs), block(nt);
            if (form == 16) FK_LAUNCH(1, (mac_slide_kernel<16, 1, 4>), grid, block, st, jr, f, Y, tiles);
            else if (form == 8) FK_LAUNCH(1, (mac_slide_kernel<8, 2, 2>), grid, block, st, jr, f, Y, tiles);
            else FK_LAUNCH(1, (mac_slide_kernel<4, 2, 2>), grid, block, st, jr, f, Y, tiles);
            return hipGetLastError();
        }
    }
    if (form == 1 && max_blocks == 1 && njobs <= 8 && P2 >= 64 && tn.mac_form == 0) {
        dim3 grid(P2 / 64, f.cout, njobs), block(64);
        FK_LAUNCH(1, mac_small_kernel<11>, grid, block, st, jr, f, Y);
        return hipGetLastError();
    }
    const int nt = P2 < 256 ? P2 : 256;
    int tt = 1;
    while (tt * 2 <= time_tile && tt < 16) tt *= 2;
    const int tiles = (max_blocks + tt - 1) / tt;
    dim3 grid(P2 / nt, f.cout * tiles, njobs), block(nt);
    switch (tt) {
        case 1: FK_LAUNCH(1, mac_kernel<1>, grid, block, st, jr, f, Y, tiles); break;
        case 2: FK_LAUNCH(1, mac_kernel<2>, grid, block, st, jr, f, Y, tiles); break;
        case 4: FK_LAUNCH(1, mac_kernel<4>, grid, block, st, jr, f, Y, tiles); break;
        case 8: FK_LAUNCH(1, mac_kernel<8>, grid, block, st, jr, f, Y, tiles); break;
        default: FK_LAUNCH(1, mac_kernel<16>, grid, block, st, jr, f, Y, tiles); break;
    }
    return hipGetLastError();
}

}  // namespace fk

#ifdef FOLVE_PHASE_TRACE
// the stamp ring (256 entries) and the number of stamps written so far
extern "C" int fe_debug_stamps(unsigned long long* out256, unsigned int* count) {
    if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(fk::g_stamp), sizeof(unsigned long long) * 256) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(fk::g_stamp_n), sizeof(unsigned int)) != hipSuccess) return -1;
    return 0;
}
// kernel 0 = forward_walker, 1 = inverse_walker; out[16]; reset != 0 clears the counters afterwards
extern "C" int fe_debug_phases(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fk::g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        static const unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fk::g_phase), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
