// kernels.hip — hand-written gfx950 kernels of the folve convolution hot path.
//
// Replaces what `Convproc::process()` does behind SoundProcessor::Process
// (/root/reference/sound-processor.cc:98-127) and what `impdata_create` does at
// filter set-up (/root/reference/zita-config.cc:163): uniformly partitioned FFT
// convolution, partition = block = P = `fragm` frames, evaluated here as
// overlap-save so that every block of a call is independent:
//   K1 forward : window [x(n-1) | x(n)] (2P reals, deinterleaved from PCM) ->
//                real FFT through a P-point complex FFT in LDS -> FDL ring row
//   K2 mac     : Y(n) = sum_paths sum_j X(n-j) * H(j), time-tiled in registers
//   K3 inverse : Y(n) -> P-point complex IFFT in LDS -> last P samples of the
//                window, interleaved PCM store, per-stream peak
//   K0 filter  : taps -> H spectra, 1/(2P) folded in (as zita folds 0.5/parsize)
#include "kernels.h"

#include "fft_core.hpp"

namespace fk {

namespace {

// complex multiply-accumulate on two packed bins
__device__ __forceinline__ void cmac2(float4& acc, const float4& x, const float4& h) {
    acc.x = fmaf(x.x, h.x, acc.x); acc.x = fmaf(-x.y, h.y, acc.x);
    acc.y = fmaf(x.x, h.y, acc.y); acc.y = fmaf(x.y, h.x, acc.y);
    acc.z = fmaf(x.z, h.z, acc.z); acc.z = fmaf(-x.w, h.w, acc.z);
    acc.w = fmaf(x.z, h.w, acc.w); acc.w = fmaf(x.w, h.z, acc.w);
}

__device__ __forceinline__ int ring_slot(int slot0, int rel, int ring) {
    int s = (slot0 + rel) % ring;
    return s < 0 ? s + ring : s;
}

// After the forward passes left Z (P-point FFT of z[m] = x[2m] + i x[2m+1]) in
// the LDS image, form the 2P-point real spectrum and store it as a packed row.
template <int LOG2P>
__device__ __forceinline__ void split_and_store(const float2* s, const float2* __restrict__ tw, int tid,
                                                float2* __restrict__ row, float scale) {
    constexpr int P = 1 << LOG2P;
    constexpr int NT = threads_for(P);
    for (int k = tid; k < P / 2; k += NT) {
        if (k == 0) {
            const float2 z0 = s[phys(0)];
            row[0] = float2{(z0.x + z0.y) * scale, (z0.x - z0.y) * scale};   // (DC, Nyquist)
            const float2 zh = s[phys(P / 2)];
            row[P / 2] = float2{zh.x * scale, -zh.y * scale};
        } else {
            const float2 a = s[phys(k)], b = s[phys(P - k)];
            const float2 e = float2{0.5f * (a.x + b.x), 0.5f * (a.y - b.y)};
            const float2 o = float2{0.5f * (a.y + b.y), -0.5f * (a.x - b.x)};
            const float2 t = cmul(o, tw[k]);
            row[k] = float2{(e.x + t.x) * scale, (e.y + t.y) * scale};
            row[P - k] = float2{(e.x - t.x) * scale, -(e.y - t.y) * scale};
        }
    }
}

// ---------------------------------------------------------------------------
// K1: forward.  grid (max blocks per stream, input channels, streams)
// ---------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(threads_for(1 << LOG2P)) void forward_kernel(const StreamJob* __restrict__ jobs,
                                                                          FilterDev f) {
    constexpr int P = 1 << LOG2P;
    __shared__ float2 s[lds_elems(P)];
    const StreamJob job = jobs[blockIdx.z];
    const int b = blockIdx.x;
    if (b >= job.nblocks) return;
    const int c = blockIdx.y;
    const int tid = threadIdx.x;
    const int cin = f.cin;
    const long long f0 = (long long)(b - 1) * P;          // frame of window sample 0
    const bool last = (b == job.nblocks - 1);
    const float* __restrict__ in = job.in;
    const float2* __restrict__ tail_rd = reinterpret_cast<const float2*>(job.tail_rd + (size_t)c * P);
    float2* __restrict__ tail_wr = reinterpret_cast<float2*>(job.tail_wr + (size_t)c * P);

    auto load = [&](int m) -> float2 {
        float2 v;
        if (m < P / 2 && b == 0) {
            v = tail_rd[m];                               // block preceding this call
        } else {
            const long long fr = f0 + 2 * m;
            v.x = (fr < job.nframes) ? in[fr * cin + c] : 0.0f;
            v.y = (fr + 1 < job.nframes) ? in[(fr + 1) * cin + c] : 0.0f;
            if (last && m >= P / 2) tail_wr[m - P / 2] = v;   // becomes the next call's x(n-1)
        }
        return v;
    };
    auto lds_dst = [&](int i, float2 v) { s[phys(i)] = v; };
    fft_passes<LOG2P, false, false, true>(s, f.tw, tid, load, lds_dst);
    __syncthreads();
    const int slot = ring_slot(job.slot0, b, job.ring);
    float2* row = job.fdl + ((size_t)c * job.ring + slot) * P;
    split_and_store<LOG2P>(s, f.tw, tid, row, 1.0f);
}

// ---------------------------------------------------------------------------
// K0: filter partitions -> spectra.  grid (K, data paths)
// ---------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(threads_for(1 << LOG2P)) void filter_kernel(const float* __restrict__ taps,
                                                                         float2* __restrict__ H, int K,
                                                                         const float2* __restrict__ tw) {
    constexpr int P = 1 << LOG2P;
    __shared__ float2 s[lds_elems(P)];
    const int j = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const float2* __restrict__ part = reinterpret_cast<const float2*>(taps + ((size_t)d * K + j) * P);
    auto load = [&](int m) -> float2 { return (m < P / 2) ? part[m] : float2{0.0f, 0.0f}; };   // [h_j | 0]
    auto lds_dst = [&](int i, float2 v) { s[phys(i)] = v; };
    fft_passes<LOG2P, false, false, true>(s, tw, tid, load, lds_dst);
    __syncthreads();
    split_and_store<LOG2P>(s, tw, tid, H + ((size_t)d * K + j) * P, 0.5f / (float)P);
}

// ---------------------------------------------------------------------------
// K3: inverse.  grid (max blocks per stream, output channels, streams)
// ---------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(threads_for(1 << LOG2P)) void inverse_kernel(const StreamJob* __restrict__ jobs,
                                                                          FilterDev f,
                                                                          const float2* __restrict__ Y) {
    constexpr int P = 1 << LOG2P;
    constexpr int NT = threads_for(P);
    __shared__ float2 s[lds_elems(P)];
    const StreamJob job = jobs[blockIdx.z];
    const int b = blockIdx.x;
    if (b >= job.nblocks) return;
    const int o = blockIdx.y;
    const int tid = threadIdx.x;
    const int cout = f.cout;
    const float2* __restrict__ y = Y + ((size_t)job.yunit0 + (size_t)o * job.nblocks + b) * P;
    const float2* __restrict__ tw = f.tw;

    // Hermitian fold: Z[k] = E[k] + i O[k], E = Y[k] + conj Y[P-k], O = (Y[k] - conj Y[P-k]) W^-k
    for (int k = tid; k < P / 2; k += NT) {
        if (k == 0) {
            const float2 y0 = y[0];                       // (DC, Nyquist)
            s[phys(0)] = float2{y0.x + y0.y, y0.x - y0.y};
            const float2 yh = y[P / 2];
            s[phys(P / 2)] = float2{2.0f * yh.x, -2.0f * yh.y};
        } else {
            const float2 a = y[k], bb = y[P - k];
            const float2 e = float2{a.x + bb.x, a.y - bb.y};
            const float2 dd = float2{a.x - bb.x, a.y + bb.y};
            const float2 oo = cmulc(dd, tw[k]);           // * exp(+i*pi*k/P)
            s[phys(k)] = float2{e.x - oo.y, e.y + oo.x};
            s[phys(P - k)] = float2{e.x + oo.y, -e.y + oo.x};
        }
    }
    __syncthreads();

    float* __restrict__ out = job.out;
    const long long fb = (long long)b * P;
    float pk_s = 0.0f, pk_a = 0.0f;
    auto lds_src = [&](int i) { return s[phys(i)]; };
    // z[q] = (y[2q], y[2q+1]); overlap-save keeps samples P..2P-1 (q >= P/2)
    auto store = [&](int q, float2 z) {
        if (q >= P / 2) {
            const long long fr = fb + 2 * q - P;
            if (fr < job.nframes) {
                out[fr * cout + o] = z.x;
                pk_s = fmaxf(pk_s, z.x);
                pk_a = fmaxf(pk_a, fabsf(z.x));
            }
            if (fr + 1 < job.nframes) {
                out[(fr + 1) * cout + o] = z.y;
                pk_s = fmaxf(pk_s, z.y);
                pk_a = fmaxf(pk_a, fabsf(z.y));
            }
        }
    };
    fft_passes<LOG2P, true, true, false>(s, tw, tid, lds_src, store);

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pk_s = fmaxf(pk_s, __shfl_xor(pk_s, off, 64));
        pk_a = fmaxf(pk_a, __shfl_xor(pk_a, off, 64));
    }
    if ((tid & 63) == 0) {
        // non-negative floats order like their bit patterns
        atomicMax(job.peaks + 0, __float_as_uint(pk_s));
        atomicMax(job.peaks + 1, __float_as_uint(pk_a));
    }
}

// ---------------------------------------------------------------------------
// K2: multiply-accumulate.  grid (bin-pair tiles, outputs * time tiles, streams)
// Each thread owns two adjacent bins (16 B) and TT consecutive output blocks of
// one (stream, output channel); X rows are streamed once per time tile.
// ---------------------------------------------------------------------------
template <int TT>
__global__ __launch_bounds__(256) void mac_kernel(const StreamJob* __restrict__ jobs, FilterDev f,
                                                  float2* __restrict__ Y, int tiles) {
    const StreamJob job = jobs[blockIdx.z];
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
        const uint64_t mlo = (uint64_t)f.mask[pth.data * 4 + 0] | ((uint64_t)f.mask[pth.data * 4 + 1] << 32);
        const uint64_t mhi = (uint64_t)f.mask[pth.data * 4 + 2] | ((uint64_t)f.mask[pth.data * 4 + 3] << 32);
        for (int u = 0; u < K - 1 + TT; ++u) {
            const int rel = t0 - (K - 1) + u;             // input block, relative to the call's first
            const int slot = ring_slot(job.slot0, rel, ring);
            const float4 x = X[(size_t)slot * P2];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int j = tt + K - 1 - u;
                if (j >= 0 && j < K) {
                    const bool on = (j < 64) ? ((mlo >> j) & 1) : ((mhi >> (j - 64)) & 1);
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
    // Packed bin 0 = (DC, Nyquist): two real products, not a complex one.
    if (blockIdx.x == 0 && threadIdx.x < TT && t0 + (int)threadIdx.x < job.nblocks) {
        const int tt = threadIdx.x;
        float re = 0.f, im = 0.f;
        for (int pe = pe0; pe < pe1; ++pe) {
            const PathEntry pth = f.paths[pe];
            const float2* __restrict__ Hd = f.H + (size_t)pth.data * K * P;
            const float2* __restrict__ X = job.fdl + (size_t)pth.in_ch * ring * P;
            const uint64_t mlo = (uint64_t)f.mask[pth.data * 4 + 0] | ((uint64_t)f.mask[pth.data * 4 + 1] << 32);
            const uint64_t mhi = (uint64_t)f.mask[pth.data * 4 + 2] | ((uint64_t)f.mask[pth.data * 4 + 3] << 32);
            // same accumulation order as the main loop: oldest input block first
            for (int j = K - 1; j >= 0; --j) {
                const bool on = (j < 64) ? ((mlo >> j) & 1) : ((mhi >> (j - 64)) & 1);
                if (!on) continue;
                const int slot = ring_slot(job.slot0, t0 + tt - j, ring);
                const float2 x = X[(size_t)slot * P];
                const float2 h = Hd[(size_t)j * P];
                re = fmaf(x.x, h.x, re);
                im = fmaf(x.y, h.y, im);
            }
        }
        Y[(yrow0 + tt) * P] = float2{re, im};
    }
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
        default: return hipErrorInvalidValue;
    }
}

template <int L>
struct FwdLaunch {
    static hipError_t run(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, hipStream_t st) {
        dim3 grid(max_blocks, f.cin, njobs), block(threads_for(1 << L));
        hipLaunchKernelGGL(forward_kernel<L>, grid, block, 0, st, jobs, f);
        return hipGetLastError();
    }
};
template <int L>
struct InvLaunch {
    static hipError_t run(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, const float2* Y,
                          hipStream_t st) {
        dim3 grid(max_blocks, f.cout, njobs), block(threads_for(1 << L));
        hipLaunchKernelGGL(inverse_kernel<L>, grid, block, 0, st, jobs, f, Y);
        return hipGetLastError();
    }
};
template <int L>
struct FilterLaunch {
    static hipError_t run(const float* taps, float2* H, int ndata, int K, const float2* tw, hipStream_t st) {
        dim3 grid(K, ndata), block(threads_for(1 << L));
        hipLaunchKernelGGL(filter_kernel<L>, grid, block, 0, st, taps, H, K, tw);
        return hipGetLastError();
    }
};

}  // namespace

hipError_t launch_forward(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, hipStream_t st) {
    return dispatch_log2p<FwdLaunch>(f.log2P, jobs, njobs, max_blocks, f, st);
}

hipError_t launch_inverse(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, const float2* Y,
                          hipStream_t st) {
    return dispatch_log2p<InvLaunch>(f.log2P, jobs, njobs, max_blocks, f, Y, st);
}

hipError_t launch_filter_transform(const float* taps, float2* H, int ndata, int K, int log2P, const float2* tw,
                                   hipStream_t st) {
    return dispatch_log2p<FilterLaunch>(log2P, taps, H, ndata, K, tw, st);
}

hipError_t launch_mac(const StreamJob* jobs, int njobs, int max_blocks, const FilterDev& f, float2* Y, int time_tile,
                      hipStream_t st) {
    const int P2 = f.P / 2;
    const int nt = P2 < 256 ? P2 : 256;
    int tt = 1;
    while (tt * 2 <= time_tile && tt < 16) tt *= 2;
    const int tiles = (max_blocks + tt - 1) / tt;
    dim3 grid(P2 / nt, f.cout * tiles, njobs), block(nt);
    switch (tt) {
        case 1: hipLaunchKernelGGL(mac_kernel<1>, grid, block, 0, st, jobs, f, Y, tiles); break;
        case 2: hipLaunchKernelGGL(mac_kernel<2>, grid, block, 0, st, jobs, f, Y, tiles); break;
        case 4: hipLaunchKernelGGL(mac_kernel<4>, grid, block, 0, st, jobs, f, Y, tiles); break;
        case 8: hipLaunchKernelGGL(mac_kernel<8>, grid, block, 0, st, jobs, f, Y, tiles); break;
        default: hipLaunchKernelGGL(mac_kernel<16>, grid, block, 0, st, jobs, f, Y, tiles); break;
    }
    return hipGetLastError();
}

}  // namespace fk
