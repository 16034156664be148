// walk_common.hpp — what the translation units of K2's whole-call walk share (kernels.hip: the four-FMA walk and every
// other kernel; mac_walk3.hip: the three-FMA walk): the stream-descriptor reference, ring arithmetic, static loops over
// template indices and the DPP moves of the lane groups.  Internal to kernels/: nothing here is part of kernels.h.
#pragma once

#include <cxxabi.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include <type_traits>
#include <utility>

#include "fft_core.hpp"
#include "kernels.h"

namespace fk {

// launch shape of a whole-call walk: rows of G per lane, lanes per bin, time tiles (and their length), path sets per group
struct WalkShape { int kr, lpb, tiles, tile_len, np; };


namespace {

// Where a kernel finds its stream descriptors: an array indexed by blockIdx.z, or — a launch of ONE stream — the
// descriptor itself, among the kernel arguments (Tuning::one_job).  A one-stream call then needs no upload in front of
// K1 (a copy kernel of 3.6 us behind a 5.8 us dependency gap: 9.5 of the 68 us of a 256-block call of one stereo stream)
// and the latency kernels no dependent read over the bus.
struct JobRef {
    const StreamJob* jobs;
    StreamJob one;
};
__device__ __forceinline__ StreamJob fetch_job(const JobRef& r) { return r.jobs ? r.jobs[blockIdx.z] : r.one; }
inline JobRef make_job_ref(const StreamJob* jobs, const Tuning& tn) {
    return tn.one_job ? JobRef{nullptr, *tn.one_job} : JobRef{jobs, StreamJob{}};
}

__device__ __forceinline__ int ring_slot(int slot0, int rel, int ring) {
    int s = (slot0 + rel) % ring;
    return s < 0 ? s + ring : s;
}


template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// f(0) && f(1) && ... && f(N-1): straight-line code with an exit after every step, no joins inside
template <int... I, class F>
__device__ __forceinline__ bool static_all_impl(std::integer_sequence<int, I...>, F&& f) {
    return (f(std::integral_constant<int, I>{}) && ...);
}
template <int N, class F>
__device__ __forceinline__ bool static_all(F&& f) {
    return static_all_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ float dpp_row_shr1(float v) {
    // (bound_ctrl: lanes without a source read 0 — no `old` operand to materialise; those lanes' results are never used)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}


// The instantiation a launcher chose, spelled as rocprofv3 spells it — the device symbol's name (hipKernelNameRefByPtr),
// demangled, without namespaces and arguments: "mac_walk_kernel<33, 7, true, 4, 1, 1>" — noted per role (0 = K1, 1 = K2,
// 2 = K3) in Tuning::names when the engine asks for it.  Looked up once per instantiation.
template <auto Kern>
inline void note_kernel(const Tuning& tn, int role) {
    if (!tn.names) return;
    static const std::string name = [] {
        const char* m = hipKernelNameRefByPtr(reinterpret_cast<const void*>(Kern), nullptr);
        if (!m) return std::string("?");
        int status = 0;
        char* d = abi::__cxa_demangle(m, nullptr, nullptr, &status);
        std::string full = (status == 0 && d) ? d : m;
        free(d);
        // "void fk::(anonymous namespace)::mac_kernel<1>(fk::(anonymous namespace)::JobRef, ...)": cut the arguments (the
        // parenthesis that closes last, matched backwards), then everything up to the last "::" in front of the template's name
        size_t end = full.size();
        if (end && full[end - 1] == ')') {
            int depth = 0;
            for (size_t i = end; i-- > 0;) {
                if (full[i] == ')') ++depth;
                else if (full[i] == '(' && --depth == 0) { end = i; break; }
            }
        }
        full.resize(end);
        const size_t lt = full.find('<');
        const size_t stop = lt == std::string::npos ? full.size() : lt;
        size_t start = 0;
        for (size_t i = 0; i + 1 < stop; ++i)
            if (full[i] == ':' && full[i + 1] == ':') start = i + 2;
        if (start == 0) { const size_t sp = full.rfind(' ', stop); if (sp != std::string::npos) start = sp + 1; }
        return full.substr(start);
    }();
    snprintf(tn.names->k[role], sizeof tn.names->k[role], "%s", name.c_str());
}
// (Tuning::kev, profiling: the launch carries a start and a stop event BOUND TO THE DISPATCH — hipExtLaunchKernelGGL —,
// whose elapsed time is the command processor's begin-to-end of this one packet: what rocprofv3's kernel trace prints,
// without the launch boundary an event recorded behind the kernel includes; tools/micro/ext_launch_timing.hip)
#define FK_LAUNCH(role, kern, grid, block, st, ...)                                                                  \
    do {                                                                                                             \
        note_kernel<kern>(tn, role);                                                                                 \
        if (tn.kev) hipExtLaunchKernelGGL(kern, grid, block, 0, st, tn.kev[role][0], tn.kev[role][1], 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, 0, st, __VA_ARGS__);                                              \
    } while (0)

}  // namespace

// mac_walk3.hip: the three-FMA form of the walk.  False if (kr, lpb, np) has no instantiation there.
bool walk3_has(int kr, int lpb, int np);
hipError_t launch_walk3(const StreamJob* jobs, int njobs, const FilterDev& f, float2* Y, const WalkShape& w, const Tuning& tn,
                        hipStream_t st);

}  // namespace fk
