// Microbenchmark: three ways to time ONE kernel of a dependent chain of three on a stream, against each other:
//   (a) hipEventRecord before / after the launch (what the engine's profiling did up to round 5): event to event, holds
//       the launch boundary behind the kernel
//   (b) hipExtLaunchKernelGGL(kernel, ..., startEvent, stopEvent, 0, ...): events bound to the dispatch itself — the
//       command processor's begin / end stamps of the packet, the figures rocprofv3's kernel trace prints
//   (c) s_memrealtime (100 MHz) stamped by every wavefront at entry and exit (atomicMin / atomicMax): first wave in to
//       last wave out, on the GPU's own clock
// for streaming kernels of ~10 .. 600 us (the engine's K1 / K2 / K3 at cfg2 .. cfg3 sizes).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/ext_launch_timing.hip -o /tmp/elt && /tmp/elt
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long ull;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copyk(const v4f* a, v4f* b, size_t n, ull* st) {
    // (stamps are spread over 1024 slots 64 bytes apart: 8192 wavefronts adding into ONE address serialise at ~20 ns each
    // — 170 us on top of a 7 us kernel, measured with this file's first version)
    if (st && (threadIdx.x & 63) == 0) atomicMin(st + (blockIdx.x & 1023) * 8, (ull)__builtin_amdgcn_s_memrealtime());
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
    if (st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) atomicMax(st + (blockIdx.x & 1023) * 8 + 1, (ull)__builtin_amdgcn_s_memrealtime());
    }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const size_t maxb = (size_t)1 << 30;
    v4f *a, *b, *c, *d;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&c, maxb)); CK(hipMalloc(&d, maxb));
    CK(hipMemset(a, 0, maxb));
    const size_t SL = 1024 * 8;                      // ull per kernel: 1024 slots of (min start, max end), 64 bytes apart
    ull* st;
    std::vector<ull> hst(3 * SL);
    CK(hipMalloc(&st, 3 * SL * sizeof(ull)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t ev[4], xs[3], xe[3];
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (auto& e : xs) CK(hipEventCreate(&e));
    for (auto& e : xe) CK(hipEventCreate(&e));
    printf("# chain of three dependent copies (a->b, b->c, c->d) on one stream; per kernel, median of 30, microseconds\n");
    printf("# %8s | %26s | %26s | %26s | %s\n", "MB each", "(a) event to event", "(b) hipExtLaunch start/stop", "(c) in-kernel s_memrealtime", "wall per chain (no timing)");
    for (size_t mb : {4, 8, 16, 32, 64, 128, 256, 512, 1024}) {
        const size_t n = (mb << 20) / 16;
        const int grid = 2048;
        std::vector<double> A[3], B[3], C[3], B2[3], G[2];
        for (int it = 0; it < 36; ++it) {
            // (a)
            CK(hipEventRecord(ev[0], s));
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, a, b, n, (ull*)nullptr);
            CK(hipEventRecord(ev[1], s));
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, b, c, n, (ull*)nullptr);
            CK(hipEventRecord(ev[2], s));
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, c, d, n, (ull*)nullptr);
            CK(hipEventRecord(ev[3], s));
            CK(hipStreamSynchronize(s));
            for (int k = 0; k < 3; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); if (it >= 6) A[k].push_back(ms * 1e3); }
            // (b): the same three launches, events bound to the dispatches
            hipExtLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, xs[0], xe[0], 0, a, b, n, (ull*)nullptr);
            hipExtLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, xs[1], xe[1], 0, b, c, n, (ull*)nullptr);
            hipExtLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, xs[2], xe[2], 0, c, d, n, (ull*)nullptr);
            CK(hipStreamSynchronize(s));
            // (c): once more, stamping
            for (size_t i = 0; i < 3 * SL; i += 8) { hst[i] = ~0ull; hst[i + 1] = 0; }
            CK(hipMemcpy(st, hst.data(), 3 * SL * sizeof(ull), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, a, b, n, st + 0 * SL);
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, b, c, n, st + 1 * SL);
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, c, d, n, st + 2 * SL);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(hst.data(), st, 3 * SL * sizeof(ull), hipMemcpyDeviceToHost));
            ull t0[3], t1[3];
            for (int k = 0; k < 3; ++k) {
                t0[k] = ~0ull; t1[k] = 0;
                for (size_t i = 0; i < SL; i += 8) { t0[k] = std::min(t0[k], hst[k * SL + i]); t1[k] = std::max(t1[k], hst[k * SL + i + 1]); }
            }
            for (int k = 0; k < 3; ++k) {
                float ms = -1.f;
                hipError_t e = hipEventElapsedTime(&ms, xs[k], xe[k]);
                if (e != hipSuccess) { (void)hipGetLastError(); ms = -1.f; }
                float ms2 = -1.f;                                             // stop(k) -> stop(k+1): end to end of neighbours
                if (k < 2) { e = hipEventElapsedTime(&ms2, xe[k], xe[k + 1]); if (e != hipSuccess) { (void)hipGetLastError(); ms2 = -1.f; } }
                if (it >= 6) { B[k].push_back(ms * 1e3); B2[k].push_back(ms2 * 1e3); C[k].push_back((t1[k] - t0[k]) * 0.01); if (k < 2) G[k].push_back(((long long)t0[k + 1] - (long long)t1[k]) * 0.01); }
            }
        }
        // wall per chain without any timing machinery
        CK(hipStreamSynchronize(s));
        hipEvent_t w0 = ev[0], w1 = ev[1];
        CK(hipEventRecord(w0, s));
        for (int it = 0; it < 50; ++it) {
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, a, b, n, (ull*)nullptr);
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, b, c, n, (ull*)nullptr);
            hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, s, c, d, n, (ull*)nullptr);
        }
        CK(hipEventRecord(w1, s));
        CK(hipStreamSynchronize(s));
        float wms; CK(hipEventElapsedTime(&wms, w0, w1));
        printf("  %8zu | %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f | %8.2f   (stop->next stop: %.2f %.2f; in-kernel gaps %.2f %.2f)\n", mb,
               med(A[0]), med(A[1]), med(A[2]), med(B[0]), med(B[1]), med(B[2]), med(C[0]), med(C[1]), med(C[2]), wms * 1e3 / 50,
               med(B2[0]), med(B2[1]), med(G[0]), med(G[1]));
    }
    return 0;
}
