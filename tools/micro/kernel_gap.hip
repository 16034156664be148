// Microbenchmark: what a DEPENDENT kernel boundary costs on MI355X as a function of the bytes the first kernel wrote and
// of the cache policy of its stores.  A (writes `mb` MB) -> B (reads them) on one stream; every workgroup stamps
// s_memrealtime at entry and exit (atomicMin / atomicMax), so durations and the gap are on the GPU's own clock (100 MHz).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/kernel_gap.hip -o /tmp/kernel_gap && /tmp/kernel_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long ull;

__device__ __forceinline__ ull now() { return __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime(); }

template <int POL>
__global__ __launch_bounds__(256) void wr(v4f* b, size_t n, float v, ull* st) {
    if (threadIdx.x == 0) atomicMin(st + 0, now());
    const v4f val{v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4f* p = b + i;
        if constexpr (POL == 0) *p = val;
        if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(val) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) atomicMax(st + 1, now());
}
__global__ __launch_bounds__(256) void rd(const v4f* a, v4f* out, size_t n, ull* st) {
    if (threadIdx.x == 0) atomicMin(st + 2, now());
    v4f s{0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
    if (s.x == 12345.f) out[0] = s;
    if (threadIdx.x == 0) atomicMax(st + 3, now());
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int POL>
void run(const char* name, size_t mb, int wgs, v4f* buf, v4f* sink, ull* st, ull* hst) {
    const size_t n = (mb << 20) / 16;
    std::vector<double> da, gap, db;
    for (int it = 0; it < 24; ++it) {
        hst[0] = hst[2] = ~0ull; hst[1] = hst[3] = 0;
        CK(hipMemcpy(st, hst, 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(wr<POL>, dim3(wgs), dim3(256), 0, 0, buf, n, (float)it, st);
        hipLaunchKernelGGL(rd, dim3(wgs), dim3(256), 0, 0, buf, sink, n, st);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hst, st, 32, hipMemcpyDeviceToHost));
        if (it < 4) continue;
        da.push_back((hst[1] - hst[0]) * 0.01); gap.push_back(((long long)hst[2] - (long long)hst[1]) * 0.01); db.push_back((hst[3] - hst[2]) * 0.01);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-14s %4zu MB, %4d WGs: write %7.2f us   gap %6.2f us   read %7.2f us\n", name, mb, wgs, med(da), med(gap), med(db));
}

int main() {
    v4f *buf, *sink; ull *st, hst[4];
    CK(hipMalloc(&buf, (size_t)256 << 20)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&st, 64));
    for (size_t mb : {1, 4, 16, 32, 64, 128}) {
        for (int wgs : {256, 2048}) {
            run<0>("plain", mb, wgs, buf, sink, st, hst);
            run<1>("nt", mb, wgs, buf, sink, st, hst);
            run<3>("sc1", mb, wgs, buf, sink, st, hst);
            run<4>("sc0 sc1", mb, wgs, buf, sink, st, hst);
            run<5>("sc0 sc1 nt", mb, wgs, buf, sink, st, hst);
        }
    }
    return 0;
}
