// Microbenchmark: socket power and shader clock (amdgpu hwmon) under pure HBM reads, pure writes and a copy,
// each looped for ~2 s.  Is the write-heavy K1 the power hog because of its stores?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <glob.h>
#include <chrono>
#include <thread>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rd(const v4f* a, v4f* out, size_t n) {
    v4f s{0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += __builtin_nontemporal_load(a + i);
    if (s.x == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256) void wr(v4f* b, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v4f{v, v, v, v};
}
// stores with cache-policy bits: 1 = nt, 2 = sc0, 3 = sc1, 4 = sc0 sc1, 5 = sc0 sc1 nt
template <int POL>
__global__ __launch_bounds__(256) void wrp(v4f* b, size_t n, float v) {
    const v4f val{v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4f* p = b + i;
        if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
        if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(val) : "memory");
    }
}
__global__ __launch_bounds__(256) void cp(const v4f* a, v4f* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

static long rdl(const std::string& p) { FILE* f = fopen(p.c_str(), "r"); if (!f) return -1; long v = -1; if (fscanf(f, "%ld", &v) != 1) v = -1; fclose(f); return v; }
static void sample(long& pw, long& fr) {
    glob_t g; pw = 0; fr = 0;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc; ++i) {
            std::string p = g.gl_pathv[i]; long v = rdl(p);
            if (v > pw) { pw = v; fr = rdl(p.substr(0, p.rfind('/')) + "/freq1_input"); }
        }
        globfree(&g);
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    v4f *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 8; ++mode) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, a, b, n);
            else if (mode == 1) hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, b, n, 1.5f);
            else if (mode == 2) hipLaunchKernelGGL(cp, dim3(4096), dim3(256), 0, 0, a, b, n);
            else if (mode == 3) hipLaunchKernelGGL(wrp<1>, dim3(4096), dim3(256), 0, 0, b, n, 1.5f);
            else if (mode == 4) hipLaunchKernelGGL(wrp<2>, dim3(4096), dim3(256), 0, 0, b, n, 1.5f);
            else if (mode == 5) hipLaunchKernelGGL(wrp<3>, dim3(4096), dim3(256), 0, 0, b, n, 1.5f);
            else if (mode == 6) hipLaunchKernelGGL(wrp<4>, dim3(4096), dim3(256), 0, 0, b, n, 1.5f);
            else hipLaunchKernelGGL(wrp<5>, dim3(4096), dim3(256), 0, 0, b, n, 1.5f);
        };
        for (int i = 0; i < 20; ++i) launch();
        hipDeviceSynchronize();
        const int N = 2500;
        hipEventRecord(e0);
        for (int i = 0; i < N; ++i) launch();
        hipEventRecord(e1);
        std::this_thread::sleep_for(std::chrono::milliseconds(700));
        long pw[5], fr[5];
        for (int k = 0; k < 5; ++k) { sample(pw[k], fr[k]); std::this_thread::sleep_for(std::chrono::milliseconds(120)); }
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double moved = (mode == 2 ? 2.0 : 1.0) * bytes;
        printf("%s: %.3f ms per pass, %.2f TB/s; power %ld %ld %ld W, sclk %ld %ld %ld MHz\n", mode == 0 ? "read " : mode == 1 ? "write" : mode == 2 ? "copy " : mode == 3 ? "write nt" : mode == 4 ? "write sc0" : mode == 5 ? "write sc1" : mode == 6 ? "write sc0 sc1" : "write sc0 sc1 nt",
               ms / N, moved / (ms / N) / 1e9, pw[1] / 1000000, pw[2] / 1000000, pw[4] / 1000000, fr[1] / 1000000, fr[2] / 1000000, fr[4] / 1000000);
    }
    return 0;
}
