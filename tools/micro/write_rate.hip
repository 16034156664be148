// Microbenchmark: HBM write rate against the width of the store (4 / 8 / 16 bytes per lane), the number of workgroups and
// the address pattern (grid-stride: consecutive workgroups write consecutive kilobytes; chunked: every workgroup its own
// contiguous region).  2 GiB per pass, HIP events.  The engine's own probe (fe_engine_hbm_rates) says 4.0 - 4.7 TB/s for
// 16-byte grid-stride stores; MI355X_MICROARCH.md quotes 6.0 - 6.2 TB/s for dword stores of 2 304-byte rows.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/write_rate.hip -o /tmp/write_rate && /tmp/write_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T> __device__ T val(float v);
template <> __device__ float val<float>(float v) { return v; }
template <> __device__ v2f val<v2f>(float v) { return v2f{v, v}; }
template <> __device__ v4f val<v4f>(float v) { return v4f{v, v, v, v}; }

template <class T, bool CHUNK>
__global__ __launch_bounds__(256) void wr(T* b, size_t n, float v) {
    const T x = val<T>(v);
    if (CHUNK) {
        const size_t per = n / gridDim.x;
        T* p = b + (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; i < per; i += 256) p[i] = x;
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = x;
    }
}

template <class T, bool CHUNK>
__global__ __launch_bounds__(256) void rd(const T* b, T* sink, size_t n) {
    T acc = val<T>(0.f);
    if (CHUNK) {
        const size_t per = n / gridDim.x;
        const T* p = b + (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; i < per; i += 256) acc += p[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += b[i];
    }
    if (((const float*)&acc)[0] == 12345.678f) sink[0] = acc;
}
template <class T, bool CHUNK>
double run_rd(void* buf, size_t bytes, int wgs) {
    const size_t n = bytes / sizeof(T);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rd<T, CHUNK>), dim3(wgs), dim3(256), 0, 0, (const T*)buf, (T*)buf, n);
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rd<T, CHUNK>), dim3(wgs), dim3(256), 0, 0, (const T*)buf, (T*)buf, n);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)bytes * reps / (ms * 1e-3) / 1e12;
}

template <class T, bool CHUNK>
double run(void* buf, size_t bytes, int wgs) {
    const size_t n = bytes / sizeof(T);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((wr<T, CHUNK>), dim3(wgs), dim3(256), 0, 0, (T*)buf, n, 1.f);
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((wr<T, CHUNK>), dim3(wgs), dim3(256), 0, 0, (T*)buf, n, (float)r);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)bytes * reps / (ms * 1e-3) / 1e12;
}

int main() {
    void* buf;
    const size_t bytes = (size_t)2 << 30;
    CK(hipMalloc(&buf, bytes));
    for (int wgs : {1024, 2048, 4096, 8192, 16384}) {
        printf("%5d WGs  grid-stride: 4 B %5.2f   8 B %5.2f   16 B %5.2f TB/s    chunked: 4 B %5.2f   8 B %5.2f   16 B %5.2f TB/s\n", wgs,
               run<float, false>(buf, bytes, wgs), run<v2f, false>(buf, bytes, wgs), run<v4f, false>(buf, bytes, wgs),
               run<float, true>(buf, bytes, wgs), run<v2f, true>(buf, bytes, wgs), run<v4f, true>(buf, bytes, wgs));
    }
    for (int wgs : {1024, 2048, 4096, 8192, 16384}) {
        printf("%5d WGs  READ grid-stride: 8 B %5.2f   16 B %5.2f TB/s    chunked: 8 B %5.2f   16 B %5.2f TB/s\n", wgs,
               run_rd<v2f, false>(buf, bytes, wgs), run_rd<v4f, false>(buf, bytes, wgs),
               run_rd<v2f, true>(buf, bytes, wgs), run_rd<v4f, true>(buf, bytes, wgs));
    }
    return 0;
}
