// Microbenchmark: does the Infinity Cache absorb writes?  Fill / read / copy over footprints from 32 MB to 2 GB,
// looped, bytes per second by HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const v4f* a, v4f* out, size_t n) {
    v4f s{0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
    if (s.x == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256) void wr(v4f* b, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v4f{v, v, v, v};
}
__global__ __launch_bounds__(256) void cp(const v4f* a, v4f* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main() {
    v4f *a, *b; hipMalloc(&a, (size_t)2 << 30); hipMalloc(&b, (size_t)2 << 30); hipMemset(a, 1, (size_t)2 << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {32, 64, 96, 128, 192, 256, 512, 2048}) {
        const size_t n = (mb << 20) / 16;
        float t[4];
        for (int mode = 0; mode < 4; ++mode) {
            const int N = (int)(40000 / mb) + 10;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < N; ++i) {
                    if (mode == 0) hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, a, b, n);
                    else if (mode == 1) hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, b, n, 1.5f);
                    else if (mode == 2) hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, 0, a, b, n);
                    else { hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, b, n, 1.5f); hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, b, a, n); }
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&t[mode], e0, e1); t[mode] /= N;
            }
        }
        const double B = (double)(mb << 20);
        printf("%4zu MB: read %.2f TB/s, write %.2f TB/s, copy %.2f TB/s (r+w), write-then-read %.2f TB/s (r+w)\n", mb, B / t[0] / 1e9, B / t[1] / 1e9,
               2 * B / t[2] / 1e9, 2 * B / t[3] / 1e9);
    }
    return 0;
}
