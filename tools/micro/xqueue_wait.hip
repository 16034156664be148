// Microbenchmark (round 5): what a cross-queue dependency costs on this platform — a copy stream (SDMA) and a compute
// stream tied by hipEventRecord / hipStreamWaitEvent, as the engine's DMA pipelines are.  A chain of N links
// [copy 4 MB host -> device on stream A] -> [tiny kernel on stream B] -> [copy 4 MB device -> host on stream C] ...,
// against the same N copies and kernels with no waits between the streams.  Event flags: DisableTiming, default.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xqueue_wait.hip -o /tmp/xqueue_wait && /tmp/xqueue_wait
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t piece = (size_t)4 << 20;
    const int N = 64;
    char *h, *d;
    CK(hipHostMalloc((void**)&h, piece * 2, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipMalloc((void**)&d, piece * 2));
    memset(h, 0, piece * 2);
    hipStream_t a, b, c;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    for (int flags_i = 0; flags_i < 2; ++flags_i) {
        const unsigned flags = flags_i == 0 ? hipEventDisableTiming : hipEventDefault;
        hipEvent_t e1[N], e2[N];
        for (int i = 0; i < N; ++i) { CK(hipEventCreateWithFlags(&e1[i], flags)); CK(hipEventCreateWithFlags(&e2[i], flags)); }
        for (int mode = 0; mode < 4; ++mode) {
            // 0: no waits (three independent streams)   1: copy-in -> kernel   2: kernel -> copy-out   3: both (the pipeline's chain)
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) {
                    CK(hipMemcpyAsync(d, h, piece, hipMemcpyHostToDevice, a));
                    if (mode == 1 || mode == 3) { CK(hipEventRecord(e1[i], a)); CK(hipStreamWaitEvent(b, e1[i], 0)); }
                    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, b, (float*)d);
                    if (mode == 2 || mode == 3) { CK(hipEventRecord(e2[i], b)); CK(hipStreamWaitEvent(c, e2[i], 0)); }
                    CK(hipMemcpyAsync(h + piece, d + piece, piece, hipMemcpyDeviceToHost, c));
                }
                CK(hipDeviceSynchronize());
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("events %-14s %-28s %7.2f ms for %d links = %6.1f us per link (4 MB in + kernel + 4 MB out)\n", flags_i == 0 ? "DisableTiming" : "default",
                   mode == 0 ? "no waits" : mode == 1 ? "copy-in -> kernel" : mode == 2 ? "kernel -> copy-out" : "copy-in -> kernel -> copy-out", best * 1e3, N, best * 1e6 / N);
        }
    }
    // the same chain with both "copies" done by a kernel on the compute stream's own queue kind (no SDMA): an upper bound of what queue-local waits cost
    return 0;
}
