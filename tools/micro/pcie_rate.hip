// Microbenchmark (round 5): what the bus gives hipMemcpyAsync between page-locked host memory and the device — one
// direction, both at once, with 1 / 2 / 4 HIP streams per direction (does a second copy stream engage a second SDMA
// engine?), in pieces of 1 / 4 / 16 MB.  The engine's duplex pipeline uses ONE copy stream per direction.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pcie_rate.hip -o /tmp/pcie_rate && /tmp/pcie_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t total = (size_t)1 << 30;
    char *hin, *hout, *din, *dout;
    CK(hipHostMalloc((void**)&hin, total, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipHostMalloc((void**)&hout, total, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipMalloc((void**)&din, total));
    CK(hipMalloc((void**)&dout, total));
    memset(hin, 1, total); memset(hout, 0, total);
    hipStream_t si[4], so[4];
    for (int i = 0; i < 4; ++i) { CK(hipStreamCreateWithFlags(&si[i], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&so[i], hipStreamNonBlocking)); }
    for (size_t piece : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20}) {
        for (int ns : {1, 2, 4}) {
            for (int mode = 0; mode < 3; ++mode) {       // 0 in, 1 out, 2 both
                double best = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipDeviceSynchronize());
                    const auto t0 = std::chrono::steady_clock::now();
                    size_t k = 0;
                    for (size_t off = 0; off < total; off += piece, ++k) {
                        if (mode != 1) CK(hipMemcpyAsync(din + off, hin + off, piece, hipMemcpyHostToDevice, si[k % ns]));
                        if (mode != 0) CK(hipMemcpyAsync(hout + off, dout + off, piece, hipMemcpyDeviceToHost, so[k % ns]));
                    }
                    CK(hipDeviceSynchronize());
                    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    best = std::max(best, (double)total / dt / 1e9);
                }
                printf("pieces of %2zu MB, %d stream(s) per direction, %s: %.1f GB/s%s\n", piece >> 20, ns,
                       mode == 0 ? "host -> device" : mode == 1 ? "device -> host" : "both at once  ", best, mode == 2 ? " each way" : "");
            }
        }
    }
    return 0;
}
