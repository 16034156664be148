// Microbenchmark: the best float4 COPY rate this GPU gives (the yardstick /opt/skills/guides/MI355X_MICROARCH.md:36 quotes as
// 6.29 TB/s = 79 % of 8), swept over what a copy kernel can choose:
//   workgroups per CU (1 .. 16 of 256 threads), 16 bytes per lane,
//   bytes in flight per wave between a read burst and a write burst (U = 1, 4, 8, 16 loads of 1 KiB per wave-instruction),
//   load / store cache policy (plain or nontemporal, each side),
//   address pattern (grid-stride front, or every workgroup its own contiguous region).
// 1 GiB -> 1 GiB per pass (far beyond the 256 MiB Infinity Cache), 12 passes, HIP events; rate = bytes read + written.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/copy_rate.hip -o /tmp/copy_rate && /tmp/copy_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NTL, bool NTS, bool CHUNK>
__global__ __launch_bounds__(256) void copyk(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    const size_t tile = (size_t)256 * U;                 // elements a workgroup moves per step
    const size_t tiles = n / tile;
    size_t t0, t1, dt;
    if (CHUNK) {
        const size_t per = (tiles + gridDim.x - 1) / gridDim.x;
        t0 = (size_t)blockIdx.x * per; t1 = t0 + per < tiles ? t0 + per : tiles; dt = 1;
    } else {
        t0 = blockIdx.x; t1 = tiles; dt = gridDim.x;
    }
    for (size_t t = t0; t < t1; t += dt) {
        const v4f* p = a + t * tile + threadIdx.x;
        v4f* q = b + t * tile + threadIdx.x;
        v4f r[U];
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = NTL ? __builtin_nontemporal_load(p + j * 256) : p[j * 256];
#pragma unroll
        for (int j = 0; j < U; ++j) { if (NTS) __builtin_nontemporal_store(r[j], q + j * 256); else q[j * 256] = r[j]; }
    }
}

static float best = 0.f;
static char best_what[160];

template <int U, bool NTL, bool NTS, bool CHUNK>
void run(const v4f* a, v4f* b, size_t n, int wg_per_cu, hipEvent_t e0, hipEvent_t e1) {
    const int grid = 256 * wg_per_cu;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copyk<U, NTL, NTS, CHUNK>), dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e0, 0));
    const int passes = 12;
    for (int i = 0; i < passes; ++i) hipLaunchKernelGGL((copyk<U, NTL, NTS, CHUNK>), dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const float tbs = 2.0f * n * 16 * passes / (ms * 1e-3f) / 1e12f;
    printf(" %5.2f", tbs);
    if (tbs > best) {
        best = tbs;
        snprintf(best_what, sizeof best_what, "%d KiB per wave burst, load %s, store %s, %s, %d workgroups of 256 per CU", U,
                 NTL ? "nontemporal" : "plain", NTS ? "nontemporal" : "plain", CHUNK ? "own regions" : "grid-stride", wg_per_cu);
    }
}

template <int U, bool NTL, bool NTS, bool CHUNK>
void row(const v4f* a, v4f* b, size_t n, hipEvent_t e0, hipEvent_t e1) {
    printf("U=%2d load %-5s store %-5s %-11s:", U, NTL ? "nt" : "plain", NTS ? "nt" : "plain", CHUNK ? "own regions" : "grid-stride");
    for (int w : {1, 2, 3, 4, 6, 8, 12, 16}) run<U, NTL, NTS, CHUNK>(a, b, n, w, e0, e1);
    printf("\n");
}

template <int U>
void block(const v4f* a, v4f* b, size_t n, hipEvent_t e0, hipEvent_t e1) {
    row<U, false, false, false>(a, b, n, e0, e1);
    row<U, true, false, false>(a, b, n, e0, e1);
    row<U, false, true, false>(a, b, n, e0, e1);
    row<U, true, true, false>(a, b, n, e0, e1);
    row<U, false, false, true>(a, b, n, e0, e1);
    row<U, true, false, true>(a, b, n, e0, e1);
    row<U, false, true, true>(a, b, n, e0, e1);
    row<U, true, true, true>(a, b, n, e0, e1);
}

int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    v4f *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# float4 copy, 1 GiB -> 1 GiB, TB/s of bytes read + written; columns: 1 2 3 4 6 8 12 16 workgroups of 256 threads per CU\n");
    block<1>(a, b, n, e0, e1);
    block<4>(a, b, n, e0, e1);
    block<8>(a, b, n, e0, e1);
    block<16>(a, b, n, e0, e1);
    printf("best: %.2f TB/s (%s) = %.3f of 8 TB/s; MI355X_MICROARCH.md:36 quotes 6.29\n", best, best_what, best / 8.0f);
    // The same copy against its FOOTPRINT: the best shape (U = 4, non-temporal both ways, grid-stride, 12 workgroups per CU) and
    // the plain own-regions shape over buffers of 128 MiB .. 8 GiB each (the engine's cfg3 batch touches ~11 GB per step).
    CK(hipFree(a)); CK(hipFree(b));
    printf("# footprint sweep: MiB per buffer | best shape TB/s | plain own-regions TB/s | best shape, source and destination halves of ONE allocation\n");
    for (size_t mib : {128, 256, 512, 1024, 2048, 4096, 8192}) {
        const size_t by = mib << 20, nn = by / 16;
        v4f *x, *y, *z;
        if (hipMalloc(&x, by) != hipSuccess || hipMalloc(&y, by) != hipSuccess || hipMalloc(&z, 2 * by) != hipSuccess) { printf("%5zu: allocation failed\n", mib); break; }
        CK(hipMemset(x, 1, by)); CK(hipMemset(y, 0, by)); CK(hipMemset(z, 1, 2 * by));
        printf("%5zu |", mib);
        run<4, true, true, false>(x, y, nn, 12, e0, e1);
        printf(" |");
        run<4, false, false, true>(x, y, nn, 8, e0, e1);
        printf(" |");
        run<4, true, true, false>(z, z + nn, nn, 12, e0, e1);
        printf("\n");
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(z));
    }
    // Is the fall with the footprint the Infinity Cache no longer holding what the previous pass left?  The SAME 1 GiB per pass,
    // best shape, (a) the same slice every pass, (b) a different 1 GiB slice of 8 GiB buffers every pass (each slice comes round
    // again after 16 GiB of other traffic: touched once, as far as any cache can tell).
    {
        const size_t slice = (size_t)1 << 30, nn = slice / 16;
        v4f *x, *y;
        if (hipMalloc(&x, 8 * slice) == hipSuccess && hipMalloc(&y, 8 * slice) == hipSuccess) {
            CK(hipMemset(x, 1, 8 * slice)); CK(hipMemset(y, 0, 8 * slice));
            for (int rolling = 0; rolling < 2; ++rolling) {
                for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((copyk<4, true, true, false>), dim3(256 * 12), dim3(256), 0, 0, x + (rolling ? i : 0) * nn, y + (rolling ? i : 0) * nn, nn);
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 16; ++i) hipLaunchKernelGGL((copyk<4, true, true, false>), dim3(256 * 12), dim3(256), 0, 0, x + (rolling ? i % 8 : 0) * nn, y + (rolling ? i % 8 : 0) * nn, nn);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("1 GiB per pass, best shape, %s: %.2f TB/s\n", rolling ? "a different slice of 8 GiB every pass (touched once)" : "the same slice every pass", 2.0 * slice * 16 / (ms * 1e-3) / 1e12);
            }
            CK(hipFree(x)); CK(hipFree(y));
        }
    }
    return 0;
}
