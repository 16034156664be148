// Microbenchmark: the memory pattern of K2's whole-call walk without its arithmetic.
// Each thread owns one 8-byte bin and walks T rows: one 8-byte load (D ahead) and one 8-byte store per step.
// Row-major rows (stride 64 KiB between the rows a thread reads: the engine's layout) against tile-major
// (a workgroup's 256 bins of consecutive rows are contiguous: stride 2 KiB).  3 workgroups of 256 per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define GL __attribute__((address_space(1)))

template <int D>
__global__ __launch_bounds__(256) void walk(const v2f* X, v2f* Y, int T, int rows_x, long xstride, long ystride, int tile_major, int spin) {
    extern __shared__ char lds[];           // occupancy limiter only
    const int tile = blockIdx.x, unit = blockIdx.y;          // 32 bin tiles, S*C units
    const long P = 8192;
    const v2f* x; v2f* y;
    // tile_major: bit 0 = X, bit 1 = Y
    if (tile_major & 1) x = X + ((long)unit * 32 + tile) * rows_x * 256 + threadIdx.x;
    else x = X + (long)unit * rows_x * P + tile * 256 + threadIdx.x;
    if (tile_major & 2) y = Y + ((long)unit * 32 + tile) * T * 256 + threadIdx.x;
    else y = Y + (long)unit * T * ystride + tile * 256 + threadIdx.x;      // (ystride = P, or P + padding)
    v2f w[D];
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = *(const GL v2f*)(x + d * xstride);
    v2f acc{0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int t = t0 + u;
            v2f v = w[u];
            const int tn = (t + D < rows_x) ? t + D : rows_x - 1;
            w[u] = *(const GL v2f*)(x + tn * xstride);
            for (int k = 0; k < spin; ++k) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(v));
            acc += v;
            *(GL v2f*)(y + t * ystride) = acc;
        }
    }
}

// the same pattern with 16 bytes per lane: 128 threads cover a 256-bin tile
typedef float v4f __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(128) void walk16(const v4f* X, v4f* Y, int T, int rows_x, long xstride, long ystride) {
    extern __shared__ char lds[];
    const int tile = blockIdx.x, unit = blockIdx.y;
    const long P2 = 4096;
    const v4f* x = X + (long)unit * rows_x * P2 + tile * 128 + threadIdx.x;
    v4f* y = Y + (long)unit * T * P2 + tile * 128 + threadIdx.x;
    v4f w[D];
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = *(const GL v4f*)(x + d * xstride);
    v4f acc{0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int t = t0 + u;
            v4f v = w[u];
            const int tn = (t + D < rows_x) ? t + D : rows_x - 1;
            w[u] = *(const GL v4f*)(x + tn * xstride);
            acc += v;
            *(GL v4f*)(y + t * ystride) = acc;
        }
    }
}

// reference: plain 16-byte grid-stride copy of the same number of bytes
__global__ __launch_bounds__(256) void copy16(const v4f* a, v4f* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main(int argc, char** argv) {
    // 128 (stream, channel) units; T outputs, RX-row rings (default: 64 outputs, 96 rows; "256 288" = the 256-block call,
    // whose 4.6 GB per launch no cache holds from one launch to the next)
    const int S = 128, T = argc > 1 ? atoi(argv[1]) : 64, RX = argc > 2 ? atoi(argv[2]) : 96;
    const long P = 8192;
    v2f *X, *Y;
    hipMalloc(&X, (size_t)S * RX * P * 8); hipMalloc(&Y, (size_t)S * T * (P + 1024) * 8);
    hipMemset(X, 0, (size_t)S * RX * P * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    {
        const size_t n = (size_t)S * T * P * 8 / 16;
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(copy16, dim3(256 * 8), dim3(256), 0, 0, (const v4f*)X, (v4f*)Y, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("copy16 of %.0f MB: %.3f ms, %.2f TB/s (read + write)\n", n * 16 / 1e6, best, 2.0 * n * 16 / best / 1e9);
    }
    for (int spin : {0, 16}) {
        for (int tm = 0; tm < 4; ++tm) {
            const long xs = (tm & 1) ? 256 : P, ys = (tm & 2) ? 256 : P;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(walk<8>, dim3(32, S), dim3(256), 50 * 1024, 0, X, Y, T, RX, xs, ys, tm, spin);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double bytes = (double)S * P * 8 * (T + 8) + (double)S * P * 8 * T;
            printf("spin=%2d %s: %.3f ms, %.2f TB/s\n", spin, tm == 0 ? "X rows, Y rows (64 KiB stride)" : tm == 1 ? "X tiles (2 KiB stride), Y rows" : tm == 2 ? "X rows, Y tiles" : "X tiles, Y tiles", best, bytes / best / 1e9);
        }
    }
    // Y rows with a padded stride: are 64 KiB-apart 2 KiB pieces an unlucky pattern for the HBM channels?
    for (int pad : {0, 32, 64, 256, 512, 1024}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(walk<8>, dim3(32, S), dim3(256), 50 * 1024, 0, X, Y, T, RX, (long)P, (long)(P + pad), 0, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double bytes = (double)S * P * 8 * (T + 8) + (double)S * P * 8 * T;
        printf("Y row stride P + %4d elements: %.3f ms, %.2f TB/s\n", pad, best, bytes / best / 1e9);
    }
    for (int wgs : {3, 6}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(walk16<8>, dim3(32, S), dim3(128), (150 / wgs) * 1024, 0, (const v4f*)X, (v4f*)Y, T, RX, 4096L, 4096L);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const double bytes = (double)S * P * 8 * (T + 8) + (double)S * P * 8 * T;
        printf("16 B per lane, %d workgroups of 128 per CU: %.3f ms, %.2f TB/s\n", wgs, best, bytes / best / 1e9);
    }
    return 0;
}
