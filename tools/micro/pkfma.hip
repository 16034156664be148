// Microbenchmark: issue rate of v_pk_fma_f32 chains shaped like K2's MAC (VGPR pair x VGPR pair + VGPR pair,
// op_sel modifiers), with NACC accumulators, at 1..4 waves per SIMD.  hipcc --offload-arch=gfx950 pkfma.hip -o pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NACC, int NW>
__global__ __launch_bounds__(256) void k(v2f* out, const v2f* in, int iters) {
    v2f w[NW], g[NW], acc[NACC];
    for (int i = 0; i < NW; ++i) { w[i] = in[threadIdx.x + 256 * i]; g[i] = in[threadIdx.x + 256 * (i + NW)]; }
    for (int a = 0; a < NACC; ++a) acc[a] = v2f{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[(2 * j) % NACC]) : "v"(w[j]), "v"(g[j]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc[(2 * j + 1) % NACC]) : "v"(w[j]), "v"(g[j]));
        }
    }
    v2f s = acc[0];
    for (int a = 1; a < NACC; ++a) s += acc[a];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NW>
void run(int wg_per_cu, const char* tag) {
    v2f *in, *out;
    hipMalloc(&in, 256 * 2 * NW * sizeof(v2f)); hipMemset(in, 0, 256 * 2 * NW * sizeof(v2f));
    const int blocks = 256 * wg_per_cu;      // 256-thread blocks: wg_per_cu waves per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(v2f));
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC, NW><<<blocks, 256>>>(out, in, 10);
    hipEventRecord(a);
    k<NACC, NW><<<blocks, 256>>>(out, in, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double insts = (double)iters * NW * 2;                  // per wave
    const double cyc = ms * 1e-3 * 2.1e9;                         // nominal 2.1 GHz
    printf("%s NACC=%d NW=%d waves/SIMD=%d: %.3f ms, %.2f cycles per pk_fma per wave, %.2f per SIMD\n", tag, NACC, NW, wg_per_cu, ms,
           cyc / insts, cyc / insts / wg_per_cu);
    hipFree(in); hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) run<2, 33>(w, "acc2");
    for (int w = 1; w <= 4; ++w) run<6, 33>(w, "acc6");
    for (int w = 1; w <= 4; ++w) run<12, 33>(w, "acc12");
    return 0;
}
