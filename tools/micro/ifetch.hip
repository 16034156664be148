// Microbenchmark: does a LARGE straight-line body of 8-byte VALU instructions issue as fast as a small loop?
// K2's walk is ~28 KB of straight-line v_pk_fma_f32 per round; this runs the same instruction with bodies of
// 0.5 KB .. 42 KB at 3 waves per SIMD (12 per CU, two CUs share an instruction cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#include <utility>
template <int... I, class F> __device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void fma_a(v2f& a, const v2f& x, const v2f& h) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(x), "v"(h)); }
__device__ __forceinline__ void fma_b(v2f& a, const v2f& x, const v2f& h) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a) : "v"(x), "v"(h)); }

template <int REP>   // body = REP * 66 instructions of 8 bytes
__global__ __launch_bounds__(256) void k(v2f* out, const v2f* in, int iters) {
    v2f w[33], g[33], acc[6];
    for (int i = 0; i < 33; ++i) { w[i] = in[threadIdx.x + 256 * i]; g[i] = in[threadIdx.x + 256 * (i + 33)]; }
    for (int a = 0; a < 6; ++a) acc[a] = v2f{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        sfor<REP>([&](auto rc) {                 // fold expressions: the loop unroller gives up beyond ~16 KB
            constexpr int r = decltype(rc)::value;
            sfor<33>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                fma_a(acc[(2 * j) % 6], w[(j + r) % 33], g[j]);
                fma_b(acc[(2 * j + 1) % 6], w[(j + r) % 33], g[j]);
            });
        });
    }
    v2f s = acc[0];
    for (int a = 1; a < 6; ++a) s += acc[a];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int REP>
void run() {
    v2f *in, *out;
    (void)hipMalloc(&in, 256 * 66 * sizeof(v2f)); (void)hipMemset(in, 0, 256 * 66 * sizeof(v2f));
    const int blocks = 256 * 3;
    (void)hipMalloc(&out, blocks * 256 * sizeof(v2f));
    const int iters = 8000 / REP;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<REP><<<blocks, 256>>>(out, in, 4);
    (void)hipEventRecord(a);
    k<REP><<<blocks, 256>>>(out, in, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double insts = (double)iters * REP * 66;
    printf("body %5.1f KB: %.3f ms, %.2f cycles per pk_fma per SIMD (3 waves, nominal 2.1 GHz)\n", REP * 66 * 8 / 1024.0, ms,
           ms * 1e-3 * 2.1e9 / insts / 3);
    (void)hipFree(in); (void)hipFree(out);
}

int main() {
    run<1>(); run<8>(); run<16>(); run<20>(); run<24>(); run<28>(); run<32>(); run<36>(); run<40>(); run<64>(); run<124>();
    return 0;
}
