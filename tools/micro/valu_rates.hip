// Microbenchmark (round 5): what K2's walk loop is made of, priced in SHADER-CLOCK cycles per wave (s_memtime, so
// no assumption about the clock): v_pk_fma_f32 with op_sel, v_fma_f32, their mixes, and scalar instructions beside
// them, at 1..4 waves per SIMD.  One "iteration" has the instruction mix of one step of the walk at 33 rows.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int... I, class F> __device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void pk_re(v2f& a, const v2f& x, const v2f& h) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(x), "v"(h)); }
__device__ __forceinline__ void pk_im(v2f& a, const v2f& x, const v2f& h) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a) : "v"(x), "v"(h)); }
__device__ __forceinline__ void pk_plain(v2f& a, const v2f& x, const v2f& h) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(h)); }
__device__ __forceinline__ void fma1(float& a, const float& x, const float& h) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(h)); }
__device__ __forceinline__ void vadd(float& a, const float& x) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(x)); }
__device__ __forceinline__ void salu(int& s) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc"); }
// a compare and a conditional branch that is never taken (its target is the next instruction anyway)
__device__ __forceinline__ void branch_nt(int& s) { asm volatile("s_cmp_eq_u32 %0, -1\n\ts_cbranch_scc1 1f\n1:" : "+s"(s) : : "scc"); }

// MODE 0: 66 pk (today's walk: re + im per row)      1: 66 v_fma_f32        2: 33 pk(plain) + 33 v_fma_f32
//      3: 33 pk(plain) + 17 pk(op_sel)  (three FMAs per MAC, T packed)      4: 132 v_fma_f32 (today's flops unpacked)
//      5: 99 v_fma_f32 (three FMAs per MAC, nothing packed)                 6: 33 pk only
// SAL: scalar instructions per iteration, spread evenly
// VAL: other plain vector instructions per iteration (v_add_f32), spread evenly
template <int MODE, int SAL, int VAL = 0, int BR = 0>
__global__ __launch_bounds__(256) void k(float* out, const v2f* in, int iters, long long* cyc) {
    extern __shared__ float lds_pad[];      // only there to limit the workgroups per CU to exactly w
    v2f w[34], g[34], acc[4];
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 34; ++i) { w[i] = in[threadIdx.x + 256 * i]; g[i] = in[threadIdx.x + 256 * (i + 34)]; }
    for (int a = 0; a < 4; ++a) acc[a] = v2f{0.f, 0.f};
    int sc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        sfor<33>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (MODE == 0) { pk_re(acc[(2 * j) % 4], w[j], g[j]); pk_im(acc[(2 * j + 1) % 4], w[j], g[j]); }
            if constexpr (MODE == 1) { fma1(t[(2 * j) % 4], w[j].x, g[j].x); fma1(t[(2 * j + 1) % 4], w[j].y, g[j].y); }
            if constexpr (MODE == 2) { pk_plain(acc[j % 4], w[j], g[j]); fma1(t[j % 4], w[j].x, g[j].y); }
            if constexpr (MODE == 3) { pk_plain(acc[j % 2], w[j], g[j]); if constexpr (j % 2 == 0) pk_re(acc[2 + (j / 2) % 2], w[j + 1], g[j + 1]); }
            if constexpr (MODE == 4) { fma1(t[0], w[j].x, g[j].x); fma1(t[1], w[j].y, g[j].y); fma1(t[2], w[j].x, g[j].y); fma1(t[3], w[j].y, g[j].x); }
            if constexpr (MODE == 5) { fma1(t[0], w[j].x, g[j].x); fma1(t[1], w[j].y, g[j].y); fma1(t[2], w[j].x, g[j].y); }
            if constexpr (MODE == 6) { pk_plain(acc[j % 4], w[j], g[j]); }
            if constexpr (VAL > 0) { if constexpr ((j * VAL) / 33 != ((j + 1) * VAL) / 33) { sfor<((j + 1) * VAL) / 33 - (j * VAL) / 33>([&](auto ic) { vadd(t[decltype(ic)::value % 4], w[33].x); }); } }
            if constexpr (SAL > 0) { if constexpr ((j * SAL) / 33 != ((j + 1) * SAL) / 33) { sfor<((j + 1) * SAL) / 33 - (j * SAL) / 33>([&](auto) { salu(sc); }); } }
            if constexpr (BR > 0) { if constexpr ((j * BR) / 33 != ((j + 1) * BR) / 33) { sfor<((j + 1) * BR) / 33 - (j * BR) / 33>([&](auto) { branch_nt(sc); }); } }
        });
    }
    const long long t1 = __builtin_readcyclecounter();
    v2f s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + t[0] + t[1] + t[2] + t[3] + (float)sc;
    if (iters < 0) lds_pad[threadIdx.x] = s.x;
    if (threadIdx.x % 64 == 0) atomicAdd((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int MODE, int SAL, int VAL = 0, int BR = 0>
void run(const char* what, int ninst) {
    v2f* in; float* out; long long* cyc;
    (void)hipMalloc(&in, 256 * 68 * sizeof(v2f)); (void)hipMemset(in, 0, 256 * 68 * sizeof(v2f));
    (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    (void)hipMalloc(&cyc, 8);
    const int iters = 4000;
    printf("%-58s", what); fflush(stdout);
    (void)hipFuncSetAttribute((const void*)k<MODE, SAL, VAL, BR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 1; w <= 3; ++w) {
        const int blocks = 256 * w;                 // 256-thread blocks, one wave per SIMD each: w waves per SIMD
        const size_t lds = (size_t)(150 * 1024 / w);   // ... and at most w of them fit a CU's 160 KB: every CU gets exactly w
        k<MODE, SAL, VAL, BR><<<blocks, 256, lds>>>(out, in, 10, cyc);
        (void)hipMemset(cyc, 0, 8);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        k<MODE, SAL, VAL, BR><<<blocks, 256, lds>>>(out, in, iters, cyc);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave_iter = (double)c / (blocks * 4) / iters;     // shader cycles per iteration, seen by one wave
        printf(" | %dw: %6.1f cyc/it (%.2f/inst/SIMD) %.3f ms", w, per_wave_iter, per_wave_iter / ninst / w, ms); fflush(stdout);
    }
    printf("\n");
    (void)hipFree(in); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per iteration as ONE wave sees them (s_memtime), at w waves per SIMD; /inst/SIMD = per instruction and SIMD\n");
    run<0, 0>("0: 66 pk_fma op_sel (today: 4 FMA / MAC)", 66);
    run<6, 0>("6: 33 pk_fma plain", 33);
    run<1, 0>("1: 66 v_fma_f32", 66);
    run<4, 0>("4: 132 v_fma_f32 (4 FMA / MAC unpacked)", 132);
    run<5, 0>("5: 99 v_fma_f32 (3 FMA / MAC unpacked)", 99);
    run<2, 0>("2: 33 pk + 33 v_fma_f32 (3 FMA / MAC, T scalar)", 66);
    run<3, 0>("3: 33 pk + 17 pk op_sel (3 FMA / MAC, T packed)", 50);
    run<0, 16>("0 + 16 SALU", 82);
    run<0, 32>("0 + 32 SALU", 98);
    run<3, 16>("3 + 16 SALU", 66);
    run<3, 32>("3 + 32 SALU", 82);
    run<2, 16>("2 + 16 SALU", 82);
    run<0, 15, 13>("today's step: 66 pk + 13 VALU + 15 SALU", 94);
    run<3, 15, 14>("3-FMA packed T: 50 pk + 14 VALU + 15 SALU", 79);
    run<2, 15, 14>("3-FMA scalar T: 33 pk + 33 fma + 14 VALU + 15 SALU", 95);
    run<3, 8, 10>("3-FMA packed T, lean: 50 pk + 10 VALU + 8 SALU", 68);
    run<0, 8, 10>("4-FMA, lean: 66 pk + 10 VALU + 8 SALU", 84);
    // what a never-taken branch costs beside scalar arithmetic (a compare + s_cbranch_scc1 each, counted as two instructions)
    run<3, 8, 10, 1>("lean 3-FMA + 1 untaken branch (cmp + cbranch)", 70);
    run<3, 8, 10, 2>("lean 3-FMA + 2 untaken branches", 72);
    run<3, 8, 10, 4>("lean 3-FMA + 4 untaken branches", 76);
    run<3, 10, 10>("lean 3-FMA + 2 SALU", 70);
    run<3, 16, 10>("lean 3-FMA + 8 SALU", 76);
    return 0;
}
