// fft_core.hpp — workgroup-resident complex FFT for gfx950 (CDNA4).
//
// One workgroup transforms N = P complex points held in LDS (P = folve's
// `fragm`, 64..8192: zita-fconfig.cc:74-77 in the reference), which is the
// half-size complex transform behind the 2P-point real FFT of one partition.
// Stockham autosort passes (no bit reversal), radix 16/8/4/2 butterflies in
// registers, 64-wide wavefronts, in-place in a padded LDS image so that the
// strided writes of the early passes are bank-conflict free.
//
// The butterflies are plain C++ templates (usable from host code too, which is
// how tests/host_fft_check.cpp verifies them without a GPU).
#pragma once

#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FK_HD __host__ __device__ __forceinline__
#define FK_D __device__ __forceinline__
#else
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
#define FK_HD inline
#endif

namespace fk {

// ---- complex arithmetic ---------------------------------------------------------
// Device code: a complex number is a VGPR pair and every operation below is ONE or TWO packed
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), the swaps, broadcasts and sign
// flips of complex arithmetic expressed by op_sel / op_sel_hi / neg_lo / neg_hi instead of
// v_mov / v_xor.  hipcc finds only part of this from scalar source (a radix-16 butterfly:
// 138 VALU instructions from the scalar form, 88 from this one).  Host code (the CPU check of
// the butterflies, tests/host_fft_check.cpp) compiles the scalar forms.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float fk_v2f __attribute__((ext_vector_type(2)));
#define FK_V(a) (fk_v2f{(a).x, (a).y})
#define FK_F2(v) (float2{(v).x, (v).y})
FK_HD float2 cadd(float2 a, float2 b) { const fk_v2f r = FK_V(a) + FK_V(b); return FK_F2(r); }
FK_HD float2 csub(float2 a, float2 b) { const fk_v2f r = FK_V(a) - FK_V(b); return FK_F2(r); }
// a + i*b, a - i*b
FK_HD float2 cadd_i(float2 a, float2 b) {
    fk_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
FK_HD float2 csub_i(float2 a, float2 b) {
    fk_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
// a + conj(b), a - conj(b)
FK_HD float2 cadd_conj(float2 a, float2 b) {
    fk_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
FK_HD float2 csub_conj(float2 a, float2 b) {
    fk_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
// conj(a) + i*conj(b) = conj(a - i*b)
FK_HD float2 conj_csub_i(float2 a, float2 b) {
    fk_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
// a * b
// Both instructions in ONE asm statement: between two statements where the second reads what the
// first wrote hipcc's hazard recognizer inserts an s_nop (it cannot see that they are plain VALU).
FK_HD float2 cmul(float2 a, float2 b) {
    fk_v2f t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=v"(r), "=&v"(t) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
// a * conj(b)
FK_HD float2 cmulc(float2 a, float2 b) {
    fk_v2f t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r), "=&v"(t) : "v"(FK_V(a)), "v"(FK_V(b)));
    return FK_F2(r);
}
// a * (wr + i*wi), wr and wi known at compile time
// The constant pair is an SGPR operand ("s"): as a VGPR operand every distinct constant of the
// butterflies would occupy a loop-invariant VGPR pair (about 30 VGPRs in a walker kernel).
FK_HD float2 cmul_const(float2 a, float wr, float wi) {
    const fk_v2f t = FK_V(a) * wr;
    const fk_v2f ws = fk_v2f{wi, wi};
    fk_v2f r;                                             // r.lo = t.lo - a.hi*wi,  r.hi = t.hi + a.lo*wi
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r) : "v"(FK_V(a)), "s"(ws), "v"(t));
    return FK_F2(r);
}
#else
FK_HD float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
FK_HD float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
FK_HD float2 cadd_i(float2 a, float2 b) { return float2{a.x - b.y, a.y + b.x}; }
FK_HD float2 csub_i(float2 a, float2 b) { return float2{a.x + b.y, a.y - b.x}; }
FK_HD float2 cadd_conj(float2 a, float2 b) { return float2{a.x + b.x, a.y - b.y}; }
FK_HD float2 csub_conj(float2 a, float2 b) { return float2{a.x - b.x, a.y + b.y}; }
FK_HD float2 conj_csub_i(float2 a, float2 b) { return float2{a.x + b.y, -a.y + b.x}; }
FK_HD float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
FK_HD float2 cmulc(float2 a, float2 b) { return float2{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }  // a * conj(b)
FK_HD float2 cmul_const(float2 a, float wr, float wi) { return float2{a.x * wr - a.y * wi, a.x * wi + a.y * wr}; }
#endif

// 16th roots of unity: cos/sin(2*pi*i/16), i = 0..7.
constexpr float kCos16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
constexpr float kSin16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                             1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// 32nd roots of unity: cos/sin(2*pi*i/32), i = 0..7.
constexpr float kCos32[8] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                             0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
constexpr float kSin32[8] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f,
                             0.70710678118654752f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f};

template <int R, bool INV>
FK_HD void dft(float2 (&v)[R]);

// v[K], v[K + R/2] = e[K] +/- o[K] * exp(-/+ 2*pi*i*K/R)  (INV: +); the quarter turn is part of the add
template <int R, bool INV, int... K>
FK_HD void dft_combine(float2 (&v)[R], const float2 (&e)[R / 2], const float2 (&o)[R / 2],
                       std::integer_sequence<int, K...>) {
    ((void)([&] {
         constexpr int IDX = K * 16 / R;
         if constexpr (IDX == 0) {
             v[K] = cadd(e[K], o[K]);
             v[K + R / 2] = csub(e[K], o[K]);
         } else if constexpr (IDX == 4) {
             v[K] = INV ? cadd_i(e[K], o[K]) : csub_i(e[K], o[K]);
             v[K + R / 2] = INV ? csub_i(e[K], o[K]) : cadd_i(e[K], o[K]);
         } else {
             const float2 t = cmul_const(o[K], kCos16[IDX], INV ? kSin16[IDX] : -kSin16[IDX]);
             v[K] = cadd(e[K], t);
             v[K + R / 2] = csub(e[K], t);
         }
     }()),
     ...);
}

// R-point DFT, natural order in and out (decimation in time, fully unrolled).
template <int R, bool INV>
FK_HD void dft(float2 (&v)[R]) {
    static_assert(R == 2 || R == 4 || R == 8 || R == 16, "radix");
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else {
        float2 e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = v[2 * k];
            o[k] = v[2 * k + 1];
        }
        dft<R / 2, INV>(e);
        dft<R / 2, INV>(o);
        dft_combine<R, INV>(v, e, o, std::make_integer_sequence<int, R / 2>{});
    }
}

// ---- radix plan -----------------------------------------------------------
// log2(N) -> up to four radices, product N.  13 (P = 8192) -> 16,16,8,4.
struct Plan {
    int r[4];       // radix of each pass
    int ns[4];      // product of the earlier radices (twiddle period of the pass)
    int off[4];     // offset of the pass's twiddle table in the per-P pass-twiddle buffer
    int n;          // passes
    int total;      // entries in the pass-twiddle buffer
};
constexpr Plan make_plan(int log2n) {
    Plan p{{1, 1, 1, 1}, {1, 1, 1, 1}, {0, 0, 0, 0}, 0, 0};
    int left = log2n;
    while (left >= 4 && left != 5) { p.r[p.n++] = 16; left -= 4; }
    while (left >= 3) { p.r[p.n++] = 8; left -= 3; }
    if (left == 2) { p.r[p.n++] = 4; left = 0; }
    if (left == 1) { p.r[p.n++] = 2; left = 0; }
    int ns = 1, off = 0;
    for (int i = 0; i < p.n; ++i) {
        p.ns[i] = ns;
        p.off[i] = off;
        if (ns > 1) off += (p.r[i] - 1) * ns;     // table[(r-1)*ns + k] = exp(-2*pi*i*k*r/(ns*R)), r = 1..R-1
        ns *= p.r[i];
    }
    p.total = off;
    return p;
}
// Passes whose twiddle period is short read every row of their table (it stays in
// L1); longer periods read only the power-of-two rows, coalesced, and form the
// other powers as products (at most three factors).
constexpr int kTableMaxPeriod = 64;

// Threads per workgroup for an N-point transform: 16 points per thread, at
// least one wavefront.
constexpr int threads_for(int n) { return n / 16 < 64 ? 64 : n / 16; }

// LDS image: one float2 of padding after every 16, so element i lives at
// i + i/16.  Pass-1 writes (stride = radix) and every later access pattern are
// then conflict free for ds_write_b64 / ds_read_b64 lane groups.
constexpr int lds_elems(int n) { return n + n / 16; }
FK_HD int phys(int i) { return i + (i >> 4); }

#if defined(__HIPCC__)

// ---- global-memory accessors -------------------------------------------------
// A pointer that was itself read from memory (the StreamJob descriptors: PCM, FDL rows,
// tails) is a GENERIC pointer to the compiler, and plain dereferences become flat_load /
// flat_store: those also tick lgkmcnt and return out of order, so every LDS wait in the FFT
// kernels would wait for them too.  These helpers cast to the global address space first
// (global_load / global_store, vmcnt only).
#define FK_GLOBAL __attribute__((address_space(1)))
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
FK_D float gld(const float* p) { return *(const FK_GLOBAL float*)p; }
FK_D float2 gld(const float2* p) { const v2f v = *(const FK_GLOBAL v2f*)p; return float2{v.x, v.y}; }
FK_D float4 gld(const float4* p) { const v4f v = *(const FK_GLOBAL v4f*)p; return float4{v.x, v.y, v.z, v.w}; }
FK_D v2f gld_v2(const float2* p) { return *(const FK_GLOBAL v2f*)p; }
FK_D void gst(float* p, float v) { *(FK_GLOBAL float*)p = v; }
FK_D void gst(float2* p, float2 v) { *(FK_GLOBAL v2f*)p = v2f{v.x, v.y}; }
FK_D void gst(float4* p, float4 v) { *(FK_GLOBAL v4f*)p = v4f{v.x, v.y, v.z, v.w}; }
FK_D void gst_v2(float2* p, v2f v) { *(FK_GLOBAL v2f*)p = v; }

// Uniform base + 32-bit per-lane byte offset: the compiler emits the SGPR-base form
// (global_load/store v, voffset, s[base:base+1]) — one VGPR per address instead of a 64-bit pair
// and no per-lane 64-bit address arithmetic.  The base must be wave-uniform for that.
FK_D float2 gld_u2(const void* base, unsigned off) {
    const v2f v = *(const FK_GLOBAL v2f*)((const FK_GLOBAL char*)base + off);
    return float2{v.x, v.y};
}
FK_D void gst_u2(void* base, unsigned off, float2 v) { *(FK_GLOBAL v2f*)((FK_GLOBAL char*)base + off) = v2f{v.x, v.y}; }
// Loads and the 16-byte store carry the non-temporal hint (`nt`), for data this launch touches once and no later launch
// finds in a cache anyway (the walkers' PCM, K3's read of the Y scratch): measured -3 % on K1, -3 % on
// K3 at 64 x 256 blocks.  The hint on K1's spectrum stores or on K2's loads and stores measured slower.
FK_D float2 gld_u2_once(const void* base, unsigned off) {
    const v2f v = __builtin_nontemporal_load((const FK_GLOBAL v2f*)((const FK_GLOBAL char*)base + off));
    return float2{v.x, v.y};
}
FK_D float4 gld_u4_once(const void* base, unsigned off) {
    const v4f v = __builtin_nontemporal_load((const FK_GLOBAL v4f*)((const FK_GLOBAL char*)base + off));
    return float4{v.x, v.y, v.z, v.w};
}
FK_D void gst_u2_once(void* base, unsigned off, float2 v) {
    __builtin_nontemporal_store(v2f{v.x, v.y}, (FK_GLOBAL v2f*)((FK_GLOBAL char*)base + off));
}
FK_D void gst_u4_once(void* base, unsigned off, float4 v) {
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, (FK_GLOBAL v4f*)((FK_GLOBAL char*)base + off));
}

// ---- synchronisation policies ------------------------------------------------
struct WorkgroupSync {
    static FK_D void sync() { __syncthreads(); }
};
// The lanes of ONE wavefront exchanging data through LDS: DS operations of a
// wave execute in order, so all that is needed is that the compiler keeps the
// program order and every lane's data have returned.
struct WaveSync {
    static FK_D void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// Twiddles w[r] = W^r, r = 1..R-1, from the power-of-two rows of a pass table
// (table[(r-1)*NS + k]): rows 1,2,4,8 are read (coalesced), the other powers are
// products of at most three factors.  FULL: read every row (short periods, L1).
template <int R, int NS, bool FULL>
FK_D void load_twiddles(float2 (&w)[R], const float2* __restrict__ table, int k) {
    if constexpr (FULL) {
#pragma unroll
        for (int r = 1; r < R; ++r) w[r] = table[(r - 1) * NS + k];
    } else {
#pragma unroll
        for (int r = 1; r < R; r *= 2) w[r] = table[(r - 1) * NS + k];
#pragma unroll
        for (int r = 3; r < R; ++r) {
            if ((r & (r - 1)) != 0) {
                int top = 1;
                while (top * 2 <= r) top *= 2;
                w[r] = cmul(w[top], w[r - top]);
            }
        }
    }
}

// One in-place Stockham pass over an N-point sequence held in a padded LDS image
// (element i at s[phys(i)]), executed by NT threads.  Sync separates the read and
// write phases (the caller synchronises after the pass).
//   tw: this pass's table, tw[(r-1)*NS + k] = exp(-2*pi*i*k*r/(NS*R))
template <int N, int NT, int R, int NS, bool INV, class Sync>
FK_D void stockham_pass(float2* s, const float2* __restrict__ tw, int tid) {
    constexpr int NB = N / R;                       // butterflies in this pass
    constexpr int CNT = (NB + NT - 1) / NT;         // per thread
    constexpr bool GUARD = (NB % NT) != 0;
    float2 v[CNT][R];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int j = tid + c * NT;
        if (!GUARD || j < NB) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[c][r] = s[phys(j + r * NB)];
        }
    }
    Sync::sync();
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int j = tid + c * NT;
        if (!GUARD || j < NB) {
            const int k = j & (NS - 1);
            if constexpr (NS > 1) {
                float2 w[R];
                load_twiddles<R, NS, (NS <= kTableMaxPeriod)>(w, tw, k);
#pragma unroll
                for (int r = 1; r < R; ++r) v[c][r] = INV ? cmulc(v[c][r], w[r]) : cmul(v[c][r], w[r]);
            }
            dft<R, INV>(v[c]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) s[phys(j0 + r * NS)] = v[c][r];
        }
    }
}

// ---- cross-lane 4 x 4 transpose ----------------------------------------------------
// The four lanes b, b+16, b+32, b+48 of a wavefront (one lane of each 16-lane row) exchange the
// registers r0..r3: afterwards lane 16a + b holds in r_i what lane 16i + b held in r_a.  Two
// half exchanges of 32 lanes (v_permlane32_swap: lanes 32..63 of the first operand swap with
// lanes 0..31 of the second) and two of 16 (v_permlane16_swap: the odd rows of the first with
// the even rows of the second) — the register-file shuffle gfx950 adds for exactly this; no LDS
// access, no address arithmetic.  tests/test_forms_gpu.py (test_xlane_exchange_semantics) pins the semantics on the device.
FK_D void xlane_swap32(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
FK_D void xlane_swap16(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
FK_D void xlane_transpose4(float& r0, float& r1, float& r2, float& r3) {
    xlane_swap32(r0, r2);
    xlane_swap32(r1, r3);
    xlane_swap16(r0, r1);
    xlane_swap16(r2, r3);
}

// The last two passes of a wavefront's 1024-point row (radix 16 with period 16, radix 4 with
// period 256) as ONE pass: after the radix-16 butterflies lane 16a + b holds the elements
// 256a + b + 16r (r < 16) and the radix-4 butterflies of lane t want t + 64c + 256r' (c, r' < 4):
// the same 16 elements per lane group {b, b+16, b+32, b+48}, transposed between lane row a and
// register index r & 3.  So the exchange between the passes is the cross-lane transpose above
// instead of a round trip through LDS (16 ds_write_b64 + 16 ds_read_b64 per lane saved).
//   UPPER: write only the upper half of the row (the inverse transform's overlap-save output
//   keeps the last P of 2P samples = the upper half of every row; the rest is never read).
template <bool INV, bool UPPER, class Sync>
FK_D void fused_pass_16_4(float2* s, const float2* __restrict__ tw16, const float2* __restrict__ tw4, int tid) {
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = s[phys(tid + r * 64)];
    Sync::sync();
    {
        float2 w[16];
        load_twiddles<16, 16, true>(w, tw16, tid & 15);
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = INV ? cmulc(v[r], w[r]) : cmul(v[r], w[r]);
    }
    dft<16, INV>(v);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        xlane_transpose4(v[4 * c].x, v[4 * c + 1].x, v[4 * c + 2].x, v[4 * c + 3].x);
        xlane_transpose4(v[4 * c].y, v[4 * c + 1].y, v[4 * c + 2].y, v[4 * c + 3].y);
    }
    // radix-4 twiddles W_1024^(k*r) at k = tid + 64*c: the values at k = tid (two table reads, r = 1, 2)
    // rotated by W_16^(c*r) — compile-time constants — instead of eight table reads
    float2 wk[4];
    load_twiddles<4, 256, false>(wk, tw4, tid);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = tid + 64 * c;
        float2 u[4];
        u[0] = v[4 * c];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const int q = (c * r) & 15;                      // W_16^q = (cos, -sin)(2*pi*q/16)
            const float cq = (q < 8) ? kCos16[q & 7] : -kCos16[q & 7];
            const float sq = (q < 8) ? kSin16[q & 7] : -kSin16[q & 7];
            const float2 w = (q == 0) ? wk[r] : cmul_const(wk[r], cq, -sq);
            u[r] = INV ? cmulc(v[4 * c + r], w) : cmul(v[4 * c + r], w);
        }
        dft<4, INV>(u);
#pragma unroll
        for (int r = UPPER ? 2 : 0; r < 4; ++r) s[phys(k + 256 * r)] = u[r];
        }
}

// N-point FFT in place in one padded LDS image by NT threads: natural order in,
// natural order out.  Ends with the data written and synchronised.
//   XL: a single wavefront's 1024-point row runs its last two passes fused (fused_pass_16_4).
template <int LOG2N, int NT, bool INV, class Sync, bool XL = false, bool UPPER = false>
FK_D void lds_fft(float2* s, const float2* __restrict__ ptw, int tid) {
    constexpr int N = 1 << LOG2N;
    constexpr Plan pl = make_plan(LOG2N);
    static_assert(pl.n >= 2 && pl.n <= 4, "plan");
    constexpr int R0 = pl.r[0], R1 = pl.r[1], R2 = pl.r[2], R3 = pl.r[3];
    if constexpr (XL && LOG2N == 10 && NT == 64) {
        static_assert(R0 == 16 && R1 == 16 && R2 == 4 && pl.n == 3, "1024 = 16 * 16 * 4");
        stockham_pass<N, NT, 16, 1, INV, Sync>(s, ptw, tid);
        Sync::sync();
        fused_pass_16_4<INV, UPPER, Sync>(s, ptw + pl.off[1], ptw + pl.off[2], tid);
        Sync::sync();
    } else {
        stockham_pass<N, NT, R0, 1, INV, Sync>(s, ptw, tid);
        Sync::sync();
        stockham_pass<N, NT, R1, R0, INV, Sync>(s, ptw + pl.off[1], tid);
        Sync::sync();
        if constexpr (pl.n >= 3) {
            stockham_pass<N, NT, R2, R0 * R1, INV, Sync>(s, ptw + pl.off[2], tid);
            Sync::sync();
        }
        if constexpr (pl.n >= 4) {
            stockham_pass<N, NT, R3, R0 * R1 * R2, INV, Sync>(s, ptw + pl.off[3], tid);
            Sync::sync();
        }
    }
}

// ---- wave-autonomous P-point FFT ----------------------------------------------
// P = N1 * N2 with N1 = wavefronts in the workgroup and N2 = 1024 (16 points per
// lane); P <= 1024 is a single wavefront.  Stage A is one radix-N1 decimation
// step across the workgroup, done in registers straight from the source:
//     A[k1][n2] = W_P^(n2*k1) * sum_n1 z[n1*N2 + n2] * W_N1^(n1*k1)
// written to LDS row k1.  After ONE workgroup barrier wavefront k1 transforms its
// own row (N2 points, wave-level synchronisation only):
//     Z[k1 + N1*k2] = sum_n2 A[k1][n2] * W_N2^(n2*k2)
// so the waves of a CU drift apart and hide each other's latencies instead of
// meeting at eight barriers per transform.
template <int LOG2P>
struct WaveGeom {
    static constexpr int P = 1 << LOG2P;
    static constexpr int NT = threads_for(P);
    static constexpr int N1 = (P >= 1024) ? P / 1024 : 1;      // == waves per workgroup
    static constexpr int N2 = P / N1;
    static constexpr int LOG2N2 = LOG2P - (N1 == 16 ? 4 : N1 == 8 ? 3 : N1 == 4 ? 2 : N1 == 2 ? 1 : 0);
    // Row stride: the padded row plus 32/N1 so that a transposed read (lanes along
    // k = k1 + N1*k2) touches 32 distinct 8-byte banks.
    static constexpr int RS = lds_elems(N2) + (N1 > 1 ? 32 / N1 : 0);
    static constexpr int LDS_ELEMS = N1 * RS;
    static constexpr int COLS = (N2 + NT - 1) / NT;             // stage-A columns per thread
    static_assert(NT == 64 * N1 || N1 == 1, "one wavefront per row");
    static_assert(N1 <= 16, "radix of stage A");
    // element Z[k] after stage B
    static FK_HD int at(int k) { return (k % N1) * RS + phys(k / N1); }
    // twiddle buffers (host fills them: kernels.hip fill_fft_tables)
    static constexpr int AROWS = 4;                             // stage A rows k1 = 1, 2, 4, 8
    static constexpr int TWA = (N1 > 1) ? AROWS * N2 : 0;
    static constexpr int TWB = make_plan(LOG2N2).total;         // stage B pass tables
};

// Stage-A twiddles of one column: the rows k1 = 1, 2, 4, 8 of the table, enough to form
// W^(n2*k1) for every k1 < 16.  They depend only on the column, so a kernel that transforms
// many blocks loads them once (StageATw) and keeps them in registers.
template <int LOG2P>
struct StageATw {
    float2 w[4];
};
template <int LOG2P>
FK_D StageATw<LOG2P> load_stage_a_tw(const float2* __restrict__ twa, int n2) {
    using G = WaveGeom<LOG2P>;
    StageATw<LOG2P> t;
#pragma unroll
    for (int r = 1, row = 0; row < 4; r *= 2, ++row) t.w[row] = (r < G::N1) ? twa[row * G::N2 + n2] : float2{1.f, 0.f};
    return t;
}

// Stage A for one column n2 whose N1 inputs z[n1*N2 + n2] are in v.
//   tw: rows 0..3 <-> k1 = 1,2,4,8 of exp(-2*pi*i*n2*k1/N), N = the transform length
template <int LOG2P, bool INV>
FK_D void stage_a_column(float2* s, const StageATw<LOG2P>& tw, int n2, float2 (&v)[WaveGeom<LOG2P>::N1]) {
    using G = WaveGeom<LOG2P>;
    constexpr int N1 = G::N1;
    if constexpr (N1 > 1) {
        dft<N1, INV>(v);
        float2 w[N1];
#pragma unroll
        for (int r = 1, row = 0; r < N1; r *= 2, ++row) w[r] = tw.w[row];
#pragma unroll
        for (int r = 3; r < N1; ++r) {
            if ((r & (r - 1)) != 0) {
                int top = 1;
                while (top * 2 <= r) top *= 2;
                w[r] = cmul(w[top], w[r - top]);
            }
        }
#pragma unroll
        for (int r = 1; r < N1; ++r) v[r] = INV ? cmulc(v[r], w[r]) : cmul(v[r], w[r]);
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) s[k1 * G::RS + phys(n2)] = v[k1];
}
template <int LOG2P, bool INV>
FK_D void stage_a_column(float2* s, const float2* __restrict__ twa, int n2, float2 (&v)[WaveGeom<LOG2P>::N1]) {
    const StageATw<LOG2P> tw = load_stage_a_tw<LOG2P>(twa, n2);
    stage_a_column<LOG2P, INV>(s, tw, n2, v);
}

// Stage A for the columns this thread owns (n2 = tid + c*NT).  src(i) -> z[i].
// All loads are issued before the first butterfly.
template <int LOG2P, bool INV, class Src>
FK_D void stage_a(float2* s, const float2* __restrict__ twa, int tid, Src&& src) {
    using G = WaveGeom<LOG2P>;
    constexpr int N1 = G::N1, N2 = G::N2, NT = G::NT, COLS = G::COLS;
    constexpr bool GUARD = (N2 % NT) != 0;
    float2 v[COLS][N1];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int n2 = tid + c * NT;
        if (!GUARD || n2 < N2) {
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) v[c][n1] = src(n1 * N2 + n2);
        }
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int n2 = tid + c * NT;
        if (!GUARD || n2 < N2) stage_a_column<LOG2P, INV>(s, twa, n2, v[c]);
    }
}

// Stage B: every wavefront transforms its own row.
//   XL: the last two passes of a 1024-point row fused through a cross-lane transpose
//   UPPER (with XL): only the upper half of every row is written by the last pass
template <int LOG2P, bool INV, bool XL = true, bool UPPER = false>
FK_D void stage_b(float2* s, const float2* __restrict__ twb, int tid) {
    using G = WaveGeom<LOG2P>;
    float2* row = s + (tid >> 6) * G::RS;
    lds_fft<G::LOG2N2, 64, INV, WaveSync, XL, UPPER>(row, twb, tid & 63);
}

#endif  // __HIPCC__

}  // namespace fk
