// mac_walk3.hip — K2's whole-call walk with THREE real multiply-adds per complex one (round 5).
//
// The walk of kernels.hip (`mac_walk_kernel`: one lane owns one bin of one (stream, output) for a whole call, the filter's
// rows of that bin and a window of the stream's last spectra in registers) spends its time issuing arithmetic wherever the
// filter is long or an output has several paths (cfg4's 65 rows on two lanes, a 2 x 2 matrix): 66 packed FMAs of 99
// instructions per step at 33 rows per lane, and a wavefront pays about five cycles per instruction whatever it is
// (tools/micro/valu_rates.hip).  This form does the same sum with three quarters of the multiplies.  With x = a + ib a
// window element and g = c + id a filter row,
//     T = c (a + b)      R = -b (c + d)      I = a (d - c)        Re(x g) = T + R      Im(x g) = T + I
// and all three are sums over the rows, so a lane keeps, per filter row, (d - c, -(c + d)) as one register pair and c in
// pairs of rows, per window element (a, b) as loaded and s = a + b in pairs of elements — one add when the element arrives,
// used by every row it meets — and accumulates
//     (I, R) += (a, b) * (d - c, -(c + d))       one v_pk_fma_f32 per row, plain operands
//     T      += s c                              one v_pk_fma_f32 per TWO rows
// 1.5 packed FMAs per complex multiply-add instead of 2.  The packing of T is where the structure of a convolution gets
// in the way: row j meets element u - j at step u, so two neighbouring rows meet two neighbouring elements in REVERSE
// order, and whether they sit in one aligned register pair alternates with the parity of u.  At odd steps they do — the
// crossed form `op_sel:[1,0,0] op_sel_hi:[0,1,1]` multiplies (s.hi c.lo, s.lo c.hi), both halves belong to T(u).  At even
// steps the aligned pairs give (s[u-2q] c[2q], s[u-2q+1] c[2q+1]): the low halves are the even rows' products of T(u), the
// high halves are the odd rows' products of T(u + 2) — every one of them, and every element they need has arrived (the
// walk waits for step u + 1's element at step u anyway).  So an even step's high half is CARRIED two steps:
// T(u) = low half now + high half of two steps ago; the first even step's carry is computed from the history before the
// loop.  W = KR + D is even, so a slot's parity is its block's parity for the whole call.
//   Bin 0 is packed (DC, Nyquist: two real spectra, products (sum a c, sum b d)): that lane keeps (c, d) in place of
//   (d - c, -(c + d)) and zeros in place of c, and takes (I, R) as its result — two selects per step, as before.
//   Rounding: the three sums have the magnitude of |x||g| where the four-FMA form's have |Re|, |Im| — measured agreement
//   with the four-FMA walk and with float64: tests/test_forms_gpu.py.
// Everything else — window as prefetch ring, pinned loads with exact s_waitcnt, several lanes per bin with DPP hand-down,
// several paths per output, time tiles — is the walk of kernels.hip; tools/check_isa.py simulates these loops too.
//   What a step costs beside its arithmetic (round 5, measured in tools/micro/valu_rates.hip: every instruction a wavefront
//   issues costs it ~3.4 cycles at two waves per SIMD, a scalar one as much as a vector one, a branch that is never taken
//   ~16): the rows are BUFFER accesses with a 32-bit scalar row offset (one s_add where a 64-bit pointer took two, 32-bit
//   compares for the ring's wrap and the tile's end), a store past the tile's end is out of the buffer's range and dropped
//   by the hardware, so the walk asks once per group of up to 8 steps whether it is over, and the hand-down / the sum
//   across a bin's lanes ride as DPP operands on the select / the add.  73.8 -> 66.0 instructions per step at 33 rows on
//   one lane, 83.6 -> 73.2 on two.
#include "walk_common.hpp"

#ifndef FOLVE_W3_D33
#define FOLVE_W3_D33 7            // prefetch depth of the 33-row windows with ONE lane per path (odd: 33 + D must be even; the forms
                                 // with a hand-down have no registers for more than 7).  Measured (-DFOLVE_W3_D33=9): cfg3's K2 and the 2 x 2 matrix
                                 // unchanged — the waits at s_waitcnt are not the loads' latency
#endif
#ifndef FOLVE_W3_NIR
#define FOLVE_W3_NIR 2            // (I, R) accumulators per lane: 2 (four measured: two packed adds per step more — cfg4 K2 +2 %, a 2 x 2 matrix +3.6 %)
#endif

namespace fk {
namespace {

// s = a + b of a window element, as ONE v_add_f32 (left to itself hipcc pairs the adds of two steps into a v_pk_add_f32
// behind four v_mov)
__device__ __forceinline__ float add_ab(const v2f& x) {
    float s;
    asm("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(x.x), "v"(x.y));
    return s;
}
__device__ __forceinline__ void pk_mul(v2f& d, const v2f& a, const v2f& b) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }
__device__ __forceinline__ void pk_fma(v2f& d, const v2f& a, const v2f& b) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); }
// crossed: d.lo (+)= a.hi * b.lo, d.hi (+)= a.lo * b.hi
__device__ __forceinline__ void pk_mul_x(v2f& d, const v2f& a, const v2f& b) {
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pk_fma_x(v2f& d, const v2f& a, const v2f& b) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(d) : "v"(a), "v"(b));
}

// The hand-down of a lane set: every lane but the set's head takes the element that leaves its neighbour's window (one DPP
// row shift per register), the head keeps what its load brought.  (A DPP bank mask cannot do the select: a bank is four
// CONSECUTIVE lanes of a row, not lane % 4.)
// One v_cndmask_b32_dpp per register does both the shift and the select (D = VCC ? S1 : dpp(S0): the heads keep theirs;
// a row's first lane has no source and is not written — it is a head): 2 VALU per step where a DPP move and a select each
// took 4 (cfg4's K2 is 84 % VALU issue, SQ counters).  `heads`: the mask of the lane sets' first lanes, a scalar pair.
#ifndef FOLVE_W3_PLAIN_DPP
__device__ __forceinline__ void hand_down(v2f& mine, const v2f& leaving, bool, unsigned long long heads) {
    asm("s_mov_b64 vcc, %4\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %3, %1, vcc row_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(mine.x), "+v"(mine.y) : "v"(leaving.x), "v"(leaving.y), "s"(heads) : "vcc");
}
// Inside the loop the mask stays in VCC from the loop's head on (the marker statement sets it once per round of W steps):
// nothing between the head and the round's last hand-down writes VCC — the steps are VALU / SALU-on-SCC / VMEM only —
// and tools/check_isa.py holds the built code to that (`vcc_writers_in_loop`).  One scalar instruction per step less.
__device__ __forceinline__ void hand_down_vcc_live(v2f& mine, const v2f& leaving) {
    asm volatile("v_cndmask_b32_dpp %0, %2, %0, vcc row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %1, %3, %1, vcc row_shr:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(mine.x), "+v"(mine.y) : "v"(leaving.x), "v"(leaving.y));
}
// sum += its neighbour's (lane ^ 1 / lane ^ 2), every lane: the shift rides on the add's first operand.  (s_nop 1: a DPP
// operand written by the VALU instruction in front needs two wait states, and inside an asm block nobody counts them.)
template <int CTRL0, int CTRL1, int CTRL2, int CTRL3>
__device__ __forceinline__ void add_across(v2f& sum) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf"
        : "+v"(sum.x), "+v"(sum.y) : "n"(CTRL0), "n"(CTRL1), "n"(CTRL2), "n"(CTRL3));
}
#else
__device__ __forceinline__ void hand_down(v2f& mine, const v2f& leaving, bool head, unsigned long long) {
    const float hx = dpp_row_shr1(leaving.x), hy = dpp_row_shr1(leaving.y);
    mine.x = head ? mine.x : hx;
    mine.y = head ? mine.y : hy;
}
#endif

#ifndef FOLVE_W3_WG
#define FOLVE_W3_WG 256          // threads per workgroup (the walk uses neither LDS nor barriers: the size only shapes dispatch and DRAM
                                 // locality).  Measured (-DFOLVE_W3_WG=512, tools/build_variant.sh): cfg3's K2 0.93 -> 1.02 ms, cfg4 the same
#endif
#define FK_W3_BOUNDS __launch_bounds__(FOLVE_W3_WG, ((3 * (KR + D) + 3 * KR + 1 + 24 + (LPB > 1 ? 8 : 0) <= 168) ? 3 : 2) * 256 / FOLVE_W3_WG)
// NT: the walk's row loads AND its stores carry the non-temporal hint — for launches whose rows of X and Y are streams far
// larger than the 256 MB Infinity Cache (cfg3: 2.4 GB in, 2.1 GB out).  Both or neither: a float4 copy on this GPU moves 5.9 -
// 6.0 TB/s with the hint on both sides and 5.3 - 5.7 with it on one side or none (tools/micro/copy_rate.hip,
// profiles/r06_copy_rate.txt), and so does the walk: cfg3's K2 0.889 -> 0.859 ms with both, 0.897 / 0.900 with only the
// loads / only the stores (profiles/r06_ab_nt.txt).  A lone stream's Y fits the cache and K3 finds it there: plain.
template <int KR, int D, bool PIN, int LPB, int NP, bool NT>
__device__ __forceinline__ void walk3_body(const JobRef& jr, const FilterDev& f, float2* __restrict__ Y, int tiles, int tile_len) {
    constexpr int W = KR + D;
    constexpr int KRP = (KR + 1) / 2;                       // pairs of rows
    static_assert(W % 2 == 0, "a slot's parity must be its block's parity");
    static_assert(KR >= 2 && 2 * KRP <= W, "rows");
    static_assert(LPB == 1 || LPB == 2 || LPB == 4, "lanes per bin");
    static_assert((NP == 1 || NP == 2 || NP == 4) && NP <= LPB, "paths per output");
    constexpr int LPP = LPB / NP;                           // lanes per path
    const StreamJob job = fetch_job(jr);
    const int o = blockIdx.y / tiles;
    const int tb = (blockIdx.y - o * tiles) * tile_len;     // first block of this workgroup's time tile
    if (tb >= job.nblocks) return;
    const int nb = min(tile_len, job.nblocks - tb);
    const int P = f.P, K = f.K, ring = job.ring;
    // Workgroup ids go round the 8 XCDs (id % 8), each with its own L2: in plain grid order an XCD would get every eighth
    // 2 KB piece of every row.  The bin tiles are dealt so that every XCD gets a CONTIGUOUS eighth of each row instead
    // (cfg3's K2 -1.2 %, the step -0.4 %, four A/B pairs on one box).
    const int ntile = gridDim.x;
    const int tile = (ntile % 8 == 0) ? (int)(blockIdx.x % 8) * (ntile / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int tid = tile * blockDim.x + threadIdx.x;
    const int bin = tid / LPB, sub = tid % LPB;
    const int pi = sub / LPP;                               // which of the output's paths this lane works for
    const int jb = (sub % LPP) * KR;                        // this lane's first row of G
    const bool head = sub % LPP == 0;                       // the lane of its set that takes the loaded element
    const unsigned long long heads = LPP == 4 ? 0x1111111111111111ull : LPP == 2 ? 0x5555555555555555ull : ~0ull;
    const unsigned voff = (unsigned)bin * 8u;               // this thread's bin inside any spectrum row
    const bool packed = bin == 0;
    const int pe0 = f.out_first[o], pe1 = f.out_first[o + 1];
    const size_t yrow0 = (size_t)job.yunit0 + (size_t)o * job.nblocks + tb;
    if (pe1 > pe0) {
        const bool path_on = NP == 1 || pi < pe1 - pe0;
        const PathEntry pth = f.paths[NP == 1 ? pe0 : pe0 + (path_on ? pi : 0)];
        const float2* __restrict__ Hd = NP == 1 ? f.H + (size_t)pth.data * K * P : f.H;
        const float2* __restrict__ X = NP == 1 ? job.fdl + (size_t)pth.in_ch * ring * P : job.fdl;
        const unsigned voff_g = NP == 1 ? voff : voff + (unsigned)pth.data * (unsigned)K * (unsigned)P * 8u;
        const unsigned voff_x = NP == 1 ? voff : voff + (unsigned)pth.in_ch * (unsigned)ring * (unsigned)P * 8u;
        auto ldrow_g = [&](const float2* rowbase) -> v2f {
            return *(const FK_GLOBAL v2f*)((const FK_GLOBAL char*)rowbase + voff_g);
        };
        auto ldrow = [&](const float2* rowbase) -> v2f {
            return *(const FK_GLOBAL v2f*)((const FK_GLOBAL char*)rowbase + voff_x);
        };
        v2f ge[KR], cp[KRP], w[W], sp[W / 2];
        const v2f sel = packed ? v2f{0.f, 1.f} : v2f{1.f, 0.f};   // (see the reduction)
        // the filter's rows of this bin: (c, d) -> (d - c, -(c + d)) and c; rows that do not exist are zeros
        static_for<KRP>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            cp[q] = v2f{0.f, 0.f};
        });
        // Every load of the prologue is issued before anything uses one (a scheduling barrier between the two halves): left to
        // itself the scheduler has, in some instantiations, paired each load with its use — fifty round trips to memory one
        // after the other in front of a lone stream's 32 steps (<26, 8>, <29, 7>: K2 20 -> 28 us).
        static_for<KR>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const bool on = path_on && jb + j < K;
            ge[j] = ldrow_g(Hd + (size_t)(on ? jb + j : 0) * P);
        });
        // history: block -j of the lane's frame (the window of lane `sub` starts KR * sub blocks back) in slot W - j, its s beside
        // it; slot D (block -KR: the first element handed down to the next lane, a zero with one lane per bin) too
        static_for<KR>([&](auto jc) {
            constexpr int j = decltype(jc)::value + 1;      // 1 .. KR
            constexpr int slot = W - j;
            if constexpr (!(LPB == 1 && j == KR)) {
                const bool on = path_on && jb + j < K;      // (an element no row will ever meet is a zero)
                w[slot] = ldrow(X + (size_t)ring_slot(job.slot0, on ? tb - j - jb : tb, ring) * P);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<KR>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const bool on = path_on && jb + j < K;
            const float c = on ? ge[j].x : 0.f, d = on ? ge[j].y : 0.f;
            ge[j].x = packed ? c : d - c;
            ge[j].y = packed ? d : -(c + d);
            if constexpr (j % 2 == 0) cp[j / 2].x = packed ? 0.f : c;
            else cp[j / 2].y = packed ? 0.f : c;
        });
        static_for<W / 2>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            sp[p] = v2f{0.f, 0.f};                           // (slots still in flight hold finite numbers: they are multiplied, by zeros)
        });
        static_for<KR>([&](auto jc) {
            constexpr int j = decltype(jc)::value + 1;      // 1 .. KR
            constexpr int slot = W - j;
            if constexpr (LPB == 1 && j == KR) {
                w[slot] = v2f{0.f, 0.f};
            } else {
                const bool on = path_on && jb + j < K;
                w[slot] = on ? w[slot] : v2f{0.f, 0.f};
                if constexpr (slot % 2 == 0) sp[slot / 2].x = add_ab(w[slot]);
                else sp[slot / 2].y = add_ab(w[slot]);
            }
        });
        // The walk's rows are addressed as BUFFER accesses: a fixed resource (the stream's ring / this tile's rows of Y) and a
        // 32-bit scalar offset that moves one row per step — one s_add_u32 where a 64-bit row pointer took two instructions to
        // move, and 32-bit compares for the ring's wrap and the tile's end.  A wavefront pays ~3.4 cycles for every instruction
        // it issues, scalar ones included (tools/micro/valu_rates.hip): the pointer arithmetic was 12 of ~75 per step.
        auto make_rsrc = [](const void* base, unsigned bytes) {   // raw buffer of `bytes` bytes: stride 0, gfx9's format word
            // (the base is uniform, but where lanes also compute their own addresses from it the compiler keeps its copy in
            // vector registers, and an asm operand gets no readfirstlane by itself)
            const uintptr_t a = reinterpret_cast<uintptr_t>(base);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
        };
        const unsigned rb = (unsigned)P * 8u;                // a row, in bytes
        const auto xres = make_rsrc(X, 0xffffffffu);
        auto issue = [&](v2f& dst, unsigned row_off) {
            if constexpr (PIN && NT) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen nt" : "=&v"(dst) : "v"(voff_x), "s"(xres), "s"(row_off) : "memory");
            else if constexpr (PIN) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff_x), "s"(xres), "s"(row_off) : "memory");
            else dst = *(const FK_GLOBAL v2f*)((const FK_GLOBAL char*)X + row_off + voff_x);
        };
        unsigned xo = (unsigned)ring_slot(job.slot0, tb, ring) * rb;       // the row the next load takes
        const unsigned ring_bytes = (unsigned)ring * rb;
        // Past the tile's last block the walk stays on it (re-read from L2, never used; running on into the ring's next rows
        // would be D rows of HBM traffic per wavefront and call: 57 MB of cfg3's 4.6 GB).  Selects, not a branch: a branch that
        // is never taken costs a wavefront ~16 cycles, a scalar instruction ~3.4 (tools/micro/valu_rates.hip).
        // ... where the walk's time is memory (one lane per bin, one path per output).  Where it is instruction issue
        // (several lanes per bin, filter matrices: 84 % VALU issue, SQ counters) the two scalar instructions per step cost
        // more than the few rows a tile reads past its end — any row of the ring is mapped memory —: no clamp there.
#ifdef FOLVE_W3_CLAMP_ALL
        constexpr bool CLAMP = true;
#else
        constexpr bool CLAMP = LPB == 1 && NP == 1;
#endif
        const unsigned xlast = (unsigned)ring_slot(job.slot0, tb + nb - 1, ring) * rb;
        auto advance = [&]() {
            unsigned nx = xo + rb;
            nx = (nx == ring_bytes) ? 0u : nx;
            if constexpr (CLAMP) xo = (xo == xlast) ? xo : nx;
            else xo = nx;
        };
        // the first even step's carry: the odd rows' products of T(0), all from the history
        v2f carry;
        static_for<KRP>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int p = W / 2 - 1 - q;                 // slots (W - 2 - 2q, W - 1 - 2q): blocks (-2 - 2q, -1 - 2q)
            if constexpr (q == 0) pk_mul(carry, sp[p], cp[q]);
            else pk_fma(carry, sp[p], cp[q]);
        });
        if constexpr (PIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // G and the history have arrived: the count starts here
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue(w[d], xo);
            advance();
        }
        if constexpr (PIN) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(D - 1) : "memory");
        if constexpr (LPP > 1) hand_down(w[0], w[D], head, heads);
        sp[0].x = add_ab(w[0]);
        // The tile's rows of Y are a buffer of exactly nb rows and the row offset rides in the store's VECTOR offset (one
        // v_add per step): a store past the tile's end is out of range and the hardware drops it (gfx9 checks the vector
        // offset against num_records, not the scalar one).  So the walk need not ask at every step whether it is over — a
        // compare and a branch, ~20 cycles of a step's ~250 — but once per group of G steps; the up to G - 1 steps beyond the
        // end compute on the (clamped) last rows and store nowhere.
#ifdef FOLVE_W3_G
        constexpr int G = FOLVE_W3_G;
#else
        constexpr int G = W % 8 == 0 ? 8 : W % 4 == 0 ? 4 : 2;
#endif
        float2* const ybase = Y + yrow0 * P;                // this tile's rows of Y: a row per step
        const auto yres = make_rsrc(ybase, (unsigned)nb * rb);
        unsigned voff_y = voff;
        int left = nb;                                       // steps to go, counted per group
        for (;;) {                                           // (nb >= 1)
#if defined(FOLVE_W3_PLAIN_DPP) || defined(FOLVE_W3_VCC_PER_STEP)
            asm volatile("s_setprio 0");                     // (does nothing: marks the loop's head for tools/check_isa.py, whatever the block layout)
#else
            // (only in the pinned build, whose loop is VALU / SALU-on-SCC / VMEM by hand: with compiler-scheduled accesses the
            // range check of the store is a v_cmp into VCC — the NO_PIN fallback sets the mask per hand-down)
            if constexpr (LPP > 1 && PIN) asm volatile("s_setprio 0\n\ts_mov_b64 vcc, %0" : : "s"(heads) : "vcc");   // the marker, and the heads' mask for this round's hand-downs
            else asm volatile("s_setprio 0");                // (does nothing: marks the loop's head for tools/check_isa.py, whatever the block layout)
#endif
            const bool more = static_all<W / G>([&](auto gc) {
              static_for<G>([&](auto ic) {
                constexpr int u = decltype(gc)::value * G + decltype(ic)::value;
                constexpr int un = (u + 1) % W;              // the next step's slot
#ifndef FOLVE_W3_NOLOAD                                      // what-if build: ... without its in-loop loads
                issue(w[(u + D) % W], xo);
#endif
                advance();
                if constexpr (PIN) {
#if defined(FOLVE_W3_NOLOAD)
                    constexpr int N = 63;
#elif defined(FOLVE_W3_NOSTORE)
                    constexpr int N = D - 1;
#else
                    constexpr int N = (D - 1) + (u < D - 1 ? u : D - 1);
#endif
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[un]) : "n"(N) : "memory");
                }
#if defined(FOLVE_W3_PLAIN_DPP) || defined(FOLVE_W3_VCC_PER_STEP)
                if constexpr (LPP > 1) hand_down(w[un], w[(u + 1 + D) % W], head, heads);
#else
                if constexpr (LPP > 1 && PIN) hand_down_vcc_live(w[un], w[(u + 1 + D) % W]);
                else if constexpr (LPP > 1) hand_down(w[un], w[(u + 1 + D) % W], head, heads);
#endif
                if constexpr (un % 2 == 0) sp[un / 2].x = add_ab(w[un]);
                else sp[un / 2].y = add_ab(w[un]);
                // Two (I, R) accumulators and two for T.  What decides the mix of compiler-generated and inline-asm arithmetic is
                // hipcc's hazard recognizer: between two inline-asm statements that touch one register it counts no wait states
                // (it cannot see what they are) and pads an s_nop before every re-use — eight per step when all fifty products were
                // asm, whatever the number of accumulators.  So the (I, R) products, plain packed FMAs, are the compiler's
                // (__builtin_elementwise_fma: it knows they need nothing), and only the T products, whose operand halves are
                // crossed at odd steps, stay asm — two compiler FMAs between any two of them.  The asm statements are volatile so
                // that they keep their order among themselves.
                constexpr int NIR = FOLVE_W3_NIR;
                v2f ir[4], tt[2];
                static_for<KRP>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    constexpr int j0 = 2 * q, j1 = 2 * q + 1;
                    // (the (I, R) products are plain packed FMAs: left to the compiler, which knows that they need no wait states;
                    // between inline-asm statements its hazard recognizer counts none and pads)
                    if constexpr (j0 < NIR) ir[j0 % NIR] = w[(u - j0 + 2 * W) % W] * ge[j0];
                    else ir[j0 % NIR] = __builtin_elementwise_fma(w[(u - j0 + 2 * W) % W], ge[j0], ir[j0 % NIR]);
                    if constexpr (j1 < KR) {
                        if constexpr (j1 < NIR) ir[j1 % NIR] = w[(u - j1 + 2 * W) % W] * ge[j1];
                        else ir[j1 % NIR] = __builtin_elementwise_fma(w[(u - j1 + 2 * W) % W], ge[j1], ir[j1 % NIR]);
                    }
                    if constexpr (u % 2 == 0) {              // aligned pairs: low half for T(u), high half for T(u + 2)
                        constexpr int p = ((u - 2 * q + 2 * W) % W) / 2;
                        if constexpr (q < 2) pk_mul(tt[q % 2], sp[p], cp[q]);
                        else pk_fma(tt[q % 2], sp[p], cp[q]);
                    } else {                                 // crossed: both halves for T(u)
                        constexpr int p = ((u - 2 * q - 1 + 2 * W) % W) / 2;
                        if constexpr (q < 2) pk_mul_x(tt[q % 2], sp[p], cp[q]);
                        else pk_fma_x(tt[q % 2], sp[p], cp[q]);
                    }
                });
                // The reduction is ONE asm block: between separate statements hipcc's hazard recognizer pads an s_nop in front of
                // every dependent packed add (eight per step), and the hardware interlocks plain VALU dependencies by itself.
                //   ir[0] = ir[0] + ir[1] (+ ir[2] + ir[3]) = (I, R);  tt[0] = tt[0] + tt[1];
                //   (T, T) = an even step: low half now + the high half carried from two steps ago; an odd one: low + high;
                //   sum = (T, T) + (R, I) * sel.lo + (I, R) * sel.hi      sel = (1, 0), the packed lane's (0, 1): it takes (I, R)
                //   as they are (its T is zero: it keeps zeros in place of c) — two packed FMAs where selects would need the
                //   halves of a register pair, which an asm operand cannot name.
                v2f sum;
                static_assert(KR >= 4 && KRP >= 2, "up to four (I, R) accumulators and two for T are in use");
#define FK_W3_HEAD4                           \
    "v_pk_add_f32 %1, %1, %4\n\t"             \
    "v_pk_add_f32 %3, %3, %8\n\t"             \
    "v_pk_add_f32 %2, %2, %5\n\t"             \
    "v_pk_add_f32 %1, %1, %3\n\t"
#define FK_W3_HEAD2                           \
    "v_pk_add_f32 %1, %1, %4\n\t"             \
    "v_pk_add_f32 %2, %2, %5\n\t"
#define FK_W3_TAIL                                                           \
    "v_pk_fma_f32 %0, %1, %7, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"       \
    "v_pk_fma_f32 %0, %1, %7, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
#define FK_W3_EVEN "v_pk_add_f32 %0, %2, %6 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
#define FK_W3_ODD "v_pk_add_f32 %0, %2, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                if constexpr (NIR == 2) ir[2] = ir[3] = v2f{0.f, 0.f};       // (operands of the asm below, not used by its two-accumulator form)
#if !defined(FOLVE_W3_PLAIN_DPP) && !defined(FOLVE_W3_SUM_ANYWHERE)
                // Several lanes per bin: the sum over the bin's lanes rides in the same block, on `sum` in a FIXED register pair
                // (v[166:167]: an asm operand cannot name the halves of a pair, a physical register can) — as two scalar
                // operands the allocator split the pair and moved a half out and back around the DPP adds in some
                // instantiations (1.2 v_mov per step).  (s_nop 1: a DPP operand written by the VALU instruction in front needs
                // two wait states.)
#define FK_W3_ACROSS2 "\n\ts_nop 1\n\t"                                                             \
    "v_add_f32_dpp v166, v166, v166 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"              \
    "v_add_f32_dpp v167, v167, v167 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define FK_W3_ACROSS4 FK_W3_ACROSS2 "\n\ts_nop 1\n\t"                                               \
    "v_add_f32_dpp v166, v166, v166 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"              \
    "v_add_f32_dpp v167, v167, v167 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#define FK_W3_FIXED(STEP, ACROSS)                                                                    \
    asm("v_pk_add_f32 %1, %1, %4\n\t"                                                                \
        "v_pk_add_f32 %2, %2, %5\n\t" STEP                                                           \
        "v_pk_fma_f32 v[166:167], %1, %7, v[166:167] op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"           \
        "v_pk_fma_f32 v[166:167], %1, %7, v[166:167] op_sel:[0,1,0] op_sel_hi:[1,1,1]" ACROSS         \
        : "=&{v[166:167]}"(sum), "+v"(ir[0]), "+v"(tt[0]) : "v"(ir[0]), "v"(ir[1]), "v"(tt[1]), "v"(carry), "v"(sel))
                if constexpr (LPB >= 2 && NIR == 2) {
                    if constexpr (u % 2 == 0) {
                        if constexpr (LPB == 2) FK_W3_FIXED("v_pk_add_f32 v[166:167], %2, %6 op_sel:[0,1] op_sel_hi:[0,1]\n\t", FK_W3_ACROSS2);
                        else FK_W3_FIXED("v_pk_add_f32 v[166:167], %2, %6 op_sel:[0,1] op_sel_hi:[0,1]\n\t", FK_W3_ACROSS4);
                        carry = tt[0];
                    } else {
                        if constexpr (LPB == 2) FK_W3_FIXED("v_pk_add_f32 v[166:167], %2, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t", FK_W3_ACROSS2);
                        else FK_W3_FIXED("v_pk_add_f32 v[166:167], %2, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t", FK_W3_ACROSS4);
                    }
                } else
#undef FK_W3_ACROSS2
#undef FK_W3_ACROSS4
#endif
                if constexpr (u % 2 == 0) {
                    if constexpr (NIR == 4)
                        asm(FK_W3_HEAD4 FK_W3_EVEN FK_W3_TAIL : "=&v"(sum), "+v"(ir[0]), "+v"(tt[0]), "+v"(ir[2])
                            : "v"(ir[1]), "v"(tt[1]), "v"(carry), "v"(sel), "v"(ir[3]));
                    else
                        asm(FK_W3_HEAD2 FK_W3_EVEN FK_W3_TAIL : "=&v"(sum), "+v"(ir[0]), "+v"(tt[0]) : "v"(ir[0]), "v"(ir[1]), "v"(tt[1]), "v"(carry), "v"(sel));
                    carry = tt[0];
                } else {
                    if constexpr (NIR == 4)
                        asm(FK_W3_HEAD4 FK_W3_ODD FK_W3_TAIL : "=&v"(sum), "+v"(ir[0]), "+v"(tt[0]), "+v"(ir[2])
                            : "v"(ir[1]), "v"(tt[1]), "v"(carry), "v"(sel), "v"(ir[3]));
                    else
                        asm(FK_W3_HEAD2 FK_W3_ODD FK_W3_TAIL : "=&v"(sum), "+v"(ir[0]), "+v"(tt[0]) : "v"(ir[0]), "v"(ir[1]), "v"(tt[1]), "v"(carry), "v"(sel));
                }
#undef FK_W3_HEAD4
#undef FK_W3_HEAD2
#undef FK_W3_EVEN
#undef FK_W3_ODD
#undef FK_W3_TAIL
#undef FK_W3_FIXED
#if !defined(FOLVE_W3_PLAIN_DPP) && !defined(FOLVE_W3_SUM_ANYWHERE)
                if constexpr (NIR != 2) {                              // (the two-accumulator form has summed across its lanes above)
                    if constexpr (LPB >= 2) add_across<1, 0, 3, 2>(sum);
                    if constexpr (LPB >= 4) add_across<2, 3, 0, 1>(sum);
                }
#elif !defined(FOLVE_W3_PLAIN_DPP)
                if constexpr (LPB >= 2) add_across<1, 0, 3, 2>(sum);   // the group's partial sums: every lane ends up with the total
                if constexpr (LPB >= 4) add_across<2, 3, 0, 1>(sum);
#else
                if constexpr (LPB >= 2) {
                    sum.x += dpp_quad<0xB1>(sum.x); sum.y += dpp_quad<0xB1>(sum.y);
                }
                if constexpr (LPB >= 4) {
                    sum.x += dpp_quad<0x4E>(sum.x); sum.y += dpp_quad<0x4E>(sum.y);
                }
#endif
#ifdef FOLVE_W3_NOSTORE                                      // what-if build (tools/build_variant.sh): the walk without its stores
                asm volatile("" : : "v"(sum));
#else
                if constexpr (PIN && NT) asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen nt" : : "v"(sum), "v"(voff_y), "s"(yres) : "memory");
                else if constexpr (PIN) asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(sum), "v"(voff_y), "s"(yres) : "memory");
                else if (voff_y < (unsigned)nb * rb) *(FK_GLOBAL v2f*)((FK_GLOBAL char*)ybase + voff_y) = sum;
#endif
                voff_y += rb;
              });
              left -= G;
              return left > 0;
            });
            if (!more) break;
        }
    } else {
        // an output without an input path: silence
        for (int t = 0; t < nb; ++t) gst_v2(Y + (yrow0 + t) * P + bin, v2f{0.f, 0.f});
    }
}

template <int KR, int D, bool PIN, int LPB = 1, int NP = 1>
__global__ FK_W3_BOUNDS void mac_walk3_kernel(JobRef jr, FilterDev f, float2* __restrict__ Y, int tiles, int tile_len) {
    walk3_body<KR, D, PIN, LPB, NP, false>(jr, f, Y, tiles, tile_len);
}
// the streaming form (rows with the non-temporal hint), a kernel of its own name so that a profile tells the two apart
template <int KR, int D, bool PIN, int LPB = 1, int NP = 1>
__global__ FK_W3_BOUNDS void mac_walk3_nt_kernel(JobRef jr, FilterDev f, float2* __restrict__ Y, int tiles, int tile_len) {
    walk3_body<KR, D, PIN, LPB, NP, true>(jr, f, Y, tiles, tile_len);
}

template <int KR, int D, int LPB, int NP = 1>
hipError_t launch3(const JobRef& jr, int njobs, const FilterDev& f, float2* Y, const WalkShape& w, const Tuning& tn, hipStream_t st) {
    dim3 grid(f.P * LPB / FOLVE_W3_WG, f.cout * w.tiles, njobs), block(FOLVE_W3_WG);
#ifdef FOLVE_WALK_NO_PIN
    constexpr bool kPin = false;
#else
    constexpr bool kPin = true;
#endif
    // the streaming form where this launch's rows of Y (and as many of X) cannot stay in the 256 MB Infinity Cache anyway
    if constexpr (LPB == 1 && NP == 1 && kPin) {                     // (every rung of the one-lane ladder: their time is memory)
        const unsigned long long y_bytes = (unsigned long long)njobs * f.cout * w.tiles * w.tile_len * f.P * 8ull;
#ifndef FOLVE_W3_NO_NT
        if (tn.walk_nt == 2 || (tn.walk_nt == 0 && y_bytes > (192ull << 20))) {
            FK_LAUNCH(1, (mac_walk3_nt_kernel<KR, D, kPin, LPB, NP>), grid, block, st, jr, f, Y, w.tiles, w.tile_len);
            return hipGetLastError();
        }
#endif
    }
    FK_LAUNCH(1, (mac_walk3_kernel<KR, D, kPin, LPB, NP>), grid, block, st, jr, f, Y, w.tiles, w.tile_len);
    return hipGetLastError();
}

}  // namespace

// the instantiated shapes: the forms whose time is arithmetic (long filters on several lanes, several paths per output)
// and the one-lane ladder of lone streams
bool walk3_has(int kr, int lpb, int np) {
    if (np == 1 && lpb == 1) return kr == 9 || kr == 13 || kr == 17 || kr == 21 || kr == 26 || kr == 29 || kr == 33;
    if (np == 1 && lpb == 2) return kr == 17 || kr == 33;
    if (np == 1 && lpb == 4) return kr == 9 || kr == 17 || kr == 33;
    if (np == 2) return (lpb == 2 || lpb == 4) && (kr == 17 || kr == 33);
    if (np == 4) return lpb == 4 && (kr == 9 || kr == 17 || kr == 33);
    return false;
}

hipError_t launch_walk3(const StreamJob* jobs, int njobs, const FilterDev& f, float2* Y, const WalkShape& ws, const Tuning& tn,
                        hipStream_t st) {
    const JobRef jr = make_job_ref(jobs, tn);
    if (!walk3_has(ws.kr, ws.lpb, ws.np)) return hipErrorInvalidValue;
    if (ws.np == 2 && ws.lpb == 2) return ws.kr == 17 ? launch3<17, 15, 2, 2>(jr, njobs, f, Y, ws, tn, st) : launch3<33, FOLVE_W3_D33, 2, 2>(jr, njobs, f, Y, ws, tn, st);
    if (ws.np == 2) return ws.kr == 17 ? launch3<17, 15, 4, 2>(jr, njobs, f, Y, ws, tn, st) : launch3<33, FOLVE_W3_D33, 4, 2>(jr, njobs, f, Y, ws, tn, st);
    if (ws.np == 4) {
        if (ws.kr == 9) return launch3<9, 15, 4, 4>(jr, njobs, f, Y, ws, tn, st);
        if (ws.kr == 17) return launch3<17, 15, 4, 4>(jr, njobs, f, Y, ws, tn, st);
        return launch3<33, FOLVE_W3_D33, 4, 4>(jr, njobs, f, Y, ws, tn, st);
    }
    if (ws.lpb == 1) {
        switch (ws.kr) {
            case 9: return launch3<9, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            case 13: return launch3<13, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            case 17: return launch3<17, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            case 21: return launch3<21, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            case 26: return launch3<26, 8, 1>(jr, njobs, f, Y, ws, tn, st);
            case 29: return launch3<29, 7, 1>(jr, njobs, f, Y, ws, tn, st);
            default: return launch3<33, FOLVE_W3_D33, 1>(jr, njobs, f, Y, ws, tn, st);
        }
    }
    if (ws.lpb == 2) return ws.kr == 17 ? launch3<17, 15, 2>(jr, njobs, f, Y, ws, tn, st) : launch3<33, FOLVE_W3_D33, 2>(jr, njobs, f, Y, ws, tn, st);
    if (ws.kr == 9) return launch3<9, 15, 4>(jr, njobs, f, Y, ws, tn, st);
    if (ws.kr == 17) return launch3<17, 15, 4>(jr, njobs, f, Y, ws, tn, st);
    return launch3<33, FOLVE_W3_D33, 4>(jr, njobs, f, Y, ws, tn, st);
}

}  // namespace fk
