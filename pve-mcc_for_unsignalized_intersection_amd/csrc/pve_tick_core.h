// pve_tick_core.h -- the environment tick as barrier-separated phases.
//
// One workgroup = one intersection; thread t = vehicle slot t (slots are kept sorted by
// (lane, j), so "order" in the reference's sequential loops == slot index, and every lane is a
// contiguous slot range).  The phases that only concern CONTROLLED vehicles (BUILD, WALK, REWARD, the
// dead-lock walk, the observation rows) run on a second, dense mapping: thread t also works for the
// t-th controlled vehicle of the intersection (slot_of[t]).  With ~50 controlled among ~85 alive vehicles
// in 128 slots all of them fall into the first wave, and the second wave skips those phases (its
// instructions were issued for a dozen active lanes before).  The reference's order-dependent
// (Gauss-Seidel) semantics are turned into closed-form parallel rules:
//   S1-S3  step()            ref traffic_interaction_scene.py:1501-1539  (in-lane brake chain)
//   SCAN   scene_update()    ref :233-334  (virtual lane, predecessor, 6 nearest, reward, XY hit)
//   FX     ordered effects   ref :333-359  (collision visibility by order, Done / finish)
//   LOCK   dead-lock scan    ref :365-370, :1469-1499 ; spawn ref :378-433
//   FIN    delete_vehicle()  ref :435-444  (stable re-pack) + coalesced write-back
//
// The same phase bodies are compiled (a) by hipcc as __device__ code called from the kernels in
// pve_hip.hip with __syncthreads() between phases, and (b) by g++ for the CPU *test* emulator
// (tests/emu), which runs each phase for t = 0..CAP-1 in a loop.  The product library contains
// only (a).
#pragma once
#include "pve_types.h"
#include <math.h>
#include <string.h>
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__)
#define PVE_AS4 __attribute__((address_space(4)))
#else
#define PVE_AS4
#endif
#if defined(__HIPCC__)
#undef PVE_HD
#define PVE_HD __device__ __forceinline__
#define PVE_DEVICE_CODE 1
#else
#define PVE_DEVICE_CODE 0
#endif

namespace pve {

typedef unsigned long long u64;

// ------------------------------------------------------------------ wave / block primitives
template <int NW> PVE_HD void vote(u64 *m, int t, bool f)
{
#if PVE_DEVICE_CODE
    u64 b = __ballot(f);               // 64-wide wavefront ballot
    if ((t & 63) == 0) m[t >> 6] = b;
#else
    if (f) m[t >> 6] |= 1ull << (t & 63);   // emulator: masks are zeroed per env
#endif
}
// several votes of one phase: all the ballots, then ONE guarded block in which the wave's first lane stores them (a vote() each
// is a block of its own: exec mask, branch, store, restore)
template <int NW, int N> PVE_HD void vote_many(u64 *const (&m)[N], int t, const bool (&f)[N])
{
#if PVE_DEVICE_CODE
    u64 b[N];
#pragma unroll
    for (int k = 0; k < N; k++) b[k] = __ballot(f[k]);
    if ((t & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) m[k][t >> 6] = b[k];
    }
#else
    for (int k = 0; k < N; k++) if (f[k]) m[k][t >> 6] |= 1ull << (t & 63);
#endif
}
PVE_HD bool mask_test(const u64 *m, int t) { return (m[t >> 6] >> (t & 63)) & 1ull; }
// The mask helpers read every word UNCONDITIONALLY (one broadcast LDS read each) and select with arithmetic: a word read
// under `if (t >= ...)` is a guarded basic block of its own (read, wait), and a tick calls these helpers dozens of times.
PVE_HD u64 below_sel(int rel)                                  // bits [0, rel) of a 64-bit word, rel may be <= 0 or >= 64
{
    return rel >= 64 ? ~0ull : (rel > 0 ? ((1ull << (rel & 63)) - 1ull) : 0ull);
}
// set bits of w at positions < rel (rel may be <= 0 or >= 64): the bits are shifted out at the top instead of being masked
// (clamp, shift, two popcounts, one select: 7 vector instructions instead of 13 for mask + and + popcount)
PVE_HD int popc_below(u64 w, int rel)
{
    const int r = rel < 0 ? 0 : (rel > 64 ? 64 : rel);
    const int c = __builtin_popcountll(w << ((64 - r) & 63));
    return r == 0 ? 0 : c;
}
template <int NW> PVE_HD int mask_below(const u64 *m, int t)   // set bits at positions < t
{
    int c = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) c += popc_below(m[k], t - k * 64);
    return c;
}
// set bits below t where t IS the calling thread: one word read and the wave's own lane-prefix count (v_mbcnt)
template <int NW> PVE_HD int mask_rank(const u64 *m, int t)
{
#if PVE_DEVICE_CODE
    // (NW <= 2; both words are read and selected: m may live in registers, where a per-lane index would go through scratch)
    const u64 w0 = m[0], w1 = m[NW - 1];
    const bool up = NW > 1 && t >= 64;
    const u64 w = up ? w1 : w0;
    int c = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(w >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)w, 0u));
    c += up ? __builtin_popcountll(w0) : 0;
    return c;
#else
    return mask_below<NW>(m, t);
#endif
}
template <int NW> PVE_HD int mask_count(const u64 *m)
{
    int c = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) c += __builtin_popcountll(m[k]);
    return c;
}
template <int NW> PVE_HD int mask_prev(const u64 *m, int t)    // highest set bit < t, or -1
{
    int best = -1;
#pragma unroll
    for (int k = 0; k < NW; k++) {                             // ascending: a hit in a higher word overrides
        const u64 b = m[k] & below_sel(t - k * 64);
        best = b ? (k * 64 + 63 - __builtin_clzll(b | 1ull)) : best;
    }
    return best;
}
PVE_HD void lds_add(int *p, int v)
{
#if PVE_DEVICE_CODE
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
PVE_HD int lds_claim(int *p, int n)  // reserves n consecutive units, returns the first (any order is fine)
{
#if PVE_DEVICE_CODE
    return atomicAdd(p, n);
#else
    int o = *p; *p += n; return o;
#endif
}
// RANK's claim protocol (ph_rank): "own stores, then the claim, then the fix-up stores".  Under the HIP memory model the claim
// is an acquire-release exchange at workgroup scope and every store that can meet another wave's store to the same word
// (the owner's mypos / s_slot, the fix-up rewrites of s_idx / s_slot / mypos) is a relaxed ATOMIC store: whoever gets the
// current tag back synchronises with every earlier claimant of that position (release sequence of the exchanges), so an
// owner's stores before its claim happen-before the fix-up stores of a later claimant, and two fix-ups of the same run write
// the same values, whatever their order.  No plain access races with them (the readers sit behind the phase barrier).
// On gfx950 outside threadgroup-split mode this costs one `s_waitcnt lgkmcnt(0)` in front of the ds_wrxchg_rtn_b32 (the
// one behind it is needed for the returned word anyway); the stores compile to the same ds_write_b8 / b16 / b32.
PVE_HD unsigned lds_xchg(unsigned *p, unsigned v)   // stores v, returns what was there
{
#if PVE_DEVICE_CODE
    return __hip_atomic_exchange(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    const unsigned o = *p; *p = v; return o;
#endif
}
template <class T> PVE_HD void lds_store_relaxed(T *p, T v)
{
#if PVE_DEVICE_CODE
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p = v;
#endif
}
// Global loads that may meet data ANOTHER workgroup of the same launch has stored (the persistent roll-out hands an
// intersection from one workgroup to the next): COH = agent-scope relaxed atomic loads (`global_load ... sc1`), which are
// served by the L2 instead of this CU's vector L1 -- the L1 is never refreshed by another CU's stores.
template <bool COH, class T> PVE_HD T gld(const T *p)
{
#if PVE_DEVICE_CODE
    if constexpr (COH) {
        static_assert(sizeof(T) == 4 || sizeof(T) == 8, "gld: 4- or 8-byte scalars");
        if constexpr (sizeof(T) == 8) {
            const unsigned long long u = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T v; __builtin_memcpy(&v, &u, 8); return v;
        } else {
            const unsigned u = __hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T v; __builtin_memcpy(&v, &u, 4); return v;
        }
    } else return *p;
#else
    return *p;
#endif
}
// The 7 x 28 state of one vehicle (ref :1325-1337): seven observation rows (28 values: 112 B as float32, 224 B as float64; rows are
// 16-byte aligned) gathered from srcs[0..6] (nullptr = an absent neighbour: zeros) into dst, in chunks of 112 bytes = 7 pieces of
// 16 bytes: the 7 loads of a chunk go out back to back, then its 7 stores.  (An element-wise `dst[k] = src[k]` loop is one full
// memory round trip per element -- the compiler cannot exclude that dst aliases src --, 196 of them per state: 305 us per tick of
// 4096 x 128 with the training outputs against 129 us in this form.  A two-deep pipeline -- the loads of chunk c + 1 in front of
// the stores of chunk c -- measured 126 us but costs 28 more registers: 8-15 spilled in the closed-loop and 8-lane trainer variants.)
// coh bit q: row q may have been stored by ANOTHER workgroup of this launch (cf. gld): agent-scope coherent loads (`sc1`, served
// by the L2) as ONE asm statement that ends with the wait for its own loads (the compiler does not track loads issued by asm).
template <class ROW> PVE_HD void gather_state(const ROW *const (&srcs)[NNB + 1], unsigned coh, ROW *dst)
{
#if PVE_DEVICE_CODE
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int CPR = (int)sizeof(ROW) * OBSW / 112;     // chunks per row: 1 (float32) or 2 (float64)
    constexpr int NC = (NNB + 1) * CPR;
    v4f buf[7];
    auto fetch = [&](int c, v4f (&q)[7]) {
        const ROW *row = srcs[c / CPR];
        const char *sp = (const char *)row + 112 * (c % CPR);
        if (!row) {
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = v4f{0.f, 0.f, 0.f, 0.f};
        } else if ((coh >> (c / CPR)) & 1u) {
            asm volatile("global_load_dwordx4 %0, %7, off sc1\n\tglobal_load_dwordx4 %1, %7, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %2, %7, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %7, off offset:48 sc1\n\t"
                         "global_load_dwordx4 %4, %7, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %7, off offset:80 sc1\n\t"
                         "global_load_dwordx4 %6, %7, off offset:96 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]) : "v"(sp) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = ((const v4f *)sp)[i];
        }
    };
#pragma unroll
    for (int c = 0; c < NC; c++) {
        fetch(c, buf);
        v4f *d4 = (v4f *)((char *)dst + 112 * c);
#pragma unroll
        for (int i = 0; i < 7; i++) d4[i] = buf[i];
    }
#else
    for (int q = 0; q <= NNB; q++)
        for (int k = 0; k < OBSW; k++) dst[q * OBSW + k] = srcs[q] ? srcs[q][k] : (ROW)0;
#endif
}
// sorted position -> entry: the low half of a tagged word (Shared<128>, see ph_rank) or a plain 16-bit index (SharedGeo)
template <class W> PVE_HD int sidx_at(const W *s, int i)
{
    if constexpr (sizeof(W) == 4) return ((const uint16_t *)s)[2 * i];
    else return s[i];
}
PVE_HD void lds_or(int *p, int v)
{
#if PVE_DEVICE_CODE
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
PVE_HD void lds_and(int *p, int v)
{
#if PVE_DEVICE_CODE
    atomicAnd(p, v);
#else
    *p &= v;
#endif
}
// inclusive prefix sum over the lanes of one wave (emulator: lanes run in increasing order -> running sum)
PVE_HD int wave_incl_scan(int t, int x, int *emu_acc)
{
#if PVE_DEVICE_CODE
    // DPP row shifts + row broadcasts (VALU-rate data movement) instead of 6 dependent ds_bpermute round trips.  Row shifts:
    // bound_ctrl makes the lanes without a source read 0 (no `old` operand to initialise); the row broadcasts only write
    // the rows of their mask, the others keep old = 0
    x += __builtin_amdgcn_mov_dpp(x, 0x111, 0xf, 0xf, true);             // row_shr:1
    x += __builtin_amdgcn_mov_dpp(x, 0x112, 0xf, 0xf, true);             // row_shr:2
    x += __builtin_amdgcn_mov_dpp(x, 0x114, 0xf, 0xf, true);             // row_shr:4
    x += __builtin_amdgcn_mov_dpp(x, 0x118, 0xf, 0xf, true);             // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return x;
#else
    *emu_acc += x;
    return *emu_acc;
#endif
}
// deterministic block sum: wave-level tree, one partial per wave (red[] summed by thread 0 later).
// Device: DPP row shifts / row broadcasts (no LDS crossbar traffic); the total lands in lane 63.
#if PVE_DEVICE_CODE
template <int CTRL, int ROW_MASK> PVE_HD double dpp_add_f64(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    int slo, shi;
    if (ROW_MASK == 0xf) {               // row shifts: bound_ctrl, lanes without a source read +0.0 (adding it is exact)
        slo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
        shi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    } else {                             // row broadcasts: the rows outside the mask keep old = +0.0
        slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
        shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    }
    return x + __hiloint2double(shi, slo);
}
#endif
PVE_HD void block_sum(double *red, int t, double x)
{
#if PVE_DEVICE_CODE
    x = dpp_add_f64<0x111, 0xf>(x);      // row_shr:1
    x = dpp_add_f64<0x112, 0xf>(x);      // row_shr:2
    x = dpp_add_f64<0x114, 0xf>(x);      // row_shr:4
    x = dpp_add_f64<0x118, 0xf>(x);      // row_shr:8   -> lane 15 of every row holds the row sum
    x = dpp_add_f64<0x142, 0xa>(x);      // row_bcast:15 into rows 1 and 3
    x = dpp_add_f64<0x143, 0xc>(x);      // row_bcast:31 into rows 2 and 3 -> lane 63 = wave sum
    if ((t & 63) == 63) red[t >> 6] = x;
#else
    red[0] += x;
#endif
}

// block_sum of a term that is zero in most waves of most ticks (jerk_sum of the vehicles that finish): a wave without a
// single term stores +0.0, which is what the tree returns for 64 zeros
PVE_HD void block_sum_sparse(double *red, int t, bool has, double x)
{
#if PVE_DEVICE_CODE
    if (__builtin_amdgcn_ballot_w64(has) == 0) {
        if ((t & 63) == 63) red[t >> 6] = 0.0;
        return;
    }
#endif
    block_sum(red, t, has ? x : 0.0);
}

// min / max of two finite doubles as ONE instruction each (v_min_f64 / v_max_f64; the host build: fmin / fmax).  Equal to
// the compare-and-select forms they replace for everything but NaN operands (never produced here) and the sign of a zero
// result when both operands are zeros of different sign (the bounds are am / aM / vm / vM / +-20; -0.0 is never produced)
#ifdef PVE_SELECT_CLAMPS                           // A/B build knob: the compare-and-select forms
PVE_HD double dmax(double a, double b) { return (a > b) ? a : b; }
PVE_HD double dmin(double a, double b) { return (a < b) ? a : b; }
#else
PVE_HD double dmax(double a, double b) { return __builtin_fmax(a, b); }
PVE_HD double dmin(double a, double b) { return __builtin_fmin(a, b); }
#endif

// Element `idx` of intersection `env`'s block of PER elements in a per-tick output array: a UNIFORM 64-bit base (scalar
// arithmetic) plus a 32-bit lane offset, which is the address form a global store takes as it is (`saddr` + `voffset`); the
// plain `base[(size_t)env * PER + idx]` is a 64-bit multiply-add per lane and store (quarter-rate v_mad_u64_u32 / two-pass
// v_lshl_add_u64: 24 of them in FIN).  idx is a slot or a slot-derived index (< 2^22).
#ifdef PVE_WIDE_INDEX                              // A/B build knob: the plain 64-bit form
template <int PER, class T> PVE_HD T *env_at(T *base, int env, int idx) { return base + ((size_t)env * PER + (size_t)idx); }
#else
template <int PER, class T> PVE_HD T *env_at(T *base, int env, int idx)
{   // (the byte offset is formed in 32 bits: idx < 2^22 elements of <= 8 bytes)
    return (T *)((char *)(base + (size_t)env * PER) + (unsigned)(idx * (int)sizeof(T)));
}
#endif

// products of small non-negative integers (counts, ranks, slots: < 2^23) as the full-rate 24-bit multiply (v_mul_i32_i24 /
// v_mad_i32_i24) instead of the quarter-rate 32-bit one the compiler must pick when it cannot see the range
PVE_HD int mul24(int a, int b)
{
#if PVE_DEVICE_CODE && !defined(PVE_NO_RANGE_ASSUME)
    return __mul24(a, b);
#else
    return a * b;
#endif
}
PVE_HD int mad24(int a, int b, int c)
{
#if PVE_DEVICE_CODE && !defined(PVE_NO_RANGE_ASSUME)
    return __mul24(a, b) + c;                      // (the compiler fuses the add: v_mad_i32_i24)
#else
    return a * b + c;
#endif
}

// ------------------------------------------------------------------ shared (LDS) block of one env
// HOME (k_rollout<128, 5, ..>: the 96-register / 10-workgroups-per-CU build of the resident kernel): the per-slot state lives
// in LDS, registers hold it only inside the phase that works on it.  The fields that are merely CARRIED through most of the
// tick -- jerk_sum, closer_p, id, seq_in_lane | id_info[1], count -- have arrays of their own: FX updates count / jerk_sum in
// place, the dense thread files closer_p where it belongs, the table source reads the id where it needs it.  vir_dis stays in
// virdis[], p / v / a in p[] / v[] / a[] (S1 reads them there, and the next action in act_next[], instead of RELOAD handing
// them over in registers).  When vehicles move, a thread takes its values at the very end of FIN (the registers of the
// observation row are free again) and puts them at the new slot behind barrier A.  The entry pool shrinks to 304 entries to pay
// for the homes AND to bring the block down to 15 264 B = 12 LDS granules of 1 280 B: ten workgroups per CU are resident only
// at <= 15 360 B (tools/occupancy_probe.hip; the runtime's occupancy query says ten up to 16 384 B, the hardware starts nine).
// BUILD .. WALK run in passes over groups of lists when a tick needs more entries than the pool holds.
template <int CAP> struct Homes {
    double h_jerk_sum[CAP], h_closer[CAP];
    int h_id[CAP], h_sv[CAP], h_count[CAP];          // h_sv = seq_in_lane << 8 | id_info[1] (id_info[1] < CAP <= 128, rows < 2^23)
};
template <int CAP> struct HomesOff {};              // (empty base: the other blocks keep their size to the byte)
struct HomeRegs { double jerk_sum, closer, vir_dis; int id, sv, count; };
// POOL_: entries of the list pool (HOME: >= 2 CAP + 36 so that the staging overlays fit; the CPU test emulator instantiates a
// smaller pool than the kernel's 304 entries to drive ordinary traffic through the multi-pass form)
template <int CAP, bool LJ = (CAP == 128), bool HOME_ = false, int POOL_ = (HOME_ ? 304 * CAP / 128 : 5 * CAP)>
struct Shared : std::conditional<HOME_, Homes<CAP>, HomesOff<CAP>>::type {
    static constexpr int NW = CAP / 64;
    static constexpr bool HOME = HOME_;
    static_assert(!HOME_ || (POOL_ * 4 >= 2 * CAP * 4 + CAP + 1 && POOL_ >= CAP && POOL_ % 8 == 0), "HOME: the staging overlays need 2 CAP ints + CAP + 1 bytes of s_idx");
    static constexpr bool PIN_READS = true;   // walk_window: keep the batched window reads from being sunk into guarded blocks
    static constexpr int POOL = POOL_;
    // virtual-lane lists (ref :238-273): list d = own controlled vehicles + those of the <=4 conflict
    // lanes; u_* = entries in segment order, s_* = sorted by (vd, slot). 5*CAP bounds the total.
    // CAP = 128: the entries sorted by (vd, slot) are an index array into u_vd / u_slot (no second copy of the
    // distances: 4.4 KB less LDS = 10 instead of 8 workgroups per CU).  CAP = 64: LDS does not limit residency there
    // (16 one-wave workgroups per CU either way), so the sorted copy is kept and WALK reads it without the index hop.
    static constexpr bool DIRECT = (CAP == 64);
    static constexpr bool DENSE = true;       // controlled-vehicle phases on the dense mapping (see the top of this file)
    EnvHeader hd;
    // post-step kinematics of every slot; cell CAP of v / a / lane_of / lj is a vehicle that is not there (zeros, written
    // once per launch): FIN gathers the absent neighbours of an observation row from it instead of selecting 24 zeros
    double p[CAP], v[CAP + 1], a[CAP + 1];
    union {                          // p1/v1 die at the barrier after S3, the lists are born after it
        struct { double p1[CAP], v1[CAP]; };   // step outcome "if braking" (S1-S3 only)
        double u_vd[POOL];           // virtual distance of every list entry (segment order); RANK's tail round may read up to 6
    };
    double virdis[CAP];              // doubles past the pool, i.e. into this array (masked): it must stay the next member
    int cnt[CAP];                    // collision hits received: early | late << 16
    // sorted position -> entry | tick tag << 16: RANK claims a position with one exchange and learns from the tag it gets
    // back whether an entry with the SAME distance was there first (ph_rank); readers take the low half
    alignas(8) unsigned s_idx[DIRECT ? 1 : POOL];
    double s_vd[DIRECT ? POOL : 1];
    alignas(8) uint8_t u_slot[POOL];
    // k_rollout keeps the state on the chip between two ticks: every persistent field of every vehicle moves to its
    // new slot through LDS.  The staging arrays live in storage that is dead by then:
    //   EARLY (written at the top of FIN, the tick's work arrays still being read): jerk, jerk_sum, vir_dis, closer_p in
    //         u_vd[CAP .. 5 CAP) (the dead-lock records only use u_vd[0 .. CAP)); the 7 ints in s_idx / p / cnt / u_slot /
    //         u_list[CAP ..) (CAP = 128; the dead-lock scratch lk_slot only uses the first CAP bytes) or s_vd (CAP = 64) -- all dead after WALK / REWARD / FX;
    //   LATE  (after the barrier behind FIN): p in u_vd[0 .. CAP), v and a in place.
    // so that FIN holds no more registers than in the single-tick kernel while it builds the observation rows.
    enum { SF_JERK = 0, SF_JERK_SUM, SF_VIR_DIS, SF_CLOSER_P, SF_P, SF_V, SF_A };
    template <int K> PVE_HD double *stf()
    {
        if (HOME) return K == SF_JERK ? u_vd + CAP : (K == SF_P ? p : (K == SF_V ? v : a));   // (p in place too; the rest has homes)
        return K < 4 ? u_vd + (K + 1) * CAP : (K == SF_P ? u_vd : (K == SF_V ? v : a));
    }
    // k_rollout, LOCK2 -> FIN: kept vehicles below every slot (index CAP: all of them), so that FIN's new lane starts and the
    // dense threads' new slots are one read each instead of a two-word popcount; lives behind the 2 CAP ints of the EARLY
    // staging in the sorted-list storage (dead since WALK)
    PVE_HD uint8_t *keep_pre() { return DIRECT ? (uint8_t *)s_vd + 7 * CAP * 4 : (uint8_t *)s_idx + 2 * CAP * 4; }
    // FX -> LOCK: what FX decided about the reward of the vehicle in slot t (0 = keep, 1 = -10, 2 = +5), for the dense
    // thread that holds the reward; u_list[CAP ..) is free between RANK and the EARLY staging of FIN
    PVE_HD uint8_t *fxcode() { return u_list + CAP; }
    // FX -> LOCK: the virtual-header pointers as a byte chain for the dead-lock walk: chain[s] = header of slot s, CAP = none;
    // chain[CAP] = CAP (sentinel).  u_slot is free between REWARD and FIN's staging
    PVE_HD uint8_t *chain() { return u_slot; }
    template <int K> PVE_HD int *sti()   // K = I_ID .. I_HDR
    {
        if (DIRECT) return (int *)s_vd + K * CAP;
        if (HOME) return K == I_STEP ? (int *)s_idx : (K == I_META ? (int *)s_idx + CAP : cnt);   // (step, meta; the header word)
        return K < 2 ? (int *)s_idx + K * CAP : (K < 4 ? (int *)p + (K - 2) * CAP : (K == 4 ? cnt : (K == 5 ? (int *)u_slot : (int *)(u_list + CAP))));
    }
    int acc_passed_steps, acc_collisions;
    // Arrays with disjoint lifetimes share storage: the CAP = 64 block must stay <= 10 KB so that 16 workgroups
    // (all 4096 envs of BASELINE config 2) are resident per CU; LDS is allocated in 1 KB granules.
    union {
        int16_t pref[NL * 5];        // inclusive prefix of the 60 segment sizes (S2 .. S3 only)
        int16_t hdr[CAP];            // slot of the virtual header (predecessor) or -1 (initialised in BUILD)
    };
    union {
        uint8_t bb[CAP];             // brake bits: bit0 if front did not brake, bit1 if it did (S2 .. S3 only)
        uint8_t rew_ovr[CAP];        // reward[-1] override (FX .. LOCK; zeroed in BUILD)
    };
    u64 m_alive[NW], m_ctl[NW];
    union {
        // entries of list d with a finite distance (own lane + the chosen conflict entries, ref :259-270): RANK sorts only
        // those, WALK never sees an unchosen entry (LISTS .. WALK; the masks below are born in FX)
        int nfin[NL];
        struct {
            u64 m_del[NW], m_fin[NW], m_ctlnow[NW], m_coll[NW];
            union { u64 m_lead[NW]; u64 m_keep[NW]; };   // tick kernel | compaction kernel
            u64 m_spawn[NW];
        };
    };
    uint8_t s_slot[DIRECT ? POOL : 1];
    union {
        alignas(8) uint8_t u_list[POOL];   // list of every entry (BUILD .. RANK)
        uint8_t lk_slot[POOL];       // dead-lock scratch: slot of the record filed at each rank (LOCK2 .. FIN)
    };
    union {                          // (both < CAP: a list holds a vehicle at most once, the cycles claim <= one unit per member)
        uint8_t mypos[CAP];          // position of each controlled vehicle inside its own lane's list (RANK .. WALK)
        uint8_t cyc_off[CAP];        // scratch offset of the dead-lock cycle led by slot t (LOCK .. FIN)
    };
    uint8_t slot_of[CAP];            // dense mapping: slot of the c-th controlled vehicle (S2 .. FIN)
    int16_t cstart[NL + 1];          // controlled vehicles in the lanes below d = dense index of lane d's first one
                                     // (cstart[d + 1] - cstart[d] = controlled vehicles of lane d = the own segment of list d)
#if !PVE_DEVICE_CODE
    int emu_scan;                    // emulator-only accumulator of wave_incl_scan
#endif
    int16_t loff[NL + 1];            // list d occupies [loff[d], loff[d+1])
    union {
        alignas(8) int16_t segoff[NL][5];                  // start of segment (own, conflict 0..3) inside list d (LISTS .. BUILD)
        struct { double red_reward[NW], red_jerk[NW]; };   // per-wave partial sums (LOCK .. FIN)
    };
    uint8_t lane_of[CAP + 1];        // lane of every alive slot
    static constexpr bool HAS_LJ = LJ;                // (CAP = 64: the block must stay <= 10 KB; k_tick<128>: <= 16 KB, 10 workgroups per CU)
    int lj[HAS_LJ ? CAP + 1 : 1];    // lane << 16 | j of every alive slot (S1 .. FIN): the `(lane, j)` names of neighbours and virtual
                                     // headers are one gather instead of lane_of + lane_start + arithmetic
    union {
        float xy32[CAP][2];          // single-precision position of every controlled vehicle (collision pre-filter; BUILD .. REWARD)
        double act_next[CAP];        // k_rollout: the NEXT tick's action of every slot, prefetched under the tail of this tick
    };
    double tabA[2][4], tabB[2][4], tabC[2][4];   // get_virtual_distance table (copy of Const, lane-indexed reads; BUILD only)
    alignas(4) uint8_t l2lp[NL][4];  // lane2lane[d][k] (15 = none) | our position inside lane2lane[that lane] << 4 | (that lane % 3) << 6
                                     // (rows are read as one dword; the high nibble IS the index m * 4 + position into tabA / B / C)
    int lead_n;                      // scratch units claimed by the dead-lock cycles
};

struct Regs {
    double p, v, a, jerk, jerk_sum, vir_dis, closer_p;
    double a0, a1;
    double reward;
    double kv[NNB];
    int kr[NNB];
    int id, seq, vnum, step, count, meta;
    int lane, j;
    int hdr;
    int hit, coll_seen, coll_fin;
    int alive, ctl, del, fin;
    int cyc;                         // dead-lock cycle membership: bit0 | len << 1 | rank << 5 | leader slot << 9
    int intent, route, ord;          // general-geometry path only (intention, direction[lane][intention], processing order)
    int mmask;                       // general-geometry path only: bit d = member of list d (COUNT .. FILL)
    // dense mapping (12-lane kernels): thread t = the t-th controlled vehicle of the intersection.  reward / kr / kv / hdr /
    // cyc above belong to THAT vehicle there (the general-geometry kernel keeps them per slot)
    int dctl, ds, dlane;             // t < number of controlled vehicles; its slot and lane
    double dp, djerk, dcloser;       // its position, this tick's jerk (handed over through virdis), closer_p (ref :302)
    double act;                      // this tick's action of the slot (loaded with the state, used by S1)
    double next_arr;                 // lane t < 12 that spawns: its next arrival time (loaded in LOCK, stored in FIN)
    double act_nx;                   // k_rollout: next tick's action of this slot (global load in flight under FX .. LOCK2)
};
// k_rollout: what FIN hands over to the staging step behind the next barrier (values read from the header / the work
// arrays before anybody rewrites them)
struct FinCarry {
    int still, meta;                 // nobody moves this tick (uniform): registers carry over, `meta` = the slot's new flags word
    int new_slot;                    // of the vehicle in slot t (< 0: deleted or empty)
    int ls;                          // t <= NL: lane_start[t] after re-pack + spawn
    int sp_slot, sp_id, sp_vnum;     // t < NL: slot / id / id_info[1] of the vehicle lane t spawns (sp_slot < 0: none)
    int sp_int;                      // general-geometry kernel: its intention (ref :382-394)
    // header: thread 0 adds the tick's counters and sums to the resident header IN FIN (nobody else reads those fields); only
    // what other threads still read during FIN changes behind barrier A (ph_stage_header): n_alive and id_seq.  As members of
    // this struct the nine counters / sums were live vector registers in EVERY lane across FIN's peak (uniform values, but
    // LDS-derived, so the compiler keeps them per lane): the training and general-geometry variants spilled 6-110 registers.
    int n_post, n_sp;
    int n_ctl;                       // controlled vehicles of this tick (the actor pass of k_rollout<.., ACT>)
};
struct CRegs {                       // MODE_COMPACT moves every persistent field verbatim
    double p, v, a, jerk, jerk_sum, vir_dis, closer_p;
    int id, seq, vnum, step, count, meta, hdr_word, alive;
    double obsrow[OBSW];
};

// ------------------------------------------------------------------ geometry: ref :1250-1289
// sin/cos on [0, pi/2] (the arc angle r_a of a vehicle inside the box) by Taylor polynomials about 0
// after folding to [0, pi/4]: |error| < 2e-16.  They only feed the XY collision distance, whose decision
// margin in the golden tapes is >= 4e-4, never an exact-IEEE decision.
PVE_HD void sincos_q1(double x, double &sn, double &cs)
{
    const double hp = 1.5707963267948966;            // pi/2
    const bool fold = x > 0.7853981633974483;        // > pi/4: sin(x) = cos(pi/2 - x), cos(x) = sin(pi/2 - x)
    const double y = fold ? (hp - x) : x;
    const double z = y * y;
    double ps = -7.6471637318198164759e-13;          // -1/15!
    ps = ps * z + 1.6059043836821614599e-10;         //  1/13!
    ps = ps * z - 2.5052108385441718775e-08;         // -1/11!
    ps = ps * z + 2.7557319223985890653e-06;         //  1/9!
    ps = ps * z - 1.9841269841269841270e-04;         // -1/7!
    ps = ps * z + 8.3333333333333333333e-03;         //  1/5!
    ps = ps * z - 1.6666666666666666667e-01;         // -1/3!
    const double s1 = y + y * z * ps;
    double pc = 4.7794773323873852974e-14;           //  1/16!
    pc = pc * z - 1.1470745597729724714e-11;         // -1/14!
    pc = pc * z + 2.0876756987868098979e-09;         //  1/12!
    pc = pc * z - 2.7557319223985890653e-07;         // -1/10!
    pc = pc * z + 2.4801587301587301587e-05;         //  1/8!
    pc = pc * z - 1.3888888888888888889e-03;         // -1/6!
    pc = pc * z + 4.1666666666666666667e-02;         //  1/4!
    const double c1 = 1.0 - 0.5 * z + z * z * pc;
    sn = fold ? c1 : s1;
    cs = fold ? s1 : c1;
}

// PVE_PIN(x): the value of x is materialised HERE.  Without it the compiler sinks a speculative LDS read into the
// conditional block that uses its result, which turns a batch of independent reads back into one guarded basic block
// (read, wait, read, wait) per element.
#if PVE_DEVICE_CODE
#define PVE_PIN(x) asm volatile("" : "+v"(x))
#else
#define PVE_PIN(x) ((void)0)
#endif

// Small constant tables of the kernel arguments are never indexed with a per-lane value (that would be a vector load
// from the argument buffer, ~1 us on the critical path): the entries are scalar-loaded and selected.
// (the empty asm makes the loaded values opaque: otherwise the compiler turns the select of loads back into one load
// from a selected address)
PVE_HD double sel2(double a0, double a1, bool second)
{
#if PVE_DEVICE_CODE
    asm volatile("" : "+s"(a0), "+s"(a1));
#endif
    return second ? a1 : a0;
}
PVE_HD double sel3(const PVE_AS4 double (&a)[3], int i)
{
    double a0 = a[0], a1 = a[1], a2 = a[2];
#if PVE_DEVICE_CODE
    asm volatile("" : "+s"(a0), "+s"(a1), "+s"(a2));
#endif
    return i == 0 ? a0 : (i == 1 ? a1 : a2);
}
PVE_HD double sel4(const PVE_AS4 double (&a)[4], int i)
{
    double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
#if PVE_DEVICE_CODE
    asm volatile("" : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3));
#endif
    return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
}

PVE_HD void get_xy(const PVE_AS4 Const &c, double p, int lane, double &X, double &Y)
{   // straight-line (select-based) so that the two evaluations per vehicle pair overlap in the pipeline
    const double cw = c.cw;
    const int m = lane % 3;
    const double Lb = sel2(c.inbox[0], c.inbox[2], m == 2);
    const bool before = p > Lb, inside = !before && p > 0;
    const bool arc = (m != 1) && inside;
    double sn, cs;
    sincos_q1(arc ? ((Lb - p) / Lb * c.arc_k / 2) : 0.0, sn, cs);   // r_a, ref :1259, :1277
    // left turn (ref :1253-1267): approach y = cw, arc about (6cw, -6cw), exit x = -cw
    // right turn (ref :1271-1286): approach y = 5cw, arc about (6cw, 6cw) clockwise, exit x = 5cw
    const double p0y = (m == 0) ? cw : 5 * cw, pry = (m == 0) ? -6 * cw : 6 * cw;
    const double p0x = 6 * cw, prx = 6 * cw;
    const double ax = (m == 0) ? (prx + (p0x - prx) * cs - (p0y - pry) * sn) : (prx + (p0x - prx) * cs + (p0y - pry) * sn);
    const double ay = (m == 0) ? (pry + (p0y - pry) * cs + (p0x - prx) * sn) : (pry + (p0y - pry) * cs - (p0x - prx) * sn);
    const double bx = p - Lb + 6 * cw, by = p0y;                       // before the box
    const double ex = (m == 0) ? -cw : 5 * cw;                          // after the box
    const double ey = (m == 0) ? (-6 * cw + p) : (6 * cw - p);
    double x = before ? bx : (inside ? ax : ex);
    double y = before ? by : (inside ? ay : ey);
    if (m == 1) { x = p - 6 * cw; y = 3 * cw; }                         // straight (ref :1268-1270)
    const double rc = sel4(c.rot_cos, lane / 3), rs = sel4(c.rot_sin, lane / 3);
    X = x * rc - y * rs;
    Y = y * rc + x * rs;
}

// single-precision position for the collision PRE-FILTER only (never for a decision): |error| < 1e-3 m
// 1 / x to ~1 ulp in ONE instruction (v_rcp_f32) for the single-precision pre-filters, which never decide anything (an IEEE
// float division is ten instructions)
PVE_HD float frcp(float x)
{
#if PVE_DEVICE_CODE
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
PVE_HD void get_xy_f32(const PVE_AS4 Const &c, double pd, int lane, float &X, float &Y)
{
    const float cw = (float)c.cw, p = (float)pd;
    const int m = lane % 3;
    const float Lb = (float)sel2(c.inbox[0], c.inbox[2], m == 2);
    const bool before = p > Lb, inside = !before && p > 0.f;
    float ra = (m != 1 && inside) ? ((Lb - p) * (1.5707965f * frcp(Lb))) : 0.f;     // (Lb - p) / Lb * 3.141593 / 2 (pre-filter: no exact division)
    const bool fold = ra > 0.78539816f;
    const float y0 = fold ? (1.5707964f - ra) : ra, z = y0 * y0;
    const float s1 = y0 + y0 * z * (-1.6666667e-1f + z * (8.3333338e-3f + z * (-1.9841270e-4f + z * 2.7557319e-6f)));
    const float c1 = 1.f + z * (-0.5f + z * (4.1666668e-2f + z * (-1.3888889e-3f + z * 2.4801587e-5f)));
    const float sn = fold ? c1 : s1, cs = fold ? s1 : c1;
    const float p0y = (m == 0) ? cw : 5.f * cw, pry = (m == 0) ? -6.f * cw : 6.f * cw, prx = 6.f * cw;
    const float ax = (m == 0) ? (prx - (p0y - pry) * sn) : (prx + (p0y - pry) * sn);
    const float ay = pry + (p0y - pry) * cs;
    float x = before ? (p - Lb + 6.f * cw) : (inside ? ax : ((m == 0) ? -cw : 5.f * cw));
    float y = before ? p0y : (inside ? ay : ((m == 0) ? (-6.f * cw + p) : (6.f * cw - p)));
    if (m == 1) { x = p - 6.f * cw; y = 3.f * cw; }
    // the approach's rotation as an EXACT quarter turn (negations and a swap): the reference rotates by 3.141593 / 2 per approach
    // (ref :1251), 1.7e-7 rad per step away from pi / 2, i.e. < 1e-4 m at 170 m -- inside this pre-filter's 5 cm margin; the
    // two table selects + conversions + four products of the general form were ~35 vector instructions per vehicle
    const int q = lane / 3;
    const bool odd = (q & 1) != 0;
    const float xr = odd ? -y : x, yr = odd ? x : y;          // q = 1: (-y, x)
    X = (q & 2) ? -xr : xr;                                   // q = 2: (-x, -y), q = 3: (y, -x)
    Y = (q & 2) ? -yr : yr;
}

// ------------------------------------------------------------------ reward terms: ref :311-320
// The reward is a float output (tolerance 1e-5, asserted at 1e-9), never an input of a discrete decision, so its two
// transcendental terms use short forms instead of the general-purpose library routines (tanh: 165 VALU instructions,
// log: 98): both are accurate to ~1e-15 on the ranges the reward can produce before it is clamped to [-20, 20].
// 1 / tanh(-t/4) for 0 < t < 4 (ref :314-315): coth(x) = (e^2x + 1) / (e^2x - 1), x = -t/4 in (-1, 0).  The
// cancellation in e^2x - 1 only bites for t < 1e-3, where the term is < -4000 and the clamp takes over.
// e^x for x in [-2, 0]: x = k ln2 + r, k = rint(x log2(e)) in {-3 .. 0}, |r| <= 0.35, Taylor polynomial of degree 12
// (0.35^13 / 13! = 2e-16) and one ldexp: 18 instructions against ~45 of the general routine; ~1 ulp.
PVE_HD double exp_m2_0(double x)
{
    const double k = rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);      // ln2 hi
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);             // ln2 lo
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
// x / y for the VALUE-only quotients of the reward terms (never an input of a decision; asserted at 1e-9, bar 1e-5):
// v_rcp_f64 + two Newton steps (~1 ulp) in 6 instructions instead of the 13 of the IEEE division sequence
PVE_HD double value_div(double x, double y)
{
#if PVE_DEVICE_CODE
    double r = __builtin_amdgcn_rcp(y);
    r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
    return x * r;
#else
    return x / y;
#endif
}
PVE_HD double reward_coth_term(double t_distance)
{
    const double u = exp_m2_0(-0.5 * t_distance);       // e^(2x); -t/4 * 2 is exact
    const double den = u - 1.0;                         // (< 0; exactly 0 only for t < 2.2e-16, where 1 / tanh(-t/4) = -4 / t is
    return den == 0.0 ? -INFINITY : value_div(u + 1.0, den);   //  far beyond the clamp: -inf, never the NaN of 2 * rcp(0) refined)
}
// log(z) for z in [1e-5, 1.00001] (ref :317-318: z = (d/10)^5 + 1e-5, d < 10): z = 2^e * m, m in [sqrt(1/2), sqrt(2)),
// log(m) = 2s(1 + s^2/3 + s^4/5 + ...), s = (m - 1)/(m + 1), |s| <= 0.172: 10 terms give 1e-16.
PVE_HD double reward_log_term(double z)
{
    int e;
    double m = frexp(z, &e);                            // m in [0.5, 1)
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = value_div(f, 2.0 + f);
    const double w = s * s;
    double q = 1.0 / 19.0;
    q = __builtin_fma(q, w, 1.0 / 17.0);
    q = __builtin_fma(q, w, 1.0 / 15.0);
    q = __builtin_fma(q, w, 1.0 / 13.0);
    q = __builtin_fma(q, w, 1.0 / 11.0);
    q = __builtin_fma(q, w, 1.0 / 9.0);
    q = __builtin_fma(q, w, 1.0 / 7.0);
    q = __builtin_fma(q, w, 1.0 / 5.0);
    q = __builtin_fma(q, w, 1.0 / 3.0);
    q = __builtin_fma(q, w, 1.0);
    return __builtin_fma((double)e, 0.69314718055994530942, (s + s) * q);
}

PVE_HD bool key_less(double d1, double v1, int r1, double d2, double v2, int r2)
{   // total order of the reference's stable sort on |vd - vd_self| over a vd-sorted list (ref :271, :1389)
    return d1 < d2 || (d1 == d2 && (v1 < v2 || (v1 == v2 && r1 < r2)));
}

// x / b for a constant b with rb = RN(1 / b): q = RN(x rb), r = x - q b (exact in one FMA), q' = RN(q + r rb) is the
// correctly rounded quotient (Markstein 1990), i.e. bit for bit what the division instruction sequence returns, in 3
// instructions instead of ~13 incl. a quarter-rate reciprocal.  Checked against IEEE division on 4.9e9 operands (random
// over 128 binades + every neighbour of the rounding midpoints) for b = 3, 6 and nine other divisors: identical except for
// the sign of a zero result and results in the subnormal range, neither of which can reach a decision here (the terms are
// 0 or >= 1e-14 in magnitude, and d_safe is only compared); pve_create refuses |am| outside [1e-6, 1e6].
PVE_HD double div_const(double x, double b, double rb)
{
    const double q = x * rb;
    const double r = __builtin_fma(-q, b, x);
    return __builtin_fma(r, rb, q);
}
PVE_HD int brake_needed(const PVE_AS4 Const &c, double p, double v, double fp, double fv)
{   // ref :1509-1516 (front = vehicle j-1 AFTER its own update)
    // straight-line (no branch) so that both outcomes overlap
    const double d_safe = v * 0.4 + div_const(v * v - fv * fv, c.two_abs_am, c.inv_two_abs_am) -
                          div_const((v - fv) * c.vm, c.abs_am, c.inv_abs_am);
    return ((fv < v) & (p - fp < d_safe)) ? 1 : 0;
}

PVE_HD int slot_lane(const EnvHeader &hd, int t)
{
    int lane = 0;
#pragma unroll
    for (int L = 1; L < NL; L++) lane += (t >= hd.lane_start[L]) ? 1 : 0;
    return lane;
}

// ShT: the LDS block the phases work on (Shared<CAP> for the 12-lane fast path; the general-geometry path of
// pve_tick_geo.h re-uses the step / dead-lock / compaction phases on its own block with the same member names)
template <int CAP, class ShT = Shared<CAP>> struct Tick {
    typedef ShT Sh;
    static constexpr int NW = CAP / 64;

    // ============================================================== L: load
    // COH: the state may have been stored by another workgroup of this launch (persistent roll-out): coherent loads.
    // act0: the first tick's actions of this roll-out item ([n_envs][CAP], or null), instead of P.actions.
    // COHA: the actions too (the persistent closed loop hands them from one item to the next; a pool is read-only input)
    template <bool COH = false, bool COHA = false>
    static PVE_HD void ph_load(const PVE_AS4 Const &c, const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r,
                               const double *act0 = nullptr, bool use_act0 = false)
    {
        const EnvHeader &gh = P.headers[env];
        {
            const int *src = (const int *)&gh;
            int *dst = (int *)&sh.hd;
            for (int w = t + 2; w < (int)(sizeof(EnvHeader) / 4); w += CAP) dst[w] = gld<COH>(src + w);   // words 0-1 = clock
        }
        if (t == 0) sh.hd.current_time = gld<COH>(&gh.current_time) + c.deltaT;      // ref :223 (repeated +=, not tick*dt)
        const int N = gld<COH>(&gh.n_alive);
        const size_t g = (size_t)env * CAP + t;        // (LOAD / FLUSH keep the plain 64-bit index: once per item, and with env_at
                                                       //  the persistent kernel's allocation tips over into 8-10 spilled registers)
        r.alive = t < N;
        r.jerk = 0;
        r.p = r.v = r.a = r.jerk_sum = r.vir_dis = r.closer_p = 0;
        r.id = r.seq = r.vnum = r.step = r.count = r.meta = 0;
        // First wave: unconditional (slots >= N hold zeros / stale values that are never used), so that the loads do
        // not wait for n_alive: one memory latency instead of two on the critical path.  Later waves (slots >= 64) are
        // off the critical path and mostly empty: they wait for n_alive and read the live slots only.
        r.act = 0;
        const double *acts = use_act0 ? act0 : P.actions;
        if (t < 64 || t < N) {
            // what S1 - S3 need comes first; the fields that are merely carried to FIN (or first read in WALK) are requested
            // behind them and arrive under the step phases (the kernel's barriers do not wait for global loads)
            if (acts) r.act = gld<COHA>(acts + g);        // with the state loads: one latency, not a second one in S1
            r.p = gld<COH>(P.f64[F_P] + g); r.v = gld<COH>(P.f64[F_V] + g); r.a = gld<COH>(P.f64[F_A] + g);
            r.meta = gld<COH>(P.i32[I_META] + g); r.step = gld<COH>(P.i32[I_STEP] + g);
            r.jerk_sum = gld<COH>(P.f64[F_JERK_SUM] + g); r.vir_dis = gld<COH>(P.f64[F_VIR_DIS] + g);
            r.closer_p = gld<COH>(P.f64[F_CLOSER_P] + g);
            r.id = gld<COH>(P.i32[I_ID] + g); r.seq = gld<COH>(P.i32[I_SEQ] + g); r.vnum = gld<COH>(P.i32[I_VNUM] + g);
            r.count = gld<COH>(P.i32[I_COUNT] + g);
        }
        sh.cnt[t] = 0;                                    // (rew_ovr / hdr share storage with S2-S3 arrays: BUILD)
        if (t == 0) {
            sh.acc_passed_steps = 0; sh.acc_collisions = 0; sh.lead_n = 0;
            sh.v[CAP] = 0; sh.a[CAP] = 0; sh.lane_of[CAP] = 0;
            if constexpr (Sh::HAS_LJ) sh.lj[CAP] = 0;
        }
        if (t < 8) {
            sh.tabA[t >> 2][t & 3] = c.vdA[t >> 2][t & 3];
            sh.tabB[t >> 2][t & 3] = c.vdB[t >> 2][t & 3];
            sh.tabC[t >> 2][t & 3] = c.vdC[t >> 2][t & 3];
        }
        if (t < NL * 4) {
            const int L = c.l2l[t >> 2][t & 3];
            sh.l2lp[t >> 2][t & 3] = (uint8_t)((L < 0 ? 15 : L) | ((c.l2l_inv[t >> 2][t & 3] & 3) << 4) | ((L < 0 ? 0 : (L % 3) & 1) << 6));
        }
    }

    // HOME: the carried fields LOAD has fetched go to their LDS homes (once per launch / queue item, in front of the tick loop)
    static PVE_HD void ph_home_store(int t, Sh &sh, Regs &r)
    {
        if constexpr (Sh::HOME) {
            sh.h_jerk_sum[t] = r.jerk_sum; sh.h_closer[t] = r.closer_p;
            sh.h_id[t] = r.id; sh.h_sv[t] = (r.seq << 8) | (r.vnum & 0xFF); sh.h_count[t] = r.count;
            sh.p[t] = r.p; sh.v[t] = r.v; sh.a[t] = r.a; sh.virdis[t] = r.vir_dis; sh.act_next[t] = r.act;
        }
    }

    // ============================================================== S1: step, both outcomes
    static PVE_HD void outcome(const PVE_AS4 Const &c, double p, double v, double a, bool ctl, double &pn, double &vn)
    {   // ref :1528-1535
        pn = p - v * c.deltaT - 0.5 * a * c.dt2;
        const double x = v + a * c.deltaT;
        vn = dmin(dmax(c.vm, x), c.vM);                  // (v_max_f64 / v_min_f64: a compare + two selects each otherwise)
        if (!ctl) vn = c.v0;
    }
    static PVE_HD double clip_a(const PVE_AS4 Const &c, double x)
    {   // min(aM, max(am, x)), ref :1502, 1521
        return dmin(c.aM, dmax(x, c.am));
    }
    static PVE_HD void ph_step1(const PVE_AS4 Const &c, const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r)
    {
        r.ctl = 0; r.lane = 0; r.j = 0; r.a0 = r.a1 = 0;
        if constexpr (Sh::HOME) {                         // the state lives in LDS (prologue / S3's publish / STAGE left it there)
            r.p = sh.p[t]; r.v = sh.v[t]; r.a = sh.a[t]; r.act = sh.act_next[t];
        }
        if (r.alive) {
            const int lane = (r.meta >> M_LANE_SHIFT) & M_LANE_MASK;    // (== slot_lane(sh.hd, t): slots are sorted by lane)
            r.lane = lane; r.j = t - sh.hd.lane_start[lane];
            sh.lane_of[t] = (uint8_t)lane;
            if constexpr (Sh::HAS_LJ) sh.lj[t] = (lane << 16) | r.j;
            r.ctl = (r.meta & M_CONTROL) ? 1 : 0;
            double act = r.act;
            if (P.mask_uncontrolled && !r.ctl) act = 0.0;   // main.py:401
            double target = clip_a(c, act);                                       // ref :1502
            if ((r.meta & M_LOCK) && (r.meta & (M_LOCKA_POS | M_LOCKA_NEG)) && r.p > 70)   // ref :1503-1505
                target = r.a + ((r.meta & M_LOCKA_POS) ? 1.0 : -1.0);
            r.meta &= ~(M_LOCK | M_LOCKA_POS | M_LOCKA_NEG);                      // ref :1506-1507
            bool ovr = ((sh.hd.head_valid >> lane) & 1) && sh.hd.head_lane[lane] == lane &&
                       sh.hd.head_j[lane] == r.j;                                // ref :1517-1518 (stale j)
            ovr = ovr || (lane % 3 == 2);                                         // ref :1519-1520
            r.a0 = clip_a(c, ovr ? c.aM : target);                                // ref :1521
            r.a1 = clip_a(c, ovr ? c.aM : c.am);                                  // ref :1516
            double pn, vn;
            outcome(c, r.p, r.v, r.a0, r.ctl, pn, vn); sh.p[t] = pn; sh.v[t] = vn;
            outcome(c, r.p, r.v, r.a1, r.ctl, pn, vn); sh.p1[t] = pn; sh.v1[t] = vn;
        }
        {
            u64 *const ms[2] = {sh.m_alive, sh.m_ctl};
            const bool fs[2] = {r.alive != 0, r.alive && r.ctl};
            vote_many<NW>(ms, t, fs);
        }
    }

    // ============================================================== S2: brake decision per front outcome
    static PVE_HD void ph_step2(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        int bb = 0;
        if (r.alive && r.j > 0 && r.ctl && mask_test(sh.m_ctl, t - 1)) {         // ref :1509-1510
            bb = brake_needed(c, r.p, r.v, sh.p[t - 1], sh.v[t - 1]) |
                 (brake_needed(c, r.p, r.v, sh.p1[t - 1], sh.v1[t - 1]) << 1);
        }
        sh.bb[t] = (uint8_t)bb;
        if constexpr (Sh::DENSE) {                        // dense mapping: the c-th controlled vehicle lives in slot t
            if (r.alive && r.ctl) sh.slot_of[mask_rank<NW>(sh.m_ctl, t)] = (uint8_t)t;
        }
    }

    // ============================================================== S3: resolve the in-lane chain
    // HOME: this tick's jerk leaves the registers right here -- a controlled vehicle's goes to its virdis[] cell (for the dense
    // thread, as BUILD does otherwise) and into jerk_sum (ref :321; FX otherwise -- nobody reads jerk_sum in between); an
    // uncontrolled vehicle's is only needed as persistent state, i.e. on the LAST tick of a launch / queue item: it waits in
    // the slot's act_next[] cell (no next action to park then; BUILD's xy32 overlay only writes controlled slots).
    static PVE_HD void ph_step3(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r, bool last = false)
    {
        if (r.alive) {
            int k = t;
            while (sh.bb[k] == 1 || sh.bb[k] == 2) k--;   // first slot whose decision ignores its front (j==0 -> 0)
            int br = sh.bb[k] & 1;
            for (int q = k + 1; q <= t; q++) br = (sh.bb[q] >> br) & 1;
            double an = br ? r.a1 : r.a0;
            r.jerk = an - r.a;                                                    // ref :1522
            r.a = an;
            r.p = br ? sh.p1[t] : sh.p[t];
            r.v = br ? sh.v1[t] : sh.v[t];
            r.step += 1;                                                          // ref :1533
            if constexpr (Sh::HOME) {
                if (r.ctl) { sh.virdis[t] = r.jerk; sh.h_jerk_sum[t] = sh.h_jerk_sum[t] + fabs(r.jerk * c.inv_dt); }
                else if (last) sh.act_next[t] = r.jerk;
            }
        }
    }
    static PVE_HD void ph_step3_publish(int t, Sh &sh, Regs &r)
    {
        if (r.alive) { sh.p[t] = r.p; sh.v[t] = r.v; sh.a[t] = r.a; }
    }

    // ============================================================== LISTS: offsets (runs beside S2 / S3)
    // thread d < 12: size and segment layout of virtual-lane list d from the controlled-vehicle ballot.
    static PVE_HD void ph_lists_a(const PVE_AS4 Const &c, int t, Sh &sh)
    {
        int cnt = 0;
        if (t < NL * 5) {                                 // lane t: segment k of list d
            const int d = t / 5, k = t - d * 5;
            const int Lp = (k == 0) ? d : (int)(sh.l2lp[d][k == 0 ? 0 : k - 1] & 15);
            if (Lp != 15) {
                const int b0 = mask_below<NW>(sh.m_ctl, sh.hd.lane_start[Lp]);
                cnt = mask_below<NW>(sh.m_ctl, sh.hd.lane_start[Lp + 1]) - b0;
                if (k == 0) { sh.cstart[d] = (int16_t)b0; if (d == NL - 1) sh.cstart[NL] = (int16_t)(b0 + cnt); }
            }
        }
#if PVE_DEVICE_CODE
        const int incl = wave_incl_scan(t, cnt, nullptr);           // all 60 segments live in wave 0
#else
        const int incl = wave_incl_scan(t, cnt, &sh.emu_scan);
#endif
        if (t < NL * 5) sh.pref[t] = (int16_t)incl;
    }
    static PVE_HD void ph_lists_b(int t, Sh &sh)
    {
        if (t < NL * 5) {
            const int d = t / 5, k = t - d * 5;
            const int base = d ? sh.pref[5 * d - 1] : 0;
            const int mine = t ? sh.pref[t - 1] : 0;
            sh.segoff[d][k] = (int16_t)(mine - base);
            if (k == 0) { sh.loff[d] = (int16_t)base; sh.nfin[d] = sh.pref[t] - base; }   // (own segment: all finite)
        }
        if (t == NL * 5) sh.loff[NL] = sh.pref[NL * 5 - 1];
    }

    // ============================================================== BUILD: every controlled vehicle files
    // itself into its own lane's list and into the lists of the lanes it conflicts with (ref :240-270)
    // HOME (entry pool of 304 entries): the lists of a tick are worked in GROUPS of consecutive lists [d0, d1) that fit the
    // pool -- BUILD files, RANK sorts and WALK reads one group per pass, entry indices relative to the group's first list.  One
    // group is the rule (loff[NL] = own + conflict entries of every controlled vehicle <= 5 per vehicle; 304 hold ~80
    // controlled vehicles); a single list never exceeds CAP entries, so every group makes progress.  Uniform.
    static PVE_HD int group_end(const Sh &sh, int d0)
    {
        if constexpr (!Sh::HOME) return NL;
        else {
            const int b = sh.loff[d0];
            if (sh.loff[NL] - b <= Sh::POOL) return NL;   // (the rule: one read)
            int d1 = d0 + 1;
            while (d1 < NL && sh.loff[d1 + 1] - b <= Sh::POOL) d1++;
            return d1;
        }
    }
    static PVE_HD void ph_build(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        ph_build_prep(c, t, sh, r);
        ph_build_fill<false>(c, t, sh, r, 0, NL);
    }
    static PVE_HD void ph_build_prep(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        sh.rew_ovr[t] = 0; sh.hdr[t] = -1;                // their storage was bb / pref until the barrier before BUILD
        // carried in LDS, not in a register (WALK overwrites it for the controlled vehicles, FIN reads it back).  A controlled
        // vehicle's old vir_dis is dead (ref :1348-1354 rewrite it): its cell hands this tick's jerk to the dense thread.
        // (HOME: S3 has done it; an uncontrolled vehicle's vir_dis stays where it is)
        if constexpr (!Sh::HOME) sh.virdis[t] = (r.alive && r.ctl) ? r.jerk : r.vir_dis;
        // ---- dense mapping from here: thread t works for the t-th controlled vehicle
        r.dctl = t < mask_count<NW>(sh.m_ctl);
        r.ds = 0; r.dlane = 0; r.dp = 0;
        if (!r.dctl) return;
        const int sl = sh.slot_of[t];
        const int lane = sh.lane_of[sl];
        const double p = sh.p[sl];
        r.ds = sl; r.dlane = lane; r.dp = p;
        get_xy_f32(c, p, lane, sh.xy32[sl][0], sh.xy32[sl][1]);
    }
    // the entries of the lists [d0, d1) (HOME: one group of lists per pass; else all twelve)
    // GRP = false: every list in one pass (the rule, and the only form of the blocks with a 5 CAP pool): no group arithmetic at all
    template <bool GRP = Sh::HOME>
    static PVE_HD void ph_build_fill(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r, int d0, int d1)
    {
        if (!r.dctl) return;
        const int sl = r.ds, lane = r.dlane;
        const double p = r.dp;
        const int gb = GRP ? (int)sh.loff[d0] : 0;            // entry index of the group's first list
        const int q = t - sh.cstart[lane];                    // its rank among the controlled vehicles of its lane
        if (!GRP || (lane >= d0 && lane < d1)) {
            const int e = sh.loff[lane] + q - gb;             // own lane: vd = p (ref :242-249)
            sh.u_vd[e] = p; sh.u_slot[e] = (uint8_t)sl; sh.u_list[e] = (uint8_t)lane;
        }
        if (lane % 3 == 2) return;                            // right turns conflict with nobody (ref :156)
        // three batches of independent LDS reads (lane tables, distance tables + list offsets), then the writes: written
        // as one loop per k the compiler emits four serial chains of three round trips each
        const int pk = *(const int *)sh.l2lp[lane];           // 4 x (lane | position << 4)
        int d[4], kk[4], mk[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {                        // (none = 15 never on these lanes; a conflicting lane is a left turn or a
            d[k] = (pk >> (8 * k)) & 15; kk[k] = (pk >> (8 * k + 4)) & 3; mk[k] = (pk >> (8 * k + 4)) & 7;   //  straight: m = lane % 3 in {0, 1})
        }
        double tA[4], tB[4], tC[4]; int lo[4], so[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            tA[k] = (&sh.tabA[0][0])[mk[k]]; tB[k] = (&sh.tabB[0][0])[mk[k]]; tC[k] = (&sh.tabC[0][0])[mk[k]];   // tab?[m][kk]
            lo[k] = sh.loff[d[k]]; so[k] = sh.segoff[d[k]][kk[k] + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { PVE_PIN(tA[k]); PVE_PIN(tB[k]); PVE_PIN(tC[k]); PVE_PIN(lo[k]); PVE_PIN(so[k]); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double delta = p - tA[k] + tB[k];                            // ref :733-803 (the relation is symmetric: kk =
            const double vd = (delta > 0) ? (tC[k] + delta) : INFINITY;        // our position inside lane2lane[d]); not chosen -> never sorted
            const int e = lo[k] + so[k] + q - gb;
            if constexpr (GRP) { if (d[k] < d0 || d[k] >= d1) continue; }   // (another pass's list)
            sh.u_vd[e] = vd; sh.u_slot[e] = (uint8_t)sl; sh.u_list[e] = (uint8_t)d[k];
            lds_add(&sh.nfin[d[k]], (delta > 0) ? 1 : 0);                       // (unconditional: no guarded block per entry)
        }
    }

    // ============================================================== RANK: counting sort of every list by
    // (vd, slot) = the reference's stable sort of the (lane, j)-ordered list by vd (ref :271)
    //
    // CAP = 128: the position is the number of smaller distances only (2 vector instructions per list element instead of
    // 4 with the count of equal ones).  Entries of one list that share the same distance -- vehicles of symmetric lanes
    // that spawned in the same tick, before their first controlled step -- compute the same position; they are found by
    // the claim itself: s_idx[position] is taken with ONE exchange that stores entry | tag << 16, tag = this tick's (and
    // env's) 16-bit stamp.  Whoever gets a word with the current stamp back is not the first at that position and files
    // the whole run of equal distances in slot order (every later arrival rewrites the same values; the LDS executes
    // the operations of a wave in order, so the last fix-up is the last write).  A stale word that happens to carry the
    // stamp (left-over LDS contents) only sends an entry through the fix-up, which then files just itself.
    template <bool GRP = false>
    static PVE_HD void ph_rank(int t, Sh &sh, int salt = 0, int d0 = 0, int d1 = NL)
    {
        const int gb = GRP ? (int)sh.loff[d0] : 0;        // (GRP: the lists [d0, d1) of this pass, entries relative to the first)
        const int M = (GRP ? (int)sh.loff[d1] : (int)sh.loff[NL]) - gb;
        const unsigned tag = Sh::DIRECT ? 0u : (((unsigned)sh.hd.ticks + (unsigned)salt * 0x9E37u) << 16);
        for (int e = t; e < M; e += CAP) {
            const int d = sh.u_list[e];
            const double vd = sh.u_vd[e];
            if (!(vd < INFINITY)) continue;               // not chosen (ref :259-270): not part of the list; the finite entries
                                                          // occupy the sorted positions [0, nfin[d])
            const int lo = sh.loff[d] - gb, hi = sh.loff[d + 1] - gb;
            int pos = 0, eq = 0;
            int f = lo;
            for (; f + 8 <= hi; f += 8) {                 // 8 independent LDS reads per round
                const double w0 = sh.u_vd[f], w1 = sh.u_vd[f + 1], w2 = sh.u_vd[f + 2], w3 = sh.u_vd[f + 3];
                const double w4 = sh.u_vd[f + 4], w5 = sh.u_vd[f + 5], w6 = sh.u_vd[f + 6], w7 = sh.u_vd[f + 7];
                pos += (w0 < vd) + (w1 < vd) + (w2 < vd) + (w3 < vd) + (w4 < vd) + (w5 < vd) + (w6 < vd) + (w7 < vd);
                if (Sh::DIRECT) eq += (w0 == vd) + (w1 == vd) + (w2 == vd) + (w3 == vd) + (w4 == vd) + (w5 == vd) + (w6 == vd) + (w7 == vd);
            }
            if (f < hi) {                                 // tail (< 8 entries): one more round of independent reads of
                const int n = hi - f;                     // CONSECUTIVE entries; what lies behind the list is masked out (the
                double w[7];                              // last list may run up to 6 doubles into `virdis`, the next member)
                const double *uv = sh.u_vd;
#pragma unroll
                for (int k = 0; k < 7; k++) w[k] = uv[f + k];
#pragma unroll
                for (int k = 0; k < 7; k++) { pos += (k < n) & (w[k] < vd); if (Sh::DIRECT) eq += (k < n) & (w[k] == vd); }
            }
            const int myslot = sh.u_slot[e];
            const int nown = sh.cstart[d + 1] - sh.cstart[d];                     // own-lane segment comes first
            if (Sh::DIRECT) {
                if (eq > 1)                               // exact vd ties: lower slot first
                    for (f = lo; f < hi; f++) pos += (sh.u_vd[f] == vd && sh.u_slot[f] < myslot) ? 1 : 0;
                sh.s_vd[lo + pos] = vd; sh.s_slot[lo + pos] = (uint8_t)myslot;
                if (e - lo < nown) sh.mypos[myslot] = (uint8_t)pos;
            } else {
                if (e - lo < nown) lds_store_relaxed(&sh.mypos[myslot], (uint8_t)pos);   // (before the claim -- released by it: a fix-up comes after it)
                const unsigned old = lds_xchg(&sh.s_idx[Sh::DIRECT ? 0 : lo + pos], tag | (unsigned)e);
                if ((old & 0xFFFF0000u) == tag) {         // somebody with the same distance was here first (or a stale word)
                    for (f = lo; f < hi; f++) {
                        if (!(sh.u_vd[f] == vd)) continue;
                        const int sf = sh.u_slot[f];
                        int rk = 0;
                        for (int g = lo; g < hi; g++) rk += (sh.u_vd[g] == vd && sh.u_slot[g] < sf) ? 1 : 0;
                        lds_store_relaxed(&sh.s_idx[Sh::DIRECT ? 0 : lo + pos + rk], tag | (unsigned)f);
                        if (f - lo < nown) lds_store_relaxed(&sh.mypos[sf], (uint8_t)(pos + rk));
                    }
                }
            }
        }
    }

    // Predecessor and the 6 nearest of the vehicle at sorted position s of the list [base, base + n) (n = its entries, all
    // with finite distances, n >= 1: the ego's own entry): r.kr / r.kv, pr, pvd.
    // Shared by the 12-lane WALK and the general-geometry kernel (per-route lists, pve_tick_geo.h).
    //
    // The 6 nearest in the reference's stable |vd - vd_self| sort = the 6 smallest keys (|d|, vd, slot) (ref :1383-1397).
    // Both sides of our own position are already sorted by |d|: left = keys below ours (walking left), right = keys above
    // (walking right); on equal |d| the left entry comes first (smaller vd, or equal vd and smaller slot).
    // FAST PATH: every candidate of the prefetched 6 + 6 window becomes ONE 32-bit key: the float32 image of |d| (rounding
    // is monotone, non-negative floats order like unsigned integers) with its low 4 bits replaced by the candidate code
    // (0..5 = walking index on the left, 8..13 = 8 + index on the right).  min(L_i, R_5-i) are the 6 smallest overall
    // (first half of a bitonic merge), a 12-comparator network sorts them: one v_min_u32 + one v_max_u32 per comparator
    // instead of a float64 compare and six selects.  The result is exact when (a) the 6 (winner, loser) pairs of the merge
    // step differ in the upper 28 bits, (b) the float64 distances of the sorted winners do not decrease (checked on the
    // values that are read for the output anyway) and (c) no two LEFT neighbours (incl. the first one beyond the window)
    // share the same float64 DISTANCE -- equal vd, or vd one ulp apart so that the subtraction rounds them together (such a
    // run is emitted in list order = ascending vd, slot, i.e. against the walking direction).  Anything else
    // -- near ties, ~1e-6 relative, or equal-vd runs -- takes walk_window_exact below, which decides on float64 keys.
    static PVE_HD unsigned f32_bits(float x)
    {
#if PVE_DEVICE_CODE
        return __float_as_uint(x);
#else
        unsigned u; memcpy(&u, &x, 4); return u;
#endif
    }
    static PVE_HD unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
    static PVE_HD unsigned umax(unsigned a, unsigned b) { return a < b ? b : a; }
    static PVE_HD void walk_window(Sh &sh, int base, int n, int s, double ps, Regs &r, int &pr, double &pvd)
    {
        const auto *sidx = sh.s_idx + base;      // sorted position -> entry
#define sv(pos_) (Sh::DIRECT ? sh.s_vd[Sh::DIRECT ? base + (pos_) : 0] : sh.u_vd[sidx_at(sidx, pos_)])
#define ss(pos_) (Sh::DIRECT ? sh.s_slot[Sh::DIRECT ? base + (pos_) : 0] : sh.u_slot[sidx_at(sidx, pos_)])
        // All window reads are UNCONDITIONAL on clamped positions and the validity is applied to the keys afterwards: a
        // guarded LDS read is a basic block of its own (index read, wait, value read, wait) and thirteen of them in a row
        // are twenty-six serial LDS round trips; this way the 13 index reads go out back to back, then the value reads.
        // Only the distances of the window are read (and the predecessor's slot): validity follows from the positions,
        // the slots of the 6 winners are read once they are known.
        const int last = n - 1;
        double lraw[NNB + 1], rraw[NNB];
        int prs = -1;
#pragma unroll
        for (int i = 0; i < NNB + 1; i++) {
            const int pos = s - 1 - i, pc = pos >= 0 ? pos : 0;
            lraw[i] = sv(pc);
            if (i == 0) prs = (int)ss(pc);
        }
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const int pos = s + 1 + i, pc = pos <= last ? pos : last;
            rraw[i] = sv(pc);
        }
        if (Sh::PIN_READS) {                          // (not in the general-geometry kernel: more spills there)
#pragma unroll
            for (int i = 0; i < NNB + 1; i++) PVE_PIN(lraw[i]);
#pragma unroll
            for (int i = 0; i < NNB; i++) PVE_PIN(rraw[i]);
            PVE_PIN(prs);
        }
        pr = (s > 0) ? prs : -1; pvd = (s > 0) ? lraw[0] : 0.0;                      // ref :1353-1354
        // Two LEFT neighbours with the same float64 distance (equal vd, or vd one ulp apart: the subtraction rounds) come in
        // list order, i.e. ascending vd = against the walking direction: such a run (incl. the first entry beyond the
        // window) takes the exact path.  (bitwise, not short-circuit: six compares, no branches)
        double dl[NNB + 1];
#pragma unroll
        for (int i = 0; i < NNB + 1; i++) dl[i] = fabs(lraw[i] - ps);                 // ref :1388
        bool amb = false;
#pragma unroll
        for (int i = 1; i < NNB + 1; i++) amb = amb | ((s - 1 - i >= 0) & (dl[i] == dl[i - 1]));
        unsigned kl[NNB], kq[NNB];
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const unsigned cl = (unsigned)i, cr = 8u + (unsigned)i;
            const unsigned bl = f32_bits((float)dl[i]), br = f32_bits((float)fabs(rraw[i] - ps));
            kl[i] = (s - 1 - i >= 0) ? ((bl & ~15u) | cl) : (0xFF800000u | (cl << 4) | cl);
            kq[i] = (s + 1 + i <= last) ? ((br & ~15u) | cr) : (0xFF800000u | (cr << 4) | cr);
        }
        unsigned w[NNB], acc = ~0u;                   // acc = the smallest XOR of a (winner, loser) pair of the merge step
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const unsigned a = kl[i], b = kq[NNB - 1 - i];
            w[i] = umin(a, b);
            acc = umin(acc, a ^ b);
        }
#define PVE_CE(A, B) { const unsigned lo_ = umin(w[A], w[B]), hi_ = umax(w[A], w[B]); w[A] = lo_; w[B] = hi_; }
        PVE_CE(0, 5) PVE_CE(1, 3) PVE_CE(2, 4)
        PVE_CE(1, 2) PVE_CE(3, 4)
        PVE_CE(0, 3) PVE_CE(2, 5)
        PVE_CE(0, 1) PVE_CE(2, 3) PVE_CE(4, 5)
        PVE_CE(1, 2) PVE_CE(3, 4)
#undef PVE_CE
        amb = amb | (acc < 16u);
        // the winners' slots and distances: 6 index reads back to back, then the 12 value reads (not 6 serial chains)
        int ei[NNB], sl[NNB]; double vv[NNB];
#pragma unroll
        for (int k = 0; k < NNB; k++) {
            const int code = (int)(w[k] & 15u);
            const int pos = (code & 8) ? (s - 7 + code) : (s - 1 - code);        // = s + 1 + (code - 8) on the right
            const int pc = ((int)w[k] >= 0) ? pos : 0;                           // (the keys of absent candidates have bit 31 set)
            ei[k] = Sh::DIRECT ? (base + pc) : sidx_at(sidx, pc);
        }
#pragma unroll
        for (int k = 0; k < NNB; k++) PVE_PIN(ei[k]);
#pragma unroll
        for (int k = 0; k < NNB; k++) {
            sl[k] = Sh::DIRECT ? (int)sh.s_slot[Sh::DIRECT ? ei[k] : 0] : (int)sh.u_slot[ei[k]];
            vv[k] = Sh::DIRECT ? sh.s_vd[Sh::DIRECT ? ei[k] : 0] : sh.u_vd[ei[k]];
        }
#pragma unroll
        for (int k = 0; k < NNB; k++) { PVE_PIN(sl[k]); PVE_PIN(vv[k]); }
        // The order of the winners is the reference's iff their float64 distances do not decrease: keys that differ in
        // the upper 28 bits are ordered exactly; keys that agree there come left before right (by code), which is right
        // for an exact |d| tie (the frequent case: equally spaced platoons) and wrong only if the right one is nearer.
        // (two lefts / two rights are in walking order = ascending |d| anyway)
        double dk[NNB];
#pragma unroll
        for (int k = 0; k < NNB; k++) dk[k] = fabs(vv[k] - ps);
        // Round 6: a decrease can only sit between a left and a right winner whose float32 images agree (rounding is monotone,
        // one side alone is in walking order) -- the ulp-level near ties of equally spaced platoons, 0.45 % of the egos of the
        // headline workload, i.e. every fifth wave-tick used to run walk_window_exact for one of its ~51 egos.  The winners
        // are the right SET (the merge step's pairs are separated); one pass of adjacent exchanges on the float64 distances
        // puts a pair in order, in the waves that hold one; a longer run (rare: 14 of 4.2 M egos) still takes the exact path.
        bool dec = false;
#pragma unroll
        for (int k = 1; k < NNB; k++) dec = dec | (((int)w[k] >= 0) & (dk[k] < dk[k - 1]));
#if PVE_DEVICE_CODE
        if (__builtin_amdgcn_ballot_w64(dec) != 0)
#else
        if (dec)
#endif
        {
#pragma unroll
            for (int k = 1; k < NNB; k++) {
                const bool sw = ((int)w[k] >= 0) & (dk[k] < dk[k - 1]);
                const double td = sw ? dk[k - 1] : dk[k], tv = sw ? vv[k - 1] : vv[k]; const int ts = sw ? sl[k - 1] : sl[k];
                dk[k - 1] = sw ? dk[k] : dk[k - 1]; vv[k - 1] = sw ? vv[k] : vv[k - 1]; sl[k - 1] = sw ? sl[k] : sl[k - 1];
                dk[k] = td; vv[k] = tv; sl[k] = ts;
            }
#pragma unroll
            for (int k = 1; k < NNB; k++) amb = amb | (((int)w[k] >= 0) & (dk[k] < dk[k - 1]));
        }
        if (!amb) {
#pragma unroll
            for (int k = 0; k < NNB; k++) {
                const bool ok = (int)w[k] >= 0;
                r.kr[k] = ok ? sl[k] : -1;
                r.kv[k] = ok ? vv[k] : 0.0;
            }
        } else {
            walk_window_exact(sh, base, n, s, ps, r);
        }
#undef sv
#undef ss
    }

    // The same selection decided on the float64 keys (exact |d| ties and near ties; rare): float64 merge + 12-comparator
    // network, and for runs of equal vd on the left or equal |d| among the winners the general pointer walk.
    static PVE_HD void walk_window_exact(Sh &sh, int base, int n, int s, double ps, Regs &r)
    {
        const auto *sidx = sh.s_idx + base;
#define sv(pos_) (Sh::DIRECT ? sh.s_vd[Sh::DIRECT ? base + (pos_) : 0] : sh.u_vd[sidx_at(sidx, pos_)])
#define ss(pos_) (Sh::DIRECT ? sh.s_slot[Sh::DIRECT ? base + (pos_) : 0] : sh.u_slot[sidx_at(sidx, pos_)])
        const int last = n - 1;
        double lv[NNB + 1], rv[NNB];
#pragma unroll
        for (int i = 0; i < NNB + 1; i++) { const int pos = s - 1 - i; lv[i] = pos >= 0 ? sv(pos >= 0 ? pos : 0) : -INFINITY; }
#pragma unroll
        for (int i = 0; i < NNB; i++) { const int pos = s + 1 + i; rv[i] = pos <= last ? sv(pos <= last ? pos : last) : INFINITY; }
        const double pvd = lv[0];
        bool tie = false;                             // (runs of equal DISTANCE on the left: equal vd, or vd one ulp apart)
#pragma unroll
        for (int i = 1; i < NNB + 1; i++) tie = tie || (s - 1 - i >= 0 && fabs(lv[i] - ps) == fabs(lv[i - 1] - ps));
        // candidate = (d, code): code = walking index on the left (0..5) or 8 + index on the right.  min(L_i, R_5-i)
        // (left wins equal d) are the 6 smallest overall; they are then sorted by d alone with a 12-comparator
        // network.  If two of the 6 winners still share the same d (exact |d| ties, quantised states) the order
        // would need the (side, index) tie-break: that case also takes the general path.
        double cd[NNB]; int cc[NNB];
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const double dl = (s - 1 - i >= 0) ? fabs(lv[i] - ps) : INFINITY;                   // ref :1388
            const int j = NNB - 1 - i;
            const double dr = (s + 1 + j <= last) ? fabs(rv[j] - ps) : INFINITY;
            const bool takeL = dl <= dr;
            cd[i] = takeL ? dl : dr; cc[i] = takeL ? i : (8 + j);
        }
#define PVE_CE(A, B)                                                                            \
        {                                                                                       \
            const bool sw = cd[B] < cd[A];                                                      \
            const double td = sw ? cd[A] : cd[B]; const int tc = sw ? cc[A] : cc[B];            \
            cd[A] = sw ? cd[B] : cd[A]; cc[A] = sw ? cc[B] : cc[A];                             \
            cd[B] = td; cc[B] = tc;                                                             \
        }
        PVE_CE(0, 5) PVE_CE(1, 3) PVE_CE(2, 4)
        PVE_CE(1, 2) PVE_CE(3, 4)
        PVE_CE(0, 3) PVE_CE(2, 5)
        PVE_CE(0, 1) PVE_CE(2, 3) PVE_CE(4, 5)
        PVE_CE(1, 2) PVE_CE(3, 4)
#undef PVE_CE
#pragma unroll
        for (int k = 1; k < NNB; k++) tie = tie || (cd[k] == cd[k - 1] && cd[k] < INFINITY);
        if (!tie) {
#pragma unroll
            for (int k = 0; k < NNB; k++) {
                const bool ok = cd[k] < INFINITY;
                const int code = cc[k];
                const int pos = (code < 8) ? (s - 1 - code) : (s + 1 + (code - 8));
                r.kr[k] = ok ? (int)ss(ok ? pos : 0) : -1;
                r.kv[k] = ok ? sv(ok ? pos : 0) : 0.0;
            }
        } else {
            // GENERAL PATH: pointer walk; a run of equal distance on the left is emitted in list order (ascending vd, slot)
#pragma unroll
            for (int k = 0; k < NNB; k++) { r.kr[k] = -1; r.kv[k] = 0; }
            int hi = s - 1, lo = hi, cur, rr = s + 1;
            if (hi >= 0) { const double dh = fabs(pvd - ps); while (lo > 0 && fabs(sv(lo - 1) - ps) == dh) lo--; }
            cur = lo;
            for (int k = 0; k < NNB; k++) {
                const bool hasL = hi >= 0;
                const bool hasR = rr < n;
                if (!(hasL || hasR)) break;
                double vR = INFINITY;
                if (hasR) vR = sv(rr);
                double vL = 0;
                if (hasL) vL = sv(cur);
                const double dL = fabs(vL - ps), dR = fabs(vR - ps);              // ref :1388
                const bool takeL = hasL && (!hasR || dL <= dR);
                int slot; double vv;
                if (takeL) {
                    slot = ss(cur); vv = vL;
                    cur++;
                    if (cur > hi) {
                        hi = lo - 1; lo = hi;
                        if (hi >= 0) { const double dh = fabs(sv(hi) - ps); while (lo > 0 && fabs(sv(lo - 1) - ps) == dh) lo--; }
                        cur = lo;
                    }
                } else {
                    slot = ss(rr); vv = vR;
                    rr++;
                }
                // static register indices only (no scratch): select the destination by k
#pragma unroll
                for (int q = 0; q < NNB; q++) if (q == k) { r.kr[q] = slot; r.kv[q] = vv; }
            }
        }
#undef sv
#undef ss
    }

    // ============================================================== WALK: predecessor, 6 nearest, reward, hit
    static PVE_HD void ph_scan_init(Regs &r)
    {
        r.reward = 0; r.hit = 0; r.hdr = -1; r.djerk = 0;
#pragma unroll
        for (int k = 0; k < NNB; k++) { r.kr[k] = -1; r.kv[k] = 0; }
    }
    static PVE_HD void ph_scan(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        ph_scan_init(r);
        ph_scan_lists<false>(c, t, sh, r, 0, NL);
    }
    // (GRP: the lists [d0, d1) of this pass; ph_scan_init once in front of the first pass)
    template <bool GRP = Sh::HOME>
    static PVE_HD void ph_scan_lists(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r, int d0, int d1)
    {
        const int gb = GRP ? (int)sh.loff[d0] : 0;
        // thread d < 12: lane d is non-empty -> its list was rebuilt (ref :234); head persisted for next tick's step (ref :1517).
        // The valid bits of the rebuilt lists are combined by two ballots and ONE read-modify-write of the header word: an LDS
        // atomic or / and per lane on that one word is expanded by the compiler's atomic optimizer into a scalar loop over the
        // active lanes (s_ff1 / v_readlane / ... : ~7 instructions x 12 lanes on the first wave's path, every tick)
        const bool rebuilt = t < NL && sh.hd.lane_start[t < NL ? t + 1 : 0] > sh.hd.lane_start[t < NL ? t : 0] &&
                             (!GRP || (t >= d0 && t < d1));
        bool hvalid = false;
        if (rebuilt) {
            const int base = sh.loff[t] - gb;
            if (sh.nfin[t] > 0) {
                const int hr = Sh::DIRECT ? sh.s_slot[Sh::DIRECT ? base : 0] : sh.u_slot[sidx_at(sh.s_idx, base)];
                hvalid = true;
                int hl = sh.lane_of[hr];
                sh.hd.head_lane[t] = hl;
                sh.hd.head_j[t] = hr - sh.hd.lane_start[hl];
            }
        }
#if PVE_DEVICE_CODE
        {
            const unsigned mr = (unsigned)__ballot(rebuilt), mv = (unsigned)__ballot(hvalid);   // (lanes 0..11 of the first wave; 0 elsewhere)
            if (t == 0) sh.hd.head_valid = (sh.hd.head_valid & ~(int)mr) | (int)mv;
        }
#else
        if (rebuilt) { if (hvalid) sh.hd.head_valid |= 1 << t; else sh.hd.head_valid &= ~(1 << t); }
#endif
        if (!r.dctl) return;
        const int sl = r.ds, lane = r.dlane;
        if constexpr (GRP) { if (lane < d0 || lane >= d1) return; }   // (another pass's list)
        const double ps = r.dp;
        const int base = sh.loff[lane] - gb, n = sh.nfin[lane];   // (entries with a finite distance: the sorted list)
        int pr; double pvd;
        walk_window(sh, base, n, sh.mypos[sl], ps, r, pr, pvd);
        // ref :1348-1354
        r.hdr = pr;
        const double vd = (pr >= 0) ? (ps - pvd) : 100.0;
        r.djerk = sh.virdis[sl];                              // (parked there by the slot's thread in BUILD)
        sh.hdr[sl] = (int16_t)pr;
        sh.virdis[sl] = vd;
    }

    // ============================================================== REWARD: reward terms + XY collision test
    static PVE_HD void ph_reward(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        r.dcloser = 150;
        if (!r.dctl) return;
        const int sl = r.ds, lane = r.dlane;
        const double ps = r.dp;
        const double myv = sh.v[sl];                      // (published in S3)
        // ref :280-310
        double t_distance = 2, d_distance = 10;
        const int n0 = r.kr[0];
        if (n0 >= 0) {
            const double vdn = r.kv[0];
            d_distance = fabs(ps - vdn);
            r.dcloser = vdn;
            if (d_distance != 0) t_distance = (ps - vdn) / (myv - sh.v[n0] + 0.0001);
        }
        // ref :311-320
        double r_ = 0;
        if (0 < t_distance && t_distance < 4) r_ += reward_coth_term(t_distance);
        // divisions by constants become multiplications in these reward-only terms (no decision reads them)
        const double jd = r.djerk * c.inv_dt;
        r_ -= jd * jd * (3.0 / 3600.0);
        if (d_distance < 10) {
            double q1 = d_distance * 0.1, q2 = q1 * q1;
            r_ += reward_log_term(q2 * q2 * q1 + 0.00001);
        }
        r_ += (myv - c.vm) * c.inv_span * 2.0;
        r.reward = dmin(dmax(r_, -20.0), 20.0);
        // ref :322-334
        // pre-filter in single precision: the exact FP64 positions (two divisions + polynomials each) are only
        // evaluated when the pair is within 5 cm of the threshold band -- the decision itself is always FP64
        bool near = false;
        if (n0 >= 0) {
            const float fx = sh.xy32[n0][0] - sh.xy32[sl][0], fy = sh.xy32[n0][1] - sh.xy32[sl][1];
            const float lim = (float)c.collision_thr + 0.05f;
            near = fx * fx + fy * fy < lim * lim;
        }
        if (near) {
            double ax, ay, bx, by;
            get_xy(c, ps, lane, ax, ay);
            get_xy(c, sh.p[n0], sh.lane_of[n0], bx, by);
            double dx = bx - ax, dy = by - ay;
            double dxy = sqrt(dx * dx + dy * dy);
            if (fabs(dxy) < c.collision_thr) {
                lds_add(&sh.cnt[sl], 1);                          // our own hit (ref :333), seen this tick
                lds_add(&sh.cnt[n0], (sl < n0) ? 1 : (1 << 16));  // seen by n0 this tick only if we precede it
            }
        }
    }

    // ============================================================== FX: ordered effects
    static PVE_HD void ph_effects(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
#if !PVE_DEVICE_CODE
        if (t == 0) {
            sh.red_reward[0] = 0; sh.red_jerk[0] = 0;               // emulator: block_sum accumulates; storage was segoff
            for (int k = 0; k < NW; k++)                            // emulator: vote() ORs bits; storage was nfin
                sh.m_del[k] = sh.m_fin[k] = sh.m_ctlnow[k] = sh.m_coll[k] = sh.m_lead[k] = sh.m_spawn[k] = 0;
        }
#endif
        r.del = 0; r.fin = 0; r.coll_seen = 0; r.coll_fin = 0;
        int code = 0;                                     // what happens to the vehicle's reward (held by its dense thread)
        if constexpr (Sh::HOME) r.p = sh.p[t];            // (S3's publish: no register from S3 to here; unconditional: no merge with an older value)
        if (r.alive) {
            const int cc = sh.cnt[t];                     // hits: the vehicle's own one and those of the vehicles before it |
            sh.cnt[t] = 0;                                // those of the vehicles behind it << 16 (the cell is re-used by LOCK)
            const int prev = (r.meta >> M_COLL_SHIFT) & M_COLL_MASK;
            r.coll_seen = prev + (cc & 0xffff);                                   // ref :337-340
            r.coll_fin = r.coll_seen + (cc >> 16);
            if (r.ctl) {
                if constexpr (Sh::HOME) sh.h_count[t] += 1;                       // ref :292 (jerk_sum: S3)
                else {
                r.count += 1;                                                     // ref :292
                r.jerk_sum += fabs(r.jerk * c.inv_dt);                            // ref :321
                }
                if (r.coll_seen > 0) lds_add(&sh.acc_collisions, r.coll_seen);    // ref :337
            }
            if (r.p < c.exit_p || r.coll_seen > 0) {                              // ref :341-349
                r.del = 1;
                if (r.coll_seen > 0) {
                    if (r.ctl) code = 1;                                          // reward = -10
                    else {                                                        // reward[-1] of someone else
                        int pv = mask_prev<NW>(sh.m_ctl, t);
                        if (pv >= 0) sh.rew_ovr[pv] = 1;
                    }
                }
                r.meta |= M_DONE;
                sh.hdr[t] = -1;
            } else if (r.p < 0 && (r.meta & M_CONTROL)) {                         // ref :350-359
                r.fin = 1;
                r.meta |= M_DONE | M_FINISH;
                r.meta &= ~(M_CONTROL | M_LOCK);
                sh.hdr[t] = -1;
                code = 2;                                                         // reward = 5
                lds_add(&sh.acc_passed_steps, r.step);
            }
        }
        sh.fxcode()[t] = (uint8_t)code;
        {                                                 // (after this thread's own resets of hdr[t] above)
            const int h = sh.hdr[t];
            sh.chain()[t] = (uint8_t)(h < 0 ? CAP : h);
            if (t == 0) sh.chain()[CAP] = (uint8_t)CAP;
        }
        {
            u64 *const ms[5] = {sh.m_del, sh.m_fin, sh.m_ctlnow, sh.m_coll, sh.m_spawn};
            const bool fs[5] = {r.del != 0, r.fin != 0, r.alive && !r.del && (r.meta & M_CONTROL),
                                r.alive && r.ctl && r.coll_seen > 0,                                      // main.py:410-412
                                t < NL && sh.hd.current_time >= sh.hd.next_arr[t < NL ? t : 0]};        // ref :379
            vote_many<NW>(ms, t, fs);
        }
    }

    // ============================================================== LOCK: dead-lock scan + reductions
    // the next arrival time of a lane that spawns this tick: the only global load of the kernel's tail is issued here,
    // under the dead-lock scan, instead of in FIN where the first wave would sit out its latency
    static PVE_HD void ph_prefetch_arrival(const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r, int lane_num)
    {
        r.next_arr = INFINITY;
        if (t < NL && ((sh.m_spawn[0] >> t) & 1)) {
            const int rec1 = sh.hd.veh_rec[t] + 1;
            if (rec1 < P.rows) r.next_arr = P.arrivals[(size_t)env * P.arr_env_stride + (size_t)rec1 * lane_num + t];
        }
    }
    static PVE_HD void ph_lock(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r, bool last = false)
    {
        if constexpr (Sh::HOME) { if (r.dctl && last) sh.act_next[r.ds] = r.djerk; }   // (this tick's jerk back to the slot: ph_step3)
        if (r.dctl) {                                     // what FX decided for the vehicle's reward (ref :346, :357; reward[-1] -> rew_ovr)
            const int fx = sh.fxcode()[r.ds];
            const bool m10 = (sh.rew_ovr[r.ds] != 0) | (fx == 1);
            r.reward = m10 ? -10.0 : (fx == 2 ? 5.0 : r.reward);
            if constexpr (Sh::HOME) sh.h_closer[r.ds] = r.dcloser;   // closer_p (ref :302) straight to its home
            else ((double *)sh.xy32)[r.ds] = r.dcloser;   // closer_p (ref :302) back to the slot's thread; xy32 is dead after REWARD
        }
        block_sum(sh.red_reward, t, r.dctl ? r.reward : 0.0);
        if constexpr (Sh::HOME) {
            double js = 0;
            if (r.fin) js = sh.h_jerk_sum[t];             // (FX's update of this very thread)
            block_sum_sparse(sh.red_jerk, t, r.fin, js);                          // ref :358
        } else
        block_sum_sparse(sh.red_jerk, t, r.fin, r.jerk_sum);                      // ref :358
        // Dead-lock scan (ref :365-370, :1469-1499), member-parallel: every controlled vehicle follows the
        // virtual-header pointers for <= 10 hops; if the walk returns to itself it is on a cycle.  Cycles are rare:
        // the first walk only chases pointers; a member then walks its cycle again and learns the smallest slot (= the
        // member that discovers the cycle in the reference's (lane, j) scan order) and its own rank in the reference's
        // sorted record list [vir_dis, lane, j, ...] (ref :1486-1492).  The first member reserves a scratch range for the
        // cycle; the verdict word goes to the slot's thread through cnt[] (free since FX).
        bool lead = false;
        if (r.dctl && mask_test(sh.m_ctlnow, r.ds)) {
            const int s0 = r.ds;
            int cur = s0, len = 0;
#ifdef PVE_BRANCHY_LOCK_WALK                       // A/B build knob: the guarded form (a block of its own per hop)
            for (int hop = 0; hop < 10; hop++) {                                  // ref :1470-1478
                cur = sh.hdr[cur];
                if (cur < 0) break;
                if (cur == s0) { len = hop + 1; break; }
            }
#else
            // straight-line: the chain table (FX) maps "no header" and the sentinel to the sentinel, so ten unconditional
            // byte reads never leave it; the first return to s0 sets len and parks the walk on the sentinel
            const uint8_t *ch = sh.chain();
#pragma unroll
            for (int hop = 0; hop < 10; hop++) {                                  // ref :1470-1478
                cur = ch[cur];
                const bool back = cur == s0;
                len = back ? hop + 1 : len;
                cur = back ? CAP : cur;
            }
#endif
            if (len) {
                const double dv = sh.virdis[s0];
                int mn = s0, rank = 0;
                cur = s0;
                for (int hop = 1; hop < len; hop++) {
                    cur = sh.hdr[cur];
                    const double d = sh.virdis[cur];
                    rank += (d < dv || (d == dv && cur < s0)) ? 1 : 0;
                    mn = cur < mn ? cur : mn;
                }
                sh.cnt[s0] = 1 | (len << 1) | (rank << 5) | (mn << 9);
                lead = (mn == s0);
                if (lead) sh.cyc_off[s0] = (uint8_t)lds_claim(&sh.lead_n, len);
            }
        }
        vote<NW>(sh.m_lead, t, lead);
    }
    // LOCK2 (after a barrier): every cycle member files its record at its rank inside the cycle's scratch range
    // (u_vd / u_list are dead after the walk phase) = the reference's record_.sort() (ref :1492)
    static PVE_HD void ph_lock2(int t, Sh &sh, Regs &r, bool last = false)
    {
        if constexpr (Sh::HOME) { r.jerk = 0; if (last && r.alive) r.jerk = sh.act_next[t]; }   // (ph_step3; staged by FIN on the last tick only)
        r.cyc = sh.cnt[t];                                // (0 unless the dense thread of this slot's vehicle found a cycle)
        if constexpr (!Sh::HOME) if (r.alive && r.ctl) r.closer_p = ((double *)sh.xy32)[t];
        if (r.cyc & 1) {
            const int e = sh.cyc_off[r.cyc >> 9] + ((r.cyc >> 5) & 15);
            sh.u_vd[e] = sh.virdis[t];
            sh.lk_slot[e] = (uint8_t)t;
        }
    }
    // k_rollout only (same phase as LOCK2; the delete votes are complete since the barrier behind FX)
    static PVE_HD void ph_keep_prefix(int t, Sh &sh)
    {
        u64 keep[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) keep[k] = sh.m_alive[k] & ~sh.m_del[k];
        sh.keep_pre()[t] = (uint8_t)mask_rank<NW>(keep, t);
        if (t == 0) sh.keep_pre()[CAP] = (uint8_t)mask_count<NW>(keep);
    }
    // the same two phases with everything per slot (general-geometry kernel: no dense mapping there)
    static PVE_HD void ph_lock_slot(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        if (r.alive && r.ctl && sh.rew_ovr[t]) r.reward = -10;                    // ref :346 via reward[-1]
        block_sum(sh.red_reward, t, (r.alive && r.ctl) ? r.reward : 0.0);
        block_sum_sparse(sh.red_jerk, t, r.fin, r.jerk_sum);                      // ref :358
        // Dead-lock scan (ref :365-370, :1469-1499), member-parallel: every controlled vehicle follows the
        // virtual-header pointers for <= 10 hops; if the walk returns to itself it is on a cycle and has, on the
        // way, seen every other member: it knows the cycle length, the smallest slot (= the member that discovers
        // the cycle in the reference's (lane, j) scan order) and its own rank in the reference's sorted record
        // list [vir_dis, lane, j, ...] (ref :1486-1492).  The first member reserves a scratch range for the cycle.
        bool lead = false;
        r.cyc = 0;
        if (r.alive && !r.del && (r.meta & M_CONTROL)) {
            // cycles are rare: the first walk only chases pointers (one read per hop); a member walks its cycle again for
            // the smallest slot and its rank in the sorted record list
            int cur = t, len = 0, mn = t, rank = 0;
            {                                             // straight-line over the byte chain written in FX (cf. ph_lock)
                const uint8_t *ch = sh.chain();
#pragma unroll
                for (int hop = 0; hop < 10; hop++) {                              // ref :1470-1478
                    cur = ch[cur];
                    const bool back = cur == t;
                    len = back ? hop + 1 : len;
                    cur = back ? CAP : cur;
                }
            }
            const bool found = len > 0;
            if (found) {
                const double dv = sh.virdis[t];
                cur = t;
                for (int hop = 1; hop < len; hop++) {
                    cur = sh.hdr[cur];
                    const double d = sh.virdis[cur];
                    rank += (d < dv || (d == dv && cur < t)) ? 1 : 0;
                    mn = cur < mn ? cur : mn;
                }
                r.cyc = 1 | (len << 1) | (rank << 5) | (mn << 9);
                lead = (mn == t);
                if (lead) sh.cyc_off[t] = lds_claim(&sh.lead_n, len);
            }
        }
        vote<NW>(sh.m_lead, t, lead);
    }
    // LOCK2 (after a barrier): every cycle member files its record at its rank inside the cycle's scratch range
    // (u_vd / u_list are dead after the walk phase) = the reference's record_.sort() (ref :1492)
    static PVE_HD void ph_lock2_slot(int t, Sh &sh, Regs &r)
    {
        if (r.cyc & 1) {
            const int e = sh.cyc_off[r.cyc >> 9] + ((r.cyc >> 5) & 15);
            sh.u_vd[e] = sh.virdis[t];
            sh.lk_slot[e] = (uint8_t)t;
        }
    }
    // NOTE: if the tightest record's header is the tightest vehicle itself (1-cycle) the reference
    // writes +1 then -1; a vehicle is never its own predecessor, so cycles have length >= 2.

    // ============================================================== FIN: re-pack + write-back
    // Bytes are time in the store burst of FIN, so a vehicle that keeps its slot does not rewrite what
    // cannot have changed: with_ids = false skips the immutable fields (id, seq_in_lane, id_info[1]); with_carry = false
    // (it was not controlled this tick either) also skips jerk_sum / vir_dis / closer_p / count, which only the
    // controlled branch of scene_update touches (ref :292, :302, :321, :1348-1354).
    template <class R> static PVE_HD void store_slot(const PVE_AS4 Params &P, int env, int slot, const R &r, int meta, int hdr_word,
                                                     bool with_ids = true, bool with_carry = true)
    {
        *env_at<CAP>(P.f64[F_P], env, slot) = r.p; *env_at<CAP>(P.f64[F_V], env, slot) = r.v; *env_at<CAP>(P.f64[F_A], env, slot) = r.a; *env_at<CAP>(P.f64[F_JERK], env, slot) = r.jerk;
        if (with_carry) {
            *env_at<CAP>(P.f64[F_JERK_SUM], env, slot) = r.jerk_sum; *env_at<CAP>(P.f64[F_VIR_DIS], env, slot) = r.vir_dis; *env_at<CAP>(P.f64[F_CLOSER_P], env, slot) = r.closer_p;
            *env_at<CAP>(P.i32[I_COUNT], env, slot) = r.count;
        }
        if (with_ids) { *env_at<CAP>(P.i32[I_ID], env, slot) = r.id; *env_at<CAP>(P.i32[I_SEQ], env, slot) = r.seq; *env_at<CAP>(P.i32[I_VNUM], env, slot) = r.vnum; }
        *env_at<CAP>(P.i32[I_STEP], env, slot) = r.step; *env_at<CAP>(P.i32[I_META], env, slot) = meta;
        *env_at<CAP>(P.i32[I_HDR], env, slot) = hdr_word;
    }
    static PVE_HD int pack_lanej(const Sh &sh, int slot)
    {
        const int sc = slot < 0 ? 0 : slot;                   // unconditional reads on a clamped slot (no guarded LDS blocks)
        if constexpr (Sh::HAS_LJ) {
            const int w = sh.lj[sc];
            return slot < 0 ? -1 : w;
        } else {
            const int l = sh.lane_of[sc];
            const int w = (l << 16) | (sc - sh.hd.lane_start[l]);
            return slot < 0 ? -1 : w;
        }
    }

    // RES = false: the tick kernel -- state and header go back to HBM (`O` = P.out).
    // RES = true:  k_rollout -- outputs only (`O` = this tick's block of the output buffers); the persistent fields and
    //              the header updates are handed to ph_stage through `fc` and stay on the chip.
    // k_rollout, `still` ticks: when nobody is deleted and nobody spawns every vehicle keeps its slot, so nothing has to
    // move: no staging, no barriers A / B, no reload -- the registers simply carry over (about every second tick at the bench
    // load).  `full` (uniform) forces the staged form: the last tick of a launch, whose state FLUSH takes from the staging area.
    // adst (k_rollout with the actor on the chip): the post-compaction slot of the dense thread's vehicle (< 0: gone) for the
    // actor pass behind STAGE
    // ZROW: the fields of an absent neighbour are gathered from the zero cell instead of being selected (48 selects less;
    // only where the longer live ranges of the gathered values do not spill: the default and actor forms of k_rollout)
    template <bool RES, bool ZROW = false, class OutT>
    static PVE_HD void ph_final(const PVE_AS4 Const &c, const PVE_AS4 Params &P, const OutT &O, int env, int t, Sh &sh,
                                Regs &r, FinCarry &fc, bool full = true, int *adst = nullptr)
    {
        EnvHeader &gh = P.headers[env];
        const int N = sh.hd.n_alive;
        const size_t gpre = (size_t)env * CAP + t;
        const bool fused = RES || (P.mode == MODE_FUSED);
        // ---- spawn set (ref :361, :378-433); a full env defers the spawn (cursor not advanced)
        const unsigned want = (unsigned)(sh.m_spawn[0] & 0xFFFull);
        unsigned sp = want; int room = CAP - N;
        if (__builtin_popcount(want) > room) {            // (uniform, rare: a full intersection defers the spawns of the higher lanes)
            sp = 0;
#pragma unroll
            for (int l = 0; l < NL; l++) if ((want >> l) & 1) { if (room > 0) { sp |= 1u << l; room--; } }
        }
        const int n_over = __builtin_popcount(want) - __builtin_popcount(sp);
        // ---- keep mask: FUSED drops delete_veh now, SCENE keeps everybody (marked M_DEL)
        u64 keep[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) keep[k] = fused ? (sh.m_alive[k] & ~sh.m_del[k]) : sh.m_alive[k];
        bool still = RES && !full && sp == 0;
#pragma unroll
        for (int k = 0; k < NW; k++) still = still && (sh.m_del[k] == 0);
        fc.still = still; fc.meta = 0;
        // ---- per-slot meta
        int meta = 0, hdr_word = -1, new_slot = -1, lockf = 0;
        if (r.alive) {
            // back from LDS (S3 published them; dead in registers since then).  (HOME: only STAGE needs them -- ph_home_take reads
            //  them at the end of FIN, not across the observation rows)
            if constexpr (!(RES && Sh::HOME)) { r.v = sh.v[t]; r.a = sh.a[t]; }
            int coll = r.coll_fin > M_COLL_MASK ? M_COLL_MASK : r.coll_fin;
            meta = (r.meta & (M_CONTROL | M_FINISH | M_DONE | (M_LANE_MASK << M_LANE_SHIFT))) | M_ALIVE | (coll << M_COLL_SHIFT);
            if (r.cyc & 1) {
                // every member evaluates its cycle's verdict itself: records sorted by rank, python's left-to-right
                // sum(), tightest record first (ref :1493-1497); +1 for the tightest vehicle, -1 for its header
                const int len = (r.cyc >> 1) & 15, off = sh.cyc_off[r.cyc >> 9];
                double sum = 0;
#pragma unroll
                for (int q = 0; q < 10; q++) if (q < len) sum = sum + sh.u_vd[off + q];
                const int best_o = sh.lk_slot[off];
                meta |= M_LOCK;                                                    // ref :1482
                lockf = 1;
                if (sh.u_vd[off] < c.collision_thr || sum / (double)len < c.lock_mean_thr) {
                    if (best_o == t) meta |= M_LOCKA_POS;
                    else if (sh.hdr[best_o] == t) meta |= M_LOCKA_NEG;
                }
            }
            if (r.del) meta |= M_DEL;
            fc.meta = meta;
            if constexpr (!(RES && Sh::HOME)) r.vir_dis = sh.virdis[t];   // (kept in LDS since WALK / FX, like the header)
            hdr_word = pack_lanej(sh, sh.hdr[t]);
            if (mask_test(keep, t)) {
                new_slot = mask_rank<NW>(keep, t) + __builtin_popcount(sp & ((1u << r.lane) - 1u));
                if (!RES)
                    store_slot(P, env, new_slot, r, meta, hdr_word, new_slot != t, r.ctl || new_slot != t);
                else if (!still) {         // EARLY staging: these registers die here, as in the single-tick kernel
                    const int s = new_slot;
                    if (!Sh::HOME || full) sh.template stf<Sh::SF_JERK>()[s] = r.jerk;   // (HOME: persistent state only, i.e. for FLUSH)
                    if constexpr (!Sh::HOME) {
                        sh.template stf<Sh::SF_VIR_DIS>()[s] = r.vir_dis;
                        sh.template stf<Sh::SF_JERK_SUM>()[s] = r.jerk_sum; sh.template stf<Sh::SF_CLOSER_P>()[s] = r.closer_p;
                        sh.template sti<I_ID>()[s] = r.id; sh.template sti<I_SEQ>()[s] = r.seq; sh.template sti<I_VNUM>()[s] = r.vnum;
                        sh.template sti<I_COUNT>()[s] = r.count;
                    }
                    sh.template sti<I_STEP>()[s] = r.step;
                    sh.template sti<I_META>()[s] = meta; sh.template sti<I_HDR>()[s] = hdr_word;
                }
            }
        }
        // ---- new lane starts
        int n_post = 0;                                   // (RES: thread 0's business, below)
        if (!RES) n_post = mask_below<NW>(keep, sh.hd.lane_start[NL]) + __builtin_popcount(sp);
        fc.new_slot = new_slot;
        fc.ls = 0; fc.sp_slot = -1; fc.sp_id = 0; fc.sp_vnum = 0;
        if (t <= NL) {
            int ls = (RES ? (int)sh.keep_pre()[sh.hd.lane_start[t]] : mask_below<NW>(keep, sh.hd.lane_start[t])) +
                     __builtin_popcount(sp & ((1u << t) - 1u));
            if (RES) fc.ls = ls; else gh.lane_start[t] = ls;
        }
        // ---- spawned vehicles (one per lane at most), ref :395-433
        if (t < NL && ((sp >> t) & 1)) {
            int slot = mask_below<NW>(keep, sh.hd.lane_start[t + 1]) + __builtin_popcount(sp & ((1u << t) - 1u));
            const int nid = sh.hd.id_seq + __builtin_popcount(sp & ((1u << t) - 1u));
            const int nvnum = sh.hd.lane_start[t + 1] - sh.hd.lane_start[t];
            if (RES) { fc.sp_slot = slot; fc.sp_id = nid; fc.sp_vnum = nvnum; }
            else {
                Regs nv;
                nv.p = sel3(c.spawn_p, t % 3); nv.v = c.v0; nv.a = 0; nv.jerk = 0; nv.jerk_sum = 0;
                nv.vir_dis = 100; nv.closer_p = 150;
                nv.id = nid;
                nv.seq = sh.hd.veh_rec[t];
                nv.vnum = nvnum;
                nv.step = 0; nv.count = 0;
                store_slot(P, env, slot, nv, M_CONTROL | M_ALIVE | (t << M_LANE_SHIFT), -1);
                const int rec1 = sh.hd.veh_rec[t] + 1;
                gh.veh_rec[t] = rec1;
                gh.next_arr[t] = r.next_arr;
            }
        }
        if (O.obs_post && sp) {
            // zero observation rows of the spawned vehicles (ref :380, :420): one coalesced 224-B store per spawn
            // by 28 lanes instead of 28 stores by the spawning lane; the loop over the set bits is wave-uniform
            for (unsigned rem = sp; rem; rem &= rem - 1) {
                const int l = __builtin_ctz(rem);
                const int slot = mask_below<NW>(keep, sh.hd.lane_start[l + 1]) + __builtin_popcount(sp & ((1u << l) - 1u));
                if (t < OBSW) {
                    if (P.obs_f32) *env_at<CAP * OBSW>((float *)O.obs_post, env, slot * OBSW + t) = 0.0f;
                    else *env_at<CAP * OBSW>(O.obs_post, env, slot * OBSW + t) = 0.0;
                }
            }
        }
        // ---- clear the tail so stale slots never look alive
        if (!RES && t >= n_post && t < N) { P.i32[I_META][gpre] = 0; P.i32[I_ID][gpre] = -1; }   // slots >= N are clear already
        // ---- header
        if (!RES && t < NL) { gh.head_lane[t] = sh.hd.head_lane[t]; gh.head_j[t] = sh.hd.head_j[t]; }
        const int n_ctl = mask_count<NW>(sh.m_ctl);
        // (the counters of the header / env_out are thread 0's business: the other wave does not count four masks for nothing)
        fc.n_ctl = n_ctl; fc.n_post = n_post; fc.n_sp = __builtin_popcount(sp);
        if (t == 0) {
            if (RES) n_post = (int)sh.keep_pre()[sh.hd.lane_start[NL]] + __builtin_popcount(sp);
            const int n_lock = mask_count<NW>(sh.m_lead);
            const int n_fin = mask_count<NW>(sh.m_fin);
            const int n_del = mask_count<NW>(sh.m_del);
            const int n_coll = mask_count<NW>(sh.m_coll);
            double sr = 0, sj = 0;
#if PVE_DEVICE_CODE
            for (int k = 0; k < NW; k++) { sr += sh.red_reward[k]; sj += sh.red_jerk[k]; }
#else
            sr = sh.red_reward[0]; sj = sh.red_jerk[0];
#endif
            if (RES) {
                fc.n_post = n_post;
                sh.hd.passed += n_fin;                                                // ref :356
                sh.hd.passed_step_total += sh.acc_passed_steps;                       // ref :359
                sh.hd.sum_reward = sh.hd.sum_reward + sr;
                sh.hd.sum_jerk = sh.hd.sum_jerk + sj;
                sh.hd.alive_steps += N;
                sh.hd.ctl_steps += n_ctl;
                sh.hd.ticks += 1;
                sh.hd.collided += n_coll;
                sh.hd.locks += n_lock;
                sh.hd.overflow += n_over;
            }
            if (!RES) {
                gh.current_time = sh.hd.current_time;
                gh.n_alive = n_post;
                gh.id_seq = sh.hd.id_seq + __builtin_popcount(sp);
                gh.passed = sh.hd.passed + n_fin;                                     // ref :356
                gh.passed_step_total = sh.hd.passed_step_total + sh.acc_passed_steps; // ref :359
                gh.head_valid = sh.hd.head_valid;
                gh.sum_reward = sh.hd.sum_reward + sr;
                gh.sum_jerk = sh.hd.sum_jerk + sj;
                gh.alive_steps = sh.hd.alive_steps + N;
                gh.ctl_steps = sh.hd.ctl_steps + n_ctl;
                gh.ticks = sh.hd.ticks + 1;
                gh.collided = sh.hd.collided + n_coll;
                gh.locks = sh.hd.locks + n_lock;
                gh.overflow = sh.hd.overflow + n_over;
            }
            if (O.env_out) {
                int *eo = O.env_out + (size_t)env * 8;
                eo[0] = N; eo[1] = n_ctl; eo[2] = sh.acc_collisions;
                eo[3] = n_lock; eo[4] = n_del; eo[5] = n_fin; eo[6] = __builtin_popcount(sp); eo[7] = n_post;
            }
        }
        // ---- per-tick outputs, pre-compaction indexing
        if (O.flags) {
            int f = 0;
            if (r.alive) {
                f = 0x01 | (r.ctl ? 0x02 : 0) | ((r.meta & M_DONE) ? 0x04 : 0) | (r.del ? 0x08 : 0) |
                    (r.fin ? 0x10 : 0) | (lockf ? 0x20 : 0) | (r.ctl ? (r.coll_seen << 8) : 0);
            }
            *env_at<CAP>(O.flags, env, t) = f;
        }
        // per-slot outputs other than `flags` are written for the slots that held a vehicle only (`flags` = 0 marks
        // the rest): 36 B x ~43 empty slots per env are 6 MB per launch, i.e. ~0.8 us of store burst
        if (O.reward && r.alive && !r.ctl) *env_at<CAP>(O.reward, env, t) = 0.0;
        if (O.lanej && r.alive) *env_at<CAP>(O.lanej, env, t) = (r.lane << 16) | r.j;
        if (O.new_slot && r.alive) *env_at<CAP>(O.new_slot, env, t) = new_slot;
        // ---- dense mapping: what the t-th controlled vehicle (slot ds) puts out
        if (r.dctl) {
            const int sl = r.ds;
            int ns = -1;
            if (mask_test(keep, sl))
                ns = (RES ? (int)sh.keep_pre()[sl] : mask_below<NW>(keep, sl)) + __builtin_popcount(sp & ((1u << r.dlane) - 1u));
            if (O.reward) *env_at<CAP>(O.reward, env, sl) = r.reward;      // pre-compaction indexing, as above
            if (adst) *adst = ns;
            // the 6 neighbours' speed, acceleration, lane and lane start: two batches of unconditional LDS gathers on clamped
            // slots (one guarded block per neighbour = six serial round trips), shared by the neighbour ids and the row
            int xc[NNB], nln[NNB], nlj[NNB]; double nv[NNB], na[NNB];
#pragma unroll
            for (int k = 0; k < NNB; k++) {
                xc[k] = r.kr[k] < 0 ? (ZROW ? CAP : 0) : r.kr[k];   // (cell CAP: zeros)
                if constexpr (Sh::HAS_LJ) nlj[k] = sh.lj[xc[k]]; else nlj[k] = (int)sh.lane_of[xc[k]];
                nv[k] = sh.v[xc[k]]; na[k] = sh.a[xc[k]];
            }
            const double myv = sh.v[sl], mya = sh.a[sl];
#pragma unroll
            for (int k = 0; k < NNB; k++) { PVE_PIN(nlj[k]); PVE_PIN(nv[k]); PVE_PIN(na[k]); }
            if constexpr (!Sh::HAS_LJ) {                   // (general-geometry block: lane -> lane start -> j)
#pragma unroll
                for (int k = 0; k < NNB; k++) { nln[k] = nlj[k]; nlj[k] = (nln[k] << 16) | (xc[k] - sh.hd.lane_start[nln[k]]); }
            } else {
#pragma unroll
                for (int k = 0; k < NNB; k++) nln[k] = nlj[k] >> 16;
            }
            if (O.nbr) {                               // controlled vehicles only (PVE_F_CTL in flags)
                int *nb = env_at<CAP * NNB>(O.nbr, env, sl * NNB);
#pragma unroll
                for (int k = 0; k < NNB; k++) nb[k] = r.kr[k] < 0 ? -1 : nlj[k];
            }
            if (O.obs_pre || (O.obs_post && ns >= 0)) {
                // row 0 of the state, ref :1325-1337
                double row[OBSW];
                row[0] = r.dp; row[1] = myv; row[2] = mya; row[3] = (double)r.dlane;
#pragma unroll
                for (int k = 0; k < NNB; k++) {
                    // absent neighbour: WALK left kv = 0, the gathers came from the zero cell
                    const bool has = r.kr[k] >= 0;
                    row[4 + 4 * k] = (ZROW || has) ? r.kv[k] : 0.0;       // (WALK left kv = 0 for an absent neighbour)
                    row[5 + 4 * k] = (ZROW || has) ? nv[k] : 0.0;
                    row[6 + 4 * k] = (ZROW || has) ? na[k] : 0.0;
                    row[7 + 4 * k] = (ZROW || has) ? (double)nln[k] : 0.0;
                }
                if (O.obs_pre) {
                    if (P.obs_f32) {                    // (12-lane kernels: obs_pre / state_pre follow the row type)
                        float *o = env_at<CAP * OBSW>((float *)O.obs_pre, env, sl * OBSW);
#pragma unroll
                        for (int k = 0; k < OBSW; k++) o[k] = (float)row[k];
                    } else {
                        double *o = env_at<CAP * OBSW>(O.obs_pre, env, sl * OBSW);
#pragma unroll
                        for (int k = 0; k < OBSW; k++) o[k] = row[k];
                    }
                }
                if (O.obs_post && ns >= 0) {
                    if (P.obs_f32) {                    // uniform: float32 rows (half the bytes of the largest output)
                        float *o = env_at<CAP * OBSW>((float *)O.obs_post, env, ns * OBSW);
#pragma unroll
                        for (int k = 0; k < OBSW; k++) o[k] = (float)row[k];
                    } else {
                        double *o = env_at<CAP * OBSW>(O.obs_post, env, ns * OBSW);
#pragma unroll
                        for (int k = 0; k < OBSW; k++) o[k] = row[k];
                    }
                }
            }
        }
    }
    // the tick kernel's FIN
    static PVE_HD void ph_final(const PVE_AS4 Const &c, const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r)
    {
        FinCarry fc;
        ph_final<false>(c, P, P.out, env, t, sh, r, fc);
    }

    // ============================================================== k_rollout: many ticks per launch, state on the chip
    // Per tick:  [RELOAD | first tick: LOAD]  S1 .. LOCK2  FIN<RES>  | barrier A |  STAGE  | barrier B |
    // and FLUSH after the last tick.  Only the per-tick outputs (and the prefetched actions / arrival times) touch HBM.
    //
    // this tick's block of the output buffers (trajectory roll-outs: block k; else the same buffers every tick)
    template <bool TRAIN = true>
    static PVE_HD Outputs tick_outputs(const PVE_AS4 Params &P, const PVE_AS4 RolloutArgs &R, int k)
    {
        // (all pointers are loaded in one go and shifted with arithmetic: a null test + branch per pointer is a serial chain
        //  of scalar-load round trips at the top of FIN; a null pointer stays null because s = 0 or the field is unused)
        Outputs o;
        const long long s = R.trajectory ? (long long)k * P.n_envs * CAP : 0;
        // the caller's pointers as they are ...
        o.obs_post = P.out.obs_post; o.reward = P.out.reward; o.flags = P.out.flags; o.lanej = P.out.lanej; o.nbr = P.out.nbr;
        o.new_slot = P.out.new_slot; o.env_out = P.out.env_out;
        if (TRAIN) { o.obs_pre = P.out.obs_pre; o.state_pre = P.out.state_pre; o.obs_prev_post = (const double *)R.prev_rows; }
        else { o.obs_pre = nullptr; o.state_pre = nullptr; o.obs_prev_post = nullptr; }
        if (s != 0) {
            // ... shifted to block k of a trajectory roll-out (uniform branch: a roll-out that overwrites its outputs -- the bench
            // headline -- skips these ~50 scalar instructions on every tick)
            const long long se = (long long)k * P.n_envs * 8;
            const long long rowb = OBSW * (P.obs_f32 ? 4 : 8);
            if (o.obs_post) o.obs_post = (double *)((char *)o.obs_post + s * rowb);
            // training outputs: this tick's rows with pre-compaction indexing and the 7 x 28 states; the stale
            // neighbour rows are what the PREVIOUS tick stored: the caller's rows for the call's first tick, then block k - 1
            if (TRAIN) {
                if (o.obs_pre) o.obs_pre = (double *)((char *)o.obs_pre + s * rowb);
                if (o.state_pre) o.state_pre = (double *)((char *)o.state_pre + s * rowb * (NNB + 1));
                if (o.obs_post) o.obs_prev_post = (const double *)((const char *)o.obs_post - (long long)P.n_envs * CAP * rowb);
            }
            if (o.reward) o.reward += s;
            if (o.flags) o.flags += s;
            if (o.lanej) o.lanej += s;
            if (o.nbr) o.nbr += s * NNB;
            if (o.new_slot) o.new_slot += s;
            if (o.env_out) o.env_out += se;
        }
        return o;
    }
    // next tick's action of slot t: the load is issued under FX .. LOCK2 and parked in LDS at the start of FIN
    static PVE_HD void ph_prefetch_action(const PVE_AS4 Params &P, const PVE_AS4 RolloutArgs &R, int env, int t, int pool_idx,
                                          Regs &r)
    {
        r.act_nx = 0;
        if (R.source == 1 /* PVE_SRC_POOL */ && pool_idx >= 0)
            r.act_nx = *env_at<CAP>(R.pool + (size_t)pool_idx * P.n_envs * CAP, env, t);
    }
    static PVE_HD void ph_park_action(int t, Sh &sh, Regs &r) { sh.act_next[t] = r.act_nx; }   // xy32 is dead after REWARD
    // STAGE (after barrier A: nobody reads this tick's work arrays any more): every kept vehicle moves to its new slot,
    // the spawned ones are born, the header advances in place
    // HOME: the vehicle in slot t takes its carried fields out of their homes at the very end of FIN (the last reader of the old
    // arrangement) and puts them at its new slot behind barrier A
    static PVE_HD void ph_home_take(int t, Sh &sh, Regs &r, const FinCarry &fc, HomeRegs &hr)
    {
        if constexpr (Sh::HOME) {
            hr.jerk_sum = 0; hr.closer = 0; hr.vir_dis = 0; hr.id = 0; hr.sv = 0; hr.count = 0;
            r.p = sh.p[t]; r.v = sh.v[t]; r.a = sh.a[t];         // (for STAGE's LATE staging, in place; unconditional: fresh values, no merge)
            if (!fc.still && fc.new_slot >= 0) {
                hr.jerk_sum = sh.h_jerk_sum[t]; hr.closer = sh.h_closer[t]; hr.vir_dis = sh.virdis[t];
                hr.id = sh.h_id[t]; hr.sv = sh.h_sv[t]; hr.count = sh.h_count[t];
            }
        }
    }
    static PVE_HD void ph_home_put(int t, Sh &sh, const FinCarry &fc, const HomeRegs &hr)
    {
        if constexpr (Sh::HOME) {
            if (fc.new_slot >= 0) {
                const int s = fc.new_slot;
                sh.h_jerk_sum[s] = hr.jerk_sum; sh.h_closer[s] = hr.closer; sh.virdis[s] = hr.vir_dis;
                sh.h_id[s] = hr.id; sh.h_sv[s] = hr.sv; sh.h_count[s] = hr.count;
            }
        }
    }
    static PVE_HD void ph_stage(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r, const FinCarry &fc)
    {
        if (fc.new_slot >= 0) {                          // LATE staging (the rest went at the top of FIN)
            const int s = fc.new_slot;
            sh.template stf<Sh::SF_P>()[s] = r.p; sh.template stf<Sh::SF_V>()[s] = r.v; sh.template stf<Sh::SF_A>()[s] = r.a;
        }
        if (fc.sp_slot >= 0) {                           // t < NL: the vehicle lane t spawns (ref :395-433)
            const int s = fc.sp_slot;
            sh.template stf<Sh::SF_P>()[s] = sel3(c.spawn_p, t % 3); sh.template stf<Sh::SF_V>()[s] = c.v0;
            sh.template stf<Sh::SF_A>()[s] = 0; sh.template stf<Sh::SF_JERK>()[s] = 0;
            if constexpr (Sh::HOME) sh.virdis[s] = 100;
            if constexpr (!Sh::HOME) {
                sh.template stf<Sh::SF_VIR_DIS>()[s] = 100;
                sh.template stf<Sh::SF_JERK_SUM>()[s] = 0; sh.template stf<Sh::SF_CLOSER_P>()[s] = 150;
                sh.template sti<I_ID>()[s] = fc.sp_id; sh.template sti<I_SEQ>()[s] = sh.hd.veh_rec[t];
                sh.template sti<I_VNUM>()[s] = fc.sp_vnum; sh.template sti<I_COUNT>()[s] = 0;
            }
            sh.template sti<I_STEP>()[s] = 0;
            sh.template sti<I_META>()[s] = M_CONTROL | M_ALIVE | (t << M_LANE_SHIFT); sh.template sti<I_HDR>()[s] = -1;
            if constexpr (Sh::HOME) {                    // (behind barrier A: the old occupant's values were taken at the end of FIN)
                sh.h_jerk_sum[s] = 0; sh.h_closer[s] = 150; sh.h_id[s] = fc.sp_id;
                sh.h_sv[s] = (sh.hd.veh_rec[t] << 8) | (fc.sp_vnum & 0xFF); sh.h_count[s] = 0;
            }
            sh.hd.veh_rec[t] += 1;
            sh.hd.next_arr[t] = r.next_arr;
        }
        if (t <= NL) sh.hd.lane_start[t] = fc.ls;
        ph_stage_header(t, sh, fc);
    }
    // n_alive and id_seq: what the other threads still read during FIN (a still tick changes neither, but the call is harmless)
    static PVE_HD void ph_stage_header(int t, Sh &sh, const FinCarry &fc)
    {
        if (t == 0) {                                     // (the accumulators went in FIN)
            sh.hd.n_alive = fc.n_post;
            sh.hd.id_seq += fc.n_sp;
        }
    }
    // a still tick: the vehicle stays in the registers, only the flags word and the next action change hands
    static PVE_HD void ph_carry_over(int t, Sh &sh, Regs &r, const FinCarry &fc)
    {
        r.meta = fc.meta;
        if constexpr (!Sh::HOME) r.act = sh.act_next[t];  // (parked by this very thread at the top of FIN; HOME: S1 reads it there)
    }
    // RELOAD (after barrier B): slot t's vehicle from the staging arrays, its action from the prefetch
    static PVE_HD void ph_reload(int t, Sh &sh, Regs &r)
    {
        const int N = sh.hd.n_alive;
        r.alive = t < N;
        r.jerk = 0;
        r.p = r.v = r.a = r.jerk_sum = r.vir_dis = r.closer_p = 0;
        r.id = r.seq = r.vnum = r.step = r.count = r.meta = 0;
        if (t < N) {
            if constexpr (!Sh::HOME) {
                r.p = sh.template stf<Sh::SF_P>()[t]; r.v = sh.template stf<Sh::SF_V>()[t]; r.a = sh.template stf<Sh::SF_A>()[t];
                r.vir_dis = sh.template stf<Sh::SF_VIR_DIS>()[t];
                r.jerk_sum = sh.template stf<Sh::SF_JERK_SUM>()[t];
                r.closer_p = sh.template stf<Sh::SF_CLOSER_P>()[t];
                r.id = sh.template sti<I_ID>()[t]; r.seq = sh.template sti<I_SEQ>()[t]; r.vnum = sh.template sti<I_VNUM>()[t];
                r.count = sh.template sti<I_COUNT>()[t];
            }
            r.step = sh.template sti<I_STEP>()[t]; r.meta = sh.template sti<I_META>()[t];
        }
        if constexpr (!Sh::HOME) r.act = sh.act_next[t];
    }
    // work-array initialisation of a resident tick (what LOAD does besides loading), after the barrier behind RELOAD
    static PVE_HD void ph_tick_init(const PVE_AS4 Const &c, int t, Sh &sh, Regs &r)
    {
        sh.cnt[t] = 0;
        if (t == 0) {
            sh.acc_passed_steps = 0; sh.acc_collisions = 0; sh.lead_n = 0;
            sh.hd.current_time = sh.hd.current_time + c.deltaT;                   // ref :223 (repeated +=, not tick*dt)
        }
    }
    // FLUSH (after the last tick's barrier B): staging area + header -> HBM
    static PVE_HD void ph_flush(const PVE_AS4 Params &P, int env, int t, Sh &sh)
    {
        const int N = sh.hd.n_alive;
        const size_t g = (size_t)env * CAP + t;
        if (t < N) {
            P.f64[F_P][g] = sh.template stf<Sh::SF_P>()[t]; P.f64[F_V][g] = sh.template stf<Sh::SF_V>()[t];
            P.f64[F_A][g] = sh.template stf<Sh::SF_A>()[t]; P.f64[F_JERK][g] = sh.template stf<Sh::SF_JERK>()[t];
            if constexpr (!Sh::HOME) P.f64[F_VIR_DIS][g] = sh.template stf<Sh::SF_VIR_DIS>()[t];
            if constexpr (Sh::HOME) {
                P.f64[F_VIR_DIS][g] = sh.virdis[t];
                P.f64[F_JERK_SUM][g] = sh.h_jerk_sum[t]; P.f64[F_CLOSER_P][g] = sh.h_closer[t];
                const int sv = sh.h_sv[t];
                P.i32[I_ID][g] = sh.h_id[t]; P.i32[I_SEQ][g] = sv >> 8; P.i32[I_VNUM][g] = sv & 0xFF; P.i32[I_COUNT][g] = sh.h_count[t];
            } else {
            P.f64[F_JERK_SUM][g] = sh.template stf<Sh::SF_JERK_SUM>()[t];
            P.f64[F_CLOSER_P][g] = sh.template stf<Sh::SF_CLOSER_P>()[t];
            P.i32[I_ID][g] = sh.template sti<I_ID>()[t]; P.i32[I_SEQ][g] = sh.template sti<I_SEQ>()[t];
            P.i32[I_VNUM][g] = sh.template sti<I_VNUM>()[t]; P.i32[I_COUNT][g] = sh.template sti<I_COUNT>()[t];
            }
            P.i32[I_STEP][g] = sh.template sti<I_STEP>()[t]; P.i32[I_META][g] = sh.template sti<I_META>()[t];
            P.i32[I_HDR][g] = sh.template sti<I_HDR>()[t];
        } else { P.i32[I_META][g] = 0; P.i32[I_ID][g] = -1; }                     // stale slots never look alive
        int *dst = (int *)&P.headers[env];
        const int *src = (const int *)&sh.hd;
        for (int w = t; w < (int)(sizeof(EnvHeader) / 4); w += CAP) dst[w] = src[w];
    }

    // ============================================================== STATE: full 7x28 state (training outputs)
    // rows 1..6 = the neighbour's own latest row 0 (ref :1332): already recomputed this tick if the
    // neighbour precedes us in (lane, j) order ("fresh", read back from obs_pre written in FIN), else
    // the row it stored last tick ("stale", obs_prev_post at the same slot); zeros when absent (ref :1335).
    // Runs after a workgroup barrier + fence so that obs_pre rows of the other threads are visible.
    // COH: the stale rows may have been stored by ANOTHER workgroup of this launch (persistent roll-out, first tick of an
    // item: block k - 1 of the trajectory belongs to the previous item of the intersection): coherent loads
    template <class ROW, bool COH, class OutT>
    static PVE_HD void state_rows(const OutT &O, size_t base, const Regs &r)
    {
        const int sl = r.ds;
        ROW *dst = (ROW *)O.state_pre + (base + sl) * (size_t)((NNB + 1) * OBSW);
        const ROW *pre = (const ROW *)O.obs_pre, *prev = (const ROW *)O.obs_prev_post;
        const ROW *srcs[NNB + 1];
        unsigned coh = 0;
        srcs[0] = pre + (base + sl) * OBSW;
#pragma unroll
        for (int q = 0; q < NNB; q++) {
            const int x = r.kr[q];
            const bool fresh = x < sl;                    // the neighbour precedes us in order: its row of THIS tick (ref :1332)
            srcs[q + 1] = x < 0 ? (const ROW *)nullptr : ((fresh ? pre : prev) + (base + x) * OBSW);
            if (COH && x >= 0 && !fresh) coh |= 2u << q;
        }
        gather_state<ROW>(srcs, coh, dst);
    }
    template <bool COH = false, class OutT>
    static PVE_HD void ph_state(const PVE_AS4 Params &P, const OutT &O, int env, int t, Sh &sh, Regs &r)
    {
        if (!O.state_pre || !r.dctl) return;              // (dense mapping: the rows of the vehicle in slot ds)
        if (P.obs_f32) state_rows<float, COH>(O, (size_t)env * CAP, r);
        else state_rows<double, COH>(O, (size_t)env * CAP, r);
    }
    static PVE_HD void ph_state(const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r) { ph_state(P, P.out, env, t, sh, r); }

    // ---- the same states written COOPERATIVELY (round 6; the dense-mapping kernels): a vehicle's state is 7 x 112 B (float32) or
    // 7 x 224 B (float64) of CONTIGUOUS output -- one wave instruction (two for float64 rows) writes all of it, lane g its 16-byte
    // piece g, instead of every dense thread writing its own state in 49 / 98 pieces 784 / 1 568 B apart from its neighbour lanes'
    // (the L2 had to merge them into lines).  PUBLISH: dense thread d files where its 7 rows come from -- 8 bytes in the vehicle's
    // dead virdis[] cell pair: byte q = source slot | fresh << 7 (fresh: the row of THIS tick, obs_pre; else the one stored last
    // tick, ref :1332; an absent neighbour names the vehicle's own slot, which is never its own neighbour), byte 7 = its slot;
    // COOP (behind a barrier): the waves deal the vehicles among themselves, U at a time, loads back to back, then the stores.
    template <class OutT>
    static PVE_HD void ph_state_publish(const OutT &O, int t, Sh &sh, const Regs &r)
    {
        static_assert(!Sh::HOME, "the HOME block keeps vir_dis in virdis[]: no training outputs through it");
        if (!O.state_pre || !r.dctl) return;
        uint8_t *desc = (uint8_t *)sh.virdis + 8 * t;
        const int sl = r.ds;
        desc[0] = (uint8_t)(sl | 128);
#pragma unroll
        for (int q = 0; q < NNB; q++) {
            const int x = r.kr[q];
            desc[q + 1] = (uint8_t)(x < 0 ? sl : (x | (x < sl ? 128 : 0)));
        }
        desc[7] = (uint8_t)sl;
    }
    template <class ROW, bool COH, class OutT>
    static PVE_HD void state_coop(const OutT &O, size_t base, int t, Sh &sh)
    {
        constexpr int RB = (int)sizeof(ROW) * OBSW, PPR = RB / 16, PPV = 7 * PPR, NP = (PPV + 48) / 49;   // 112 / 224 B, 7 / 14, 49 / 98, 1 / 2
        const int lane = t & 63, w = t >> 6;
        const int n = mask_count<NW>(sh.m_ctl);
        const uint8_t *desc = (const uint8_t *)sh.virdis;
        const char *pre = (const char *)O.obs_pre + base * RB, *prev = (const char *)O.obs_prev_post + base * RB;
        char *dstb = (char *)O.state_pre + base * (size_t)(7 * RB);
#if PVE_DEVICE_CODE
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr int U = 4;                              // vehicles per batch and wave
        if (lane >= 49) return;
        for (int d0 = w; d0 < n; d0 += NW * U) {
            v4f q[U][NP]; unsigned off[U][NP]; bool zero[U][NP];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int d = d0 + u * NW, dc = d < n ? d : w;       // (clamped: every lane of the batch issues its loads)
                const int sl = desc[8 * dc + 7];
#pragma unroll
                for (int h = 0; h < NP; h++) {
                    const int g = lane + 49 * h, qq = g / PPR, i = g - qq * PPR;
                    const int b = desc[8 * dc + qq];
                    zero[u][h] = qq > 0 && (b & 127) == sl;
                    const char *src = ((b & 128) ? pre : prev) + (unsigned)((b & 127) * RB + 16 * i);
                    off[u][h] = (unsigned)(sl * 7 * RB + 16 * g);
                    if constexpr (COH) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(q[u][h]) : "v"(src) : "memory");
                    else q[u][h] = *(const v4f *)src;
                }
            }
            if constexpr (COH) {                          // (the asm loads above: the compiler does not count them)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int h = 0; h < NP; h++) asm volatile("" : "+v"(q[u][h]));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (d0 + u * NW >= n) break;
#pragma unroll
                for (int h = 0; h < NP; h++) *(v4f *)(dstb + off[u][h]) = zero[u][h] ? v4f{0.f, 0.f, 0.f, 0.f} : q[u][h];
            }
        }
#else
        if (lane >= 49) return;
        for (int d = w; d < n; d += NW) {
            const int sl = desc[8 * d + 7];
            for (int g = lane; g < PPV; g += 49) {
                const int qq = g / PPR, i = g - qq * PPR, b = desc[8 * d + qq];
                char *dp = dstb + (size_t)sl * 7 * RB + 16 * g;
                if (qq > 0 && (b & 127) == sl) memset(dp, 0, 16);
                else memcpy(dp, ((b & 128) ? pre : prev) + (size_t)(b & 127) * RB + 16 * i, 16);
            }
        }
#endif
    }
    template <bool COH = false, class OutT>
    static PVE_HD void ph_state_coop(const PVE_AS4 Params &P, const OutT &O, int env, int t, Sh &sh)
    {
        if (!O.state_pre) return;
        if (P.obs_f32) state_coop<float, COH>(O, (size_t)env * CAP, t, sh);
        else state_coop<double, COH>(O, (size_t)env * CAP, t, sh);
    }

    // ============================================================== COMPACT (delete_vehicle only)
    static PVE_HD void ph_c_load(const PVE_AS4 Params &P, int env, int t, Sh &sh, CRegs &r)
    {
        const EnvHeader &gh = P.headers[env];
        {
            const int *src = (const int *)&gh;
            int *dst = (int *)&sh.hd;
            for (int w = t; w < (int)(sizeof(EnvHeader) / 4); w += CAP) dst[w] = src[w];
        }
        const int N = gh.n_alive;
        const size_t g = (size_t)env * CAP + t;
        r.alive = t < N;
        r.meta = 0;
        if (r.alive) {
            r.p = P.f64[F_P][g]; r.v = P.f64[F_V][g]; r.a = P.f64[F_A][g]; r.jerk = P.f64[F_JERK][g];
            r.jerk_sum = P.f64[F_JERK_SUM][g]; r.vir_dis = P.f64[F_VIR_DIS][g]; r.closer_p = P.f64[F_CLOSER_P][g];
            r.id = P.i32[I_ID][g]; r.seq = P.i32[I_SEQ][g]; r.vnum = P.i32[I_VNUM][g];
            r.step = P.i32[I_STEP][g]; r.count = P.i32[I_COUNT][g]; r.meta = P.i32[I_META][g];
            r.hdr_word = P.i32[I_HDR][g];
            if (P.out.obs_post) {
                const double *o = P.out.obs_post + g * OBSW;
                for (int k = 0; k < OBSW; k++) r.obsrow[k] = o[k];
            }
        }
        vote<NW>(sh.m_keep, t, r.alive && !(r.meta & M_DEL));
    }
    static PVE_HD void ph_c_store(const PVE_AS4 Params &P, int env, int t, Sh &sh, CRegs &r)
    {
        EnvHeader &gh = P.headers[env];
        const int n_post = mask_count<NW>(sh.m_keep);
        if (r.alive && !(r.meta & M_DEL)) {
            int ns = mask_below<NW>(sh.m_keep, t);
            size_t g = (size_t)env * CAP + ns;
            if (ns != t) {                            // a vehicle that keeps its slot keeps everything
                store_slot(P, env, ns, r, r.meta, r.hdr_word);
                if (P.out.obs_post) {
                    double *o = P.out.obs_post + g * OBSW;
                    for (int k = 0; k < OBSW; k++) o[k] = r.obsrow[k];
                }
            }
        }
        if (t <= NL) gh.lane_start[t] = mask_below<NW>(sh.m_keep, sh.hd.lane_start[t]);
        if (t >= n_post && r.alive) { P.i32[I_META][(size_t)env * CAP + t] = 0; P.i32[I_ID][(size_t)env * CAP + t] = -1; }
        if (t == 0) gh.n_alive = n_post;
    }
};

// ================================================================== reset / warm-up, ref :196-220
// One thread per environment: advance the clock (repeated += deltaT) and spawn (lane order) until
// at least one vehicle exists.  cap_ticks bounds the loop for streams that never deliver.
template <int CAP>
PVE_HD void reset_env(const PVE_AS4 Const &c, const PVE_AS4 Params &P, int env, int cap_ticks)
{
    EnvHeader h;
    {
        int *z = (int *)&h;
        for (int w = 0; w < (int)(sizeof(EnvHeader) / 4); w++) z[w] = 0;
    }
    for (int l = 0; l < ND; l++) { h.head_lane[l] = -1; h.head_j[l] = -1; }
    const double *arr = P.arrivals + (size_t)env * P.arr_env_stride;
    int n = 0;
    int lane_of[NL];
    for (int it = 0; it < cap_ticks && n == 0; it++) {
        h.current_time += c.deltaT;
        for (int l = 0; l < NL; l++) {
            if (h.veh_rec[l] < P.rows && h.current_time >= arr[(size_t)h.veh_rec[l] * NL + l] && n < CAP) {
                lane_of[n] = l;
                n++;
                h.veh_rec[l] += 1;
            }
        }
    }
    for (int s = 0; s < CAP; s++) {
        size_t g = (size_t)env * CAP + s;
        if (s < n) {
            int l = lane_of[s];
            P.f64[F_P][g] = c.spawn_p[l % 3]; P.f64[F_V][g] = c.v0; P.f64[F_A][g] = 0; P.f64[F_JERK][g] = 0;
            P.f64[F_JERK_SUM][g] = 0; P.f64[F_VIR_DIS][g] = 100; P.f64[F_CLOSER_P][g] = 150;
            P.i32[I_ID][g] = s; P.i32[I_SEQ][g] = 0; P.i32[I_VNUM][g] = 0; P.i32[I_STEP][g] = 0;
            P.i32[I_COUNT][g] = 0; P.i32[I_META][g] = M_CONTROL | M_ALIVE | (l << M_LANE_SHIFT); P.i32[I_HDR][g] = -1;
        } else {
            P.f64[F_P][g] = 0; P.f64[F_V][g] = 0; P.f64[F_A][g] = 0; P.f64[F_JERK][g] = 0;
            P.f64[F_JERK_SUM][g] = 0; P.f64[F_VIR_DIS][g] = 0; P.f64[F_CLOSER_P][g] = 0;
            P.i32[I_ID][g] = -1; P.i32[I_SEQ][g] = 0; P.i32[I_VNUM][g] = 0; P.i32[I_STEP][g] = 0;
            P.i32[I_COUNT][g] = 0; P.i32[I_META][g] = 0; P.i32[I_HDR][g] = -1;
        }
    }
    {
        int s = 0;
        for (int l = 0; l <= NL; l++) {
            h.lane_start[l] = s;
            while (s < n && l < NL && lane_of[s] == l) s++;
        }
    }
    for (int l = 0; l < NL; l++)
        h.next_arr[l] = (h.veh_rec[l] < P.rows) ? arr[(size_t)h.veh_rec[l] * NL + l] : INFINITY;
    h.n_alive = n;
    h.id_seq = n;
    P.headers[env] = h;
}

}  // namespace pve
