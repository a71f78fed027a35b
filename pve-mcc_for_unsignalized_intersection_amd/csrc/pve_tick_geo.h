// pve_tick_geo.h -- the environment tick for the 4- and 8-lane layouts (lane_num = 12 is
// accepted too and must reproduce the fast path of pve_tick_core.h bit for bit -- the tests use that).
//
// What is different from the 12-lane layout (ref = traffic_interaction_scene.py):
//   - a physical lane carries several movements: a vehicle has an `intention` (ref :382-394) and a route
//     `direction[lane][intention]` (ref :88-93, :135-144); the virtual-lane lists are per ROUTE (12 or 16 of them),
//     built from the vehicles of the same route, the vehicles of the same lane with another intention that have
//     not reached the box yet (ref :250-257) and the routes of lane2lane[route] (<= 7, ref :74-87, :107-124);
//   - scene_update walks lane -> intention -> j (ref :233-275): the processing order that decides collision
//     visibility (ref :333-340), the reward[-1] overrides (ref :346, :357) and fresh/stale neighbour rows (ref :1332)
//     is (lane, intention, j), not the slot order (lane, j);
//   - 4-lane only: while the left-turn vehicles of one route are processed, the list entries of the opposing
//     left-turn route are re-written by every one of them in turn (ref :1301-1319, stored back at ref :286-287);
//   - step() tests the head of virtual_lane_4[LANE index] (ref :1517) and forces aM on lanes 2, 5, 8, 11 (ref :1519)
//     whatever the layout: both quirks are part of the behaviour and are kept.
//
// Slots stay sorted by (lane, j); one workgroup = one intersection, thread t = slot t in every phase (the dense
// mapping of the 12-lane kernels is not used here).  The step chain, the dead-lock scan (ph_lock_slot) and the
// compaction are the phases of pve_tick_core.h (Tick<CAP, SharedGeo<CAP>>); the neighbour search works on per-route
// lists sorted by counting sort and the shared window walk (Tick::walk_window), with the membership scan over the
// controlled vehicles as the fallback when the entry pool overflows.
#pragma once
#include "pve_tick_core.h"

namespace pve {

#ifndef PVE_GEO_ZROW
#define PVE_GEO_ZROW 1
#endif

template <int CAP> struct SharedGeo {
    static constexpr int NW = CAP / 64;
    EnvHeader hd;
    double p[CAP], v[CAP + 1], a[CAP + 1];  // cell CAP of v / a / lane_of / route_of: a vehicle that is not there (zeros, cf. Shared)
    double virdis[CAP];
    double red_reward[NW], red_jerk[NW];
    // Virtual-lane lists (ref :240-270), one per route: list d = [lbase[d], lbase[d] + fill[d]) of the entry pool; an
    // entry = (build-time virtual distance, slot).  PAIRS: the (controlled vehicle, list its route can appear in)
    // pairs of the intersection are dealt evenly to the threads; a pair that is a member claims the next entry of its
    // list; RANK: counting sort of every list by (distance, slot) = the reference's stable sort (ref :271); WALK:
    // predecessor and 6 nearest from the window around the own position (the 12-lane kernel's walk_window).  Every ego
    // reads only its own list instead of testing every controlled vehicle of the intersection.  Segment capacities are
    // upper bounds from the per-route counts; if they do not fit the pool (pool_ok = 0: very dense traffic) WALK falls
    // back to the membership scan, with identical results.
    static constexpr int PE = 4 * CAP;
    static constexpr bool DIRECT = false;   // (walk_window: sorted lists are index arrays into the entries)
    static constexpr bool HOME = false;     // (no LDS homes for the carried fields: Shared<128, false, true> only)
    static constexpr bool HAS_LJ = false;   // (no room for the lane << 16 | j words at 10 workgroups per CU)
    static constexpr bool DENSE = false;    // every phase per slot (the 12-lane kernels run the controlled-vehicle phases on a dense mapping)
    static constexpr bool PIN_READS = false;  // (walk_window's pinned reads: no gain here, measured)
    union {                                 // p1 / v1 die at the barrier after S3, the lists are born after it
        struct { double p1[CAP], v1[CAP]; };
        double u_vd[PE + 8];                // entry pool (+ the pad RANK's tail round may read); LOCK2 re-uses [0, CAP) as the dead-lock scratch (records by rank)
    };
    double s_vd[1];
    alignas(8) unsigned s_idx[PE];          // sorted position -> entry | tick tag << 16 (cf. Shared<128>, Tick::ph_rank)
    alignas(8) uint8_t u_slot[PE], u_list[PE];
    PVE_HD uint8_t *chain() { return u_slot; }   // FX -> LOCK: the virtual headers as a byte chain with a sentinel (Tick::ph_lock_slot); u_slot is free then
    alignas(8) uint8_t s_slot[PE];          // (k_rollout_geo's staging storage: sti<4>)
    // k_rollout_geo keeps the state on the chip between two ticks (cf. Shared<CAP>): every persistent field of every vehicle
    // moves to its new slot through storage that is dead by FIN -- the entry pool beyond the dead-lock records, virdis, the
    // sorted-list arrays, cnt, ord / slot_at (EARLY, at the top of FIN); p into u_vd[0 .. CAP), v and a in place (LATE,
    // behind barrier A)
    enum { SF_JERK = 0, SF_JERK_SUM, SF_VIR_DIS, SF_CLOSER_P, SF_P, SF_V, SF_A };
    template <int K> PVE_HD double *stf()
    {
        return K < 3 ? u_vd + (K + 1) * CAP : (K == SF_CLOSER_P ? virdis : (K == SF_P ? u_vd : (K == SF_V ? v : a)));
    }
    template <int K> PVE_HD int *sti()   // K = I_ID .. I_HDR
    {
        return K < 2 ? (int *)s_idx + K * CAP
                     : (K == 2 ? (int *)u_slot : (K == 3 ? (int *)u_list : (K == 4 ? (int *)s_slot : (K == 5 ? cnt : (int *)ord))));
    }
    int16_t mypos[CAP];                     // sorted position of every controlled vehicle's own entry in its route's list
    int rc[ND], rfill[ND], fill[ND], cnt2[ND], pool_ok;   // controlled vehicles per route, claimed so far; entries filed per list;
                                            // exact member counts (only when the upper bounds overflow the pool)
    // 4-lane layout: the entries of the OPPOSING left-turn route in the list of a left-turn route are re-written ego by ego (ref
    // :1301-1319), so they have no place in the sorted list: they are filed from the BACK of the list's capacity (ofill of
    // them: entry lbase[d + 1] - 1 - k), unsorted, and every ego of the route merges those few into what the window walk over
    // the sorted rest returns (round 3 visited every member of the list per ego: 23 % of the 4-lane tick)
    int ofill[ND];
    int16_t lnew[ND];
    int16_t lbase[ND + 1];                  // list d owns [lbase[d], lbase[d] + fill[d]) (capacity: the next lbase)
    int16_t rbase[ND + 1], pbase[ND + 1];   // prefix of rc (route-sorted controlled vehicles) and of rc * nl (pairs)
    uint8_t ctl_by_route[CAP];
    // 4-lane far-conflict fix-up (ref :1301-1319) as a TABLE: row = a controlled vehicle x of a left-turn route (rk[x] = its
    // rank among the controlled vehicles of its route), column i = the value of x's entry in the opposing left-turn route's
    // list after the first i + 1 egos of that route have re-written it (the adjustments compound, ref :286-287).  Computed
    // once per (entry, ego) by the entry's own thread instead of being replayed by every ego for every member it looks at;
    // lives in the unused tail of the entry pool, u_vd[tstart ..) (fix_ok = 0: no room, the egos replay as before)
    uint8_t rk[CAP];
    int16_t tbase[5], tstart;
    int fix_ok;
    alignas(8) GeoTab tab;           // copy of GeoConst::tab
    int cnt[CAP];
    int acc_passed_steps, acc_collisions, lead_n, emu_scan, emu_scan2, emu_scan3;
    alignas(8) int16_t hdr[CAP], cyc_off[CAP], ord[CAP], slot_at[CAP];
    uint8_t bb[CAP], rew_ovr[CAP], lane_of[CAP + 1], route_of[CAP + 1], intent_of[CAP], lk_slot[CAP];
    u64 m_alive[NW], m_ctl[NW], m_del[NW], m_fin[NW], m_ctlnow[NW], m_lead[NW], m_coll[NW], m_keep[NW], m_spawn[NW];
    u64 m_int[3][NW];                // alive slots by intention
    u64 m_ctl_ord[NW];               // "controlled" flags in processing order
};

// XY position (ref :896-1249): every (lane, intention) branch of the reference is one of three canonical paths
// (approach along +x below the axis) turned by a quarter-turn count; sign changes are exact.
PVE_HD void geo_xy(const PVE_AS4 GeoConst &g, double p, int lane, int m, double &X, double &Y)
{
    if (g.lane_num == 12) { get_xy(g.base, p, lane, X, Y); return; }
    const double cw = g.base.cw, H = g.H;
    const double yo = (g.lane_num == 8 && (lane & 1)) ? 3 * cw : cw;
    const double Lb = sel2(g.base.inbox[0], g.base.inbox[2], m == 2);
    const double rl = (double)g.RL;
    const bool before = p > Lb, inside = !before && p > 0;
    double sn, cs;
    sincos_q1((inside && m != 1) ? ((m == 0) ? p / (rl * cw) : p / cw) : 0.0, sn, cs);
    double x, y;
    if (m == 1) { x = -1 * p + H; y = -yo; }
    else if (m == 0) {
        const double dy = sn * rl * cw, dx = cs * rl * cw;
        x = before ? -(p - Lb + H) : (inside ? (dx - H) : cw);
        y = before ? -cw : (inside ? (H - dy) : (-1 * p + H));
    } else {
        const double dy = sn * cw, dx = cs * cw;
        x = before ? -(p - Lb + H) : (inside ? -(H - dx) : -yo);
        y = before ? -yo : (inside ? -(H - dy) : -(-1 * p + H));
    }
    const int q = (int)((g.turn_pk >> (4 * lane)) & 15ull);
    X = (q == 0) ? x : (q == 1) ? -y : (q == 2) ? -x : y;
    Y = (q == 0) ? y : (q == 1) ? x : (q == 2) ? -y : -x;
}

// single-precision twin of geo_xy for the collision PRE-FILTER only (never for a decision): |error| < 1e-3 m
PVE_HD void geo_xy_f32(const PVE_AS4 GeoConst &g, double pd, int lane, int m, float &X, float &Y)
{
    if (g.lane_num == 12) { get_xy_f32(g.base, pd, lane, X, Y); return; }
    const float cw = (float)g.base.cw, H = (float)g.H, p = (float)pd;
    const float yo = (g.lane_num == 8 && (lane & 1)) ? 3.f * cw : cw;
    const float Lb = (float)sel2(g.base.inbox[0], g.base.inbox[2], m == 2);
    const float rl = (float)g.RL;
    const bool before = p > Lb, inside = !before && p > 0.f;
    const float ra = (inside && m != 1) ? p * frcp((m == 0) ? rl * cw : cw) : 0.f;
    const bool fold = ra > 0.78539816f;
    const float y0 = fold ? (1.5707964f - ra) : ra, z = y0 * y0;
    const float s1 = y0 + y0 * z * (-1.6666667e-1f + z * (8.3333338e-3f + z * (-1.9841270e-4f + z * 2.7557319e-6f)));
    const float c1 = 1.f + z * (-0.5f + z * (4.1666668e-2f + z * (-1.3888889e-3f + z * 2.4801587e-5f)));
    const float sn = fold ? c1 : s1, cs = fold ? s1 : c1;
    float x, y;
    if (m == 1) { x = -1.f * p + H; y = -yo; }
    else if (m == 0) {
        const float dy = sn * rl * cw, dx = cs * rl * cw;
        x = before ? -(p - Lb + H) : (inside ? (dx - H) : cw);
        y = before ? -cw : (inside ? (H - dy) : (-1.f * p + H));
    } else {
        const float dy = sn * cw, dx = cs * cw;
        x = before ? -(p - Lb + H) : (inside ? -(H - dx) : -yo);
        y = before ? -yo : (inside ? -(H - dy) : -(-1.f * p + H));
    }
    const int q = (int)((g.turn_pk >> (4 * lane)) & 15ull);
    X = (q == 0) ? x : (q == 1) ? -y : (q == 2) ? -x : y;
    Y = (q == 0) ? y : (q == 1) ? x : (q == 2) ? -y : -x;
}

template <int CAP> struct TickGeo {
    typedef SharedGeo<CAP> Sh;
    typedef Tick<CAP, Sh> Base;
    static constexpr int NW = CAP / 64;

    // ============================================================== L: load
    // COH (persistent roll-out): the state may have been stored by another workgroup of this launch -> coherent loads
    // (Tick::ph_load); act0: the first tick's actions of the item ([n_envs][CAP] or null) instead of P.actions; COHA: the
    // actions too (the persistent closed loop hands them from one item to the next)
    template <bool COH = false, bool COHA = false>
    static PVE_HD void ph_load(const PVE_AS4 GeoConst &g, const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r,
                               const double *act0 = nullptr, bool use_act0 = false)
    {
        // the lookup tables first (they are written to LDS, so their loads must be waited for before the first barrier;
        // everything issued behind them may still be in flight then)
        {
            const int *src = (const int *)&g.tab;
            int *dst = (int *)&sh.tab;
            for (int w = t; w < (int)(sizeof(GeoTab) / 4); w += CAP) dst[w] = src[w];
        }
        const EnvHeader &gh = P.headers[env];
        {
            const int *src = (const int *)&gh;
            int *dst = (int *)&sh.hd;
            for (int w = t + 2; w < (int)(sizeof(EnvHeader) / 4); w += CAP) dst[w] = gld<COH>(src + w);
        }
        if (t == 0) sh.hd.current_time = gld<COH>(&gh.current_time) + g.base.deltaT;       // ref :223
        const int N = gld<COH>(&gh.n_alive);
        r.alive = t < N;
        r.jerk = 0;
        r.act = 0;
        r.p = r.v = r.a = r.jerk_sum = r.vir_dis = r.closer_p = 0;
        r.id = r.seq = r.vnum = r.step = r.count = r.meta = 0;
        const double *acts = use_act0 ? act0 : P.actions;
        if (t < 64 || t < N) {                    // first wave unconditionally (no dependence on n_alive), later waves live slots only
            if (acts) r.act = gld<COHA>(env_at<CAP>(acts, env, t));
            r.p = gld<COH>(env_at<CAP>(P.f64[F_P], env, t)); r.v = gld<COH>(env_at<CAP>(P.f64[F_V], env, t)); r.a = gld<COH>(env_at<CAP>(P.f64[F_A], env, t));
            r.meta = gld<COH>(env_at<CAP>(P.i32[I_META], env, t)); r.step = gld<COH>(env_at<CAP>(P.i32[I_STEP], env, t));
            r.seq = gld<COH>(env_at<CAP>(P.i32[I_SEQ], env, t)); r.vnum = gld<COH>(env_at<CAP>(P.i32[I_VNUM], env, t)); r.count = gld<COH>(env_at<CAP>(P.i32[I_COUNT], env, t));
        }
        sh.cnt[t] = 0; sh.rew_ovr[t] = 0; sh.hdr[t] = -1;
        if (t == 0) {
            sh.acc_passed_steps = 0; sh.acc_collisions = 0; sh.lead_n = 0;
            sh.v[CAP] = 0; sh.a[CAP] = 0; sh.lane_of[CAP] = 0; sh.route_of[CAP] = 0;      // the zero cell (FIN's absent neighbours)
        }
        if (t < ND) { sh.rc[t] = 0; sh.rfill[t] = 0; sh.fill[t] = 0; sh.cnt2[t] = 0; sh.ofill[t] = 0; }
    }

    // jerk_sum, closer_p and vir_dis are first touched in WALK / REWARD (ref :302, :321, :1348), the id only by FIN: their
    // loads are issued behind the list phases, not with the rest of the state -- seven registers less to carry through
    // the register peak of PAIRS (four were spilled; which fields move was settled by the compiler's spill count)
    template <bool COH = false>
    static PVE_HD void ph_load_late(const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r)
    {
        if (t < 64 || t < sh.hd.n_alive) {
            r.jerk_sum = gld<COH>(env_at<CAP>(P.f64[F_JERK_SUM], env, t)); r.closer_p = gld<COH>(env_at<CAP>(P.f64[F_CLOSER_P], env, t));
            r.vir_dis = gld<COH>(env_at<CAP>(P.f64[F_VIR_DIS], env, t));
            r.id = gld<COH>(env_at<CAP>(P.i32[I_ID], env, t));            // (FIN only, for a vehicle that moves)
        }
    }

    // ============================================================== S1..S3: step() -- the 12-lane phases; the head
    // test indexes the lists by LANE number and lanes 2, 5, 8, 11 always get aM (ref :1517-1520), as there.
    static PVE_HD void ph_step1(const PVE_AS4 GeoConst &g, const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r)
    {
        Base::ph_step1(g.base, P, env, t, sh, r);
        r.intent = 0; r.route = 0;
        if (r.alive) {
            r.intent = (g.lane_num == 12) ? (r.lane % 3) : ((r.meta >> M_INT_SHIFT) & M_INT_MASK);
            unsigned long long d0 = g.dir_pk[0], d1 = g.dir_pk[1], d2 = g.dir_pk[2];
#if PVE_DEVICE_CODE
            asm volatile("" : "+s"(d0), "+s"(d1), "+s"(d2));      // scalar loads + select (not one load from a selected address)
#endif
            const unsigned long long dpk = (r.intent == 0) ? d0 : ((r.intent == 1) ? d1 : d2);
            r.route = (int)((dpk >> (5 * r.lane)) & 31ull) - 1;                     // direction[lane][intention]
            sh.route_of[t] = (uint8_t)r.route;
            sh.intent_of[t] = (uint8_t)r.intent;
            if (r.ctl) lds_add(&sh.rc[r.route], 1);
        }
        vote<NW>(sh.m_int[0], t, r.alive && r.intent == 0);
        vote<NW>(sh.m_int[1], t, r.alive && r.intent == 1);
        vote<NW>(sh.m_int[2], t, r.alive && r.intent == 2);
    }
    // processing order of scene_update: lane, then intention, then j (ref :233-275)
    static PVE_HD void ph_order(int t, Sh &sh, Regs &r)
    {
        r.ord = t;
        // list capacities, route-sorted controlled vehicles and pair counts: three prefix sums over the 16 routes / lists
        // in the first 16 lanes (rc is complete: barrier behind S1)
        if (t < 64) {                                     // (uniform per wave: the other waves skip the sums and scans)
            int cap = 0, rcv = 0, prs = 0;
            if (t < ND) {
                const unsigned mr = sh.tab.mroutes[t];
#pragma unroll
                for (int rt = 0; rt < ND; rt++) cap = mad24((int)((mr >> rt) & 1u), sh.rc[rt], cap);   // (bit extract + 24-bit multiply-add)
                rcv = sh.rc[t];
                prs = mul24(rcv, sh.tab.nl[t]);
            }
#if PVE_DEVICE_CODE
            const int i1 = wave_incl_scan(t, cap, nullptr), i2 = wave_incl_scan(t, rcv, nullptr), i3 = wave_incl_scan(t, prs, nullptr);
#else
            if (t == 0) sh.emu_scan = sh.emu_scan2 = sh.emu_scan3 = 0;
            const int i1 = wave_incl_scan(t, cap, &sh.emu_scan), i2 = wave_incl_scan(t, rcv, &sh.emu_scan2),
                      i3 = wave_incl_scan(t, prs, &sh.emu_scan3);
#endif
            if (t < ND) {
                sh.lbase[t + 1] = (int16_t)(i1 > 32767 ? 32767 : i1); sh.rbase[t + 1] = (int16_t)i2; sh.pbase[t + 1] = (int16_t)i3;
                if (t == 0) { sh.lbase[0] = 0; sh.rbase[0] = 0; sh.pbase[0] = 0; }
            }
        }
        if (t < Sh::PE / 4) ((int *)sh.u_list)[t] = -1;          // 0xFF = "no entry here" (RANK skips the gaps)
        // rank among the controlled vehicles of its route; ORDER2 files the vehicle at rbase[route] + rank (rbase is the
        // prefix sum written above: complete behind this phase's barrier -- no second 16-term sum per thread)
        if (r.alive && r.ctl) sh.rk[t] = (uint8_t)lds_claim(&sh.rfill[r.route], 1);
        if (r.alive) {
            const int ls = sh.hd.lane_start[r.lane], le = sh.hd.lane_start[r.lane + 1];
            // o = ls + (vehicles of the lane with a smaller intention) + (those with the same intention in the slots below t):
            // the two masks are selected per lane and counted in straight-line form (a loop over the intentions with a
            // branch per case runs every case for every wave that mixes intentions: ten two-word popcounts instead of three)
            u64 mlt[NW], meq[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const u64 m0 = sh.m_int[0][w], m1 = sh.m_int[1][w], m2 = sh.m_int[2][w];
                mlt[w] = r.intent == 0 ? 0ull : (r.intent == 1 ? m0 : (m0 | m1));
                meq[w] = r.intent == 0 ? m0 : (r.intent == 1 ? m1 : m2);
            }
            const int o = ls + (mask_below<NW>(mlt, le) - mask_below<NW>(mlt, ls)) + (mask_rank<NW>(meq, t) - mask_below<NW>(meq, ls));
            r.ord = o;
            sh.ord[t] = (int16_t)o;
            sh.slot_at[o] = (int16_t)t;
        }
    }
    static PVE_HD void ph_order2(int t, Sh &sh, const Regs &r)
    {
        if (r.alive && r.ctl) sh.ctl_by_route[sh.rbase[r.route] + (int)sh.rk[t]] = (uint8_t)t;
        const int N = sh.hd.n_alive;
        vote<NW>(sh.m_ctl_ord, t, t < N && mask_test(sh.m_ctl, sh.slot_at[t < N ? t : 0]));
    }
    // PAIRS (positions are final: after S3): pair q = (route r, its i-th controlled vehicle, the k-th list the route can
    // appear in), routes in order: every thread takes pairs q = t, t + CAP, ... -- the same amount of work for everybody,
    // whatever the mix of routes in the wave.  COUNT = false: a member claims the next entry of its list.  COUNT = true
    // (only when the capacity upper bounds overflow the pool): members are counted per list, ph_pairs_exact then packs
    // the segments to their exact sizes before the fill pass.
    template <bool COUNT>
    static PVE_HD void pairs_loop(const PVE_AS4 GeoConst &g, int t, Sh &sh)
    {
        const int total = sh.pbase[ND];
        // route of pair q = the largest rt with pbase[rt] <= q: the 15 inner prefix sums are read once per thread, every pair
        // then takes 4 compares and a select tree instead of 15 compares + adds
        int pb[ND];
#pragma unroll
        for (int k = 1; k < ND; k++) pb[k] = sh.pbase[k];
        for (int q = t; q < total; q += CAP) {
            const bool c8 = q >= pb[8];
            const bool c4 = q >= (c8 ? pb[12] : pb[4]);
            const bool c2 = q >= (c8 ? (c4 ? pb[14] : pb[10]) : (c4 ? pb[6] : pb[2]));
            const int o3 = c8 ? (c4 ? (c2 ? pb[15] : pb[13]) : (c2 ? pb[11] : pb[9]))
                              : (c4 ? (c2 ? pb[7] : pb[5]) : (c2 ? pb[3] : pb[1]));
            const int rt = (c8 ? 8 : 0) + (c4 ? 4 : 0) + (c2 ? 2 : 0) + ((q >= o3) ? 1 : 0);
            // membership of the pair's vehicle x in list d = member() in straight-line form: x comes from route rt's segment
            // of the route-sorted list, so its lane and intention are table entries of rt (no reads of the vehicle's own
            // bytes), every table read is unconditional on a clamped index (batches of independent LDS reads instead of a
            // chain of guarded ones), both outcomes are computed and selected
            const int off = q - sh.pbase[rt], n = sh.tab.nl[rt], ni = (int)sh.tab.ninv[rt], rb = sh.rbase[rt];
            const int lx = sh.tab.dir_lane[rt], ix = sh.tab.dir_index[rt];
            const int i = mul24(off, ni) >> 15, k = off - mul24(i, n);                  // off / n, off % n (exact: off < 2048, n <= 10)
            const int x = sh.ctl_by_route[rb + i], d = sh.tab.lst[rt][k];
            const double px = sh.p[x];
            const int li = sh.tab.dir_lane[d], m = sh.tab.dir_index[d], kk = sh.tab.pos[d][rt], ty = sh.tab.dty[d];
            const double inb_x = sh.tab.inbox[ix];
            const double *ev = sh.tab.vd[ty][kk < 0 ? 0 : kk];
            const double e0 = ev[0], e1 = ev[1], e2 = ev[2], e3 = ev[3], inb_m = sh.tab.inbox[m];
            const double qd = px - inb_x;                                               // ref :251-252
            const double delta = (px - e0) + e1;                                        // ref :453-660 / :733-803
            const bool own = rt == d, same = lx == li;
            const bool ok = own | (same ? (qd > 0) : ((kk >= 0) & (delta > 0)));       // ref :246-270
            if (!ok) continue;
            const double vo = own ? px : (same ? qd + inb_m : (delta + e2) - e3);
            if (COUNT) { lds_add(&sh.cnt2[d], 1); continue; }
            // (4-lane layout, list of a left-turn route, member of the opposing left-turn route: from the back, not sorted)
            const bool back = (g.lane_num == 4) & (ty == 0) & (rt == (int)sh.tab.opp[d]) & !own;
            const int qb = back ? lds_claim(&sh.ofill[d], 1) : 0;
            const int e = back ? (sh.lbase[d + 1] - 1 - qb) : (sh.lbase[d] + lds_claim(&sh.fill[d], 1));
            sh.u_vd[e] = vo; sh.u_slot[e] = (uint8_t)x; sh.u_list[e] = (uint8_t)(back ? 0xFF : d);
            // (x is a back entry of exactly one list, opp[its route]: its index there = its row of the far-conflict table;
            //  rk[x], the rank ORDER2 consumed, is free to hold it -- bit 7 = "filed")
            if (back) sh.rk[x] = (uint8_t)(0x80 | qb);
        }
    }
    // The common case -- the capacity upper bounds fit the entry pool -- needs neither the counting pass nor the two
    // phases that pack the segments: pairs_over() (uniform: an LDS value) lets the kernels branch around COUNT / EXACT /
    // APPLY and their three barriers; ph_pairs_mode (same phase as ORDER2, lbase is complete there) sets pool_ok for it.
    static PVE_HD bool pairs_over(const Sh &sh, bool force_scan) { return !force_scan && sh.lbase[ND] > Sh::PE; }
    static PVE_HD void ph_pairs_mode(int t, Sh &sh, bool force_scan)
    {
        if (t == 0) sh.pool_ok = force_scan ? 0 : (sh.lbase[ND] > Sh::PE ? 2 : 1);
        if (t == 1) {
            // far-conflict table of the 4-lane layout: one block per left-turn route d = 3 q: rc[opp[d]] rows x rc[d] columns,
            // behind the entry pool's capacity bound (no room, or no lists: the egos replay the adjustments themselves)
            int o = 0;
            for (int q = 0; q < 4; q++) { sh.tbase[q] = (int16_t)o; o += sh.rc[sh.tab.opp[3 * q] < 0 ? 0 : sh.tab.opp[3 * q]] * sh.rc[3 * q]; }
            sh.tbase[4] = (int16_t)o;
            const int start = sh.lbase[ND];
            sh.tstart = (int16_t)(start > 32767 ? 32767 : start);
            sh.fix_ok = (!force_scan && start + o <= Sh::PE) ? 1 : 0;
        }
    }
    // one thread per controlled vehicle x of a left-turn route (4-lane layout): the value of x's entry in the list of the
    // OPPOSING left-turn route d after each of d's egos, in slot order (ref :1301-1319).  Round 6: in the RANK phase (behind
    // FILL's barrier), row = the index q of x's entry among the back entries of list d (rk[x], filed by FILL) -- an ego of d
    // then reads entry q and table cell (q, its column) without the slot -> rank -> row chain.  Reads positions, masks and
    // its own back entry; writes beyond the capacity bound of the entry pool (RANK reads there only masked-out tail values).
    static PVE_HD void ph_fix_table(const PVE_AS4 GeoConst &g, int t, Sh &sh, Regs &r)
    {
        if (g.lane_num != 4 || !sh.fix_ok) return;
        if (!(r.alive && r.ctl) || r.route % 3 != 0) return;
        const int d = sh.tab.opp[r.route];                    // (the relation is symmetric: opp[opp[d]] = d)
        if (d < 0) return;
        const int q = sh.rk[t];
        if (!(q & 0x80)) return;                              // not in list d: no back entry, no row
        const int li = sh.tab.dir_lane[d], m = sh.tab.dir_index[d];
        double vc = sh.u_vd[sh.lbase[d + 1] - 1 - (q & 0x7F)];   // its build-time value in list d (= member())
        const int ls = sh.hd.lane_start[li], le = sh.hd.lane_start[li + 1];
        double *row = sh.u_vd + sh.tstart + sh.tbase[d / 3] + mul24(q & 0x7F, sh.rc[d]);
        int i = 0;
        for (int w2 = 0; w2 < NW; w2++) {
            u64 eb = sh.m_ctl[w2] & sh.m_int[m][w2] & below_sel(le - w2 * 64) & ~below_sel(ls - w2 * 64);
            for (; eb; eb &= eb - 1) {
                const double pe = sh.p[w2 * 64 + __builtin_ctzll(eb)];
                const double ori = vc + g.fix_d;                                       // ref :1304
                if (pe < ori) {                                                        // ref :1305-1312
                    const double r2 = ori - g.fix_hi + g.fix_lo;
                    vc = (r2 < pe) ? pe + 1 : r2;
                } else {                                                               // ref :1313-1319
                    const double r2 = ori + g.fix_hi - g.fix_lo;
                    vc = (r2 > pe) ? pe - 1 : r2;
                }
                row[i++] = vc;
            }
        }
    }
    static PVE_HD void ph_pairs_count(const PVE_AS4 GeoConst &g, int t, Sh &sh, bool force_scan)
    {
        if (force_scan || sh.lbase[ND] <= Sh::PE) return;       // the upper bounds fit (or the scan is forced): nothing to count
        pairs_loop<true>(g, t, sh);
    }
    static PVE_HD void ph_pairs_exact(int t, Sh &sh, bool force_scan)
    {
        const bool over = sh.lbase[ND] > Sh::PE;                // (uniform; read before anybody rewrites lbase: same phase,
        int c = 0;                                              //  but only the first 16 lanes write, after their own read)
        if (over && t < ND) c = sh.cnt2[t];
#if PVE_DEVICE_CODE
        const int incl = wave_incl_scan(t, c, nullptr);
#else
        if (t == 0) sh.emu_scan = 0;
        const int incl = wave_incl_scan(t, c, &sh.emu_scan);
#endif
        if (over && t < ND) sh.lnew[t] = (int16_t)(incl > 32767 ? 32767 : incl);
        if (t == 0) sh.pool_ok = force_scan ? 0 : (over ? 2 : 1);  // 2: exact segment offsets are in lnew (applied by FILL)
    }
    static PVE_HD void ph_pairs_fill(const PVE_AS4 GeoConst &g, int t, Sh &sh)
    {
        if (sh.pool_ok == 0 || (sh.pool_ok == 2 && sh.lnew[ND - 1] > Sh::PE)) return;
        pairs_loop<false>(g, t, sh);
    }
    // (between EXACT and FILL: one thread per list moves the exact offsets into lbase; own phase = own barrier)
    static PVE_HD void ph_pairs_apply(int t, Sh &sh)
    {
        if (sh.pool_ok != 2) return;
        if (t < ND) sh.lbase[t + 1] = sh.lnew[t];
        if (t == ND && sh.lnew[ND - 1] > Sh::PE) sh.pool_ok = 0;    // even the exact sizes do not fit: membership scan
    }
    // RANK: counting sort of every list by (vd, slot) = the reference's stable sort by vd of the list it builds in
    // (lane, intention, j) order (ref :271); entry-parallel, 8 independent LDS reads per round (cf. Tick::ph_rank)
    // Positions are claimed with a tagged exchange and runs of equal distances filed by whoever finds the position taken,
    // as in Tick::ph_rank.
    static PVE_HD void ph_rank(int t, Sh &sh, int salt = 0)
    {
        if (!sh.pool_ok) return;
        const int M = sh.lbase[ND] < Sh::PE ? sh.lbase[ND] : Sh::PE;
        const unsigned tag = ((unsigned)sh.hd.ticks + (unsigned)salt * 0x9E37u) << 16;
        for (int e = t; e < M; e += CAP) {
            const int d = sh.u_list[e];
            if (d == 0xFF) continue;                      // capacity nobody claimed
            const double vd = sh.u_vd[e];
            const int slot = sh.u_slot[e];
            const int lo = sh.lbase[d], hi = lo + sh.fill[d];
            int pos = 0;
            int f = lo;
            for (; f + 8 <= hi; f += 8) {
                double w[8];
#pragma unroll
                for (int k = 0; k < 8; k++) w[k] = sh.u_vd[f + k];
#pragma unroll
                for (int k = 0; k < 8; k++) pos += (w[k] < vd) ? 1 : 0;
            }
            if (f < hi) {                                 // tail (< 8 entries): one more round of independent reads; what lies
                const int n = hi - f;                     // behind the list (the next list, unclaimed capacity, the pad) is masked
                double w[7];
#pragma unroll
                for (int k = 0; k < 7; k++) w[k] = sh.u_vd[f + k];
#pragma unroll
                for (int k = 0; k < 7; k++) pos += (k < n) & (w[k] < vd);
            }
            // (relaxed atomic, released by the claim: Tick::ph_rank)
            if (d == sh.route_of[slot]) lds_store_relaxed(&sh.mypos[slot], (int16_t)pos);   // the vehicle's own entry (vd = p)
            const unsigned old = lds_xchg(&sh.s_idx[lo + pos], tag | (unsigned)e);
            if ((old & 0xFFFF0000u) == tag) {             // an entry with the same distance was here first (or a stale word)
                for (f = lo; f < hi; f++) {
                    if (!(sh.u_vd[f] == vd)) continue;
                    const int sf = sh.u_slot[f];
                    int rk = 0;
                    for (int g = lo; g < hi; g++) rk += (sh.u_vd[g] == vd && sh.u_slot[g] < sf) ? 1 : 0;
                    lds_store_relaxed(&sh.s_idx[lo + pos + rk], tag | (unsigned)f);
                    if (d == sh.route_of[sf]) lds_store_relaxed(&sh.mypos[sf], (int16_t)(pos + rk));
                }
            }
        }
    }

    // ============================================================== membership of vehicle x in the list of route d
    // (physical lane li, intention index m), ref :240-270.  vo = the entry's virtual distance at list build.
    static PVE_HD bool member(const PVE_AS4 GeoConst &g, const Sh &sh, int d, int li, int m, int x, double &vo)
    {
        const int lx = sh.lane_of[x], rx = sh.route_of[x];
        const double px = sh.p[x];
        if (lx == li) {
            if (rx == d) { vo = px; return true; }                                     // ref :246-249
            const double q = px - sh.tab.inbox[sh.intent_of[x]];                           // ref :251-252
            if (q > 0) { vo = q + sh.tab.inbox[m]; return true; }                          // ref :253-257
            return false;
        }
        const int k = sh.tab.pos[d][rx];                                                   // ref :258
        if (k < 0) return false;
        const double *e = sh.tab.vd[sh.tab.dty[d]][k];
        const double delta = (px - e[0]) + e[1];                                       // ref :453-660 / :733-803
        if (!(delta > 0)) return false;
        vo = (delta + e[2]) - e[3];
        return true;
    }

    // 4-lane far-conflict fix-up (ref :1301-1319): current value of an entry of the opposing left-turn route as seen
    // by ego t: the controlled vehicles of the ego's route up to and including the ego have, one after the other,
    // re-written it (the adjusted copy is stored back at ref :286-287, so the adjustments compound)
    static PVE_HD double adjusted(const PVE_AS4 GeoConst &g, const Sh &sh, int m, int ls, int le, int t, double vc)
    {
        const int hi = (le < t + 1) ? le : t + 1;               // the egos of the route in slots [ls, hi)
        for (int w2 = 0; w2 < NW; w2++) {
            u64 eb = sh.m_ctl[w2] & sh.m_int[m][w2] & below_sel(hi - w2 * 64) & ~below_sel(ls - w2 * 64);
            for (; eb; eb &= eb - 1) {
                const int e = w2 * 64 + __builtin_ctzll(eb);
                const double pe = sh.p[e];
                const double ori = vc + g.fix_d;                                       // ref :1304
                if (pe < ori) {                                                        // ref :1305-1312
                    const double r2 = ori - g.fix_hi + g.fix_lo;
                    vc = (r2 < pe) ? pe + 1 : r2;
                } else {                                                               // ref :1313-1319
                    const double r2 = ori + g.fix_hi - g.fix_lo;
                    vc = (r2 > pe) ? pe - 1 : r2;
                }
            }
        }
        return vc;
    }

    // 4-lane layout, ego of a left-turn route d (ref :1301-1319, :1340-1405): predecessor and 6 nearest from the sorted list
    // (everything but the opposing left-turn route's entries: the 12-lane kernel's window walk, Tick::walk_window) AND the `no`
    // unsorted opposing entries at the back of the list's capacity (entry eb - q; current value as this ego sees it = table cell
    // (q, its column): u_vd[tcell + q * tcols]) in ONE selection on 32-bit keys -- the float32 image of |d| with the low 5 bits
    // replaced by the candidate code (0..5 / 8..13: the window, 16 + q: opposing entry q).  Per opposing entry: three
    // independent LDS reads, the predecessor test on the build-time value (ref :1353), one key, six v_min_u32 / v_max_u32 pairs
    // (round 5: a float64 insertion chain of ~80 instructions per entry).  The slots and values of the 6 winners are read once
    // they are known.  -> true = not decided here (pr / pvo / pvd, the predecessor, are final all the same; the caller takes
    // Tick::walk_window_exact for the window and the float64 insertion for the opposing entries): the window's own
    // ambiguities (Tick::walk_window), more than 16 opposing entries, an opposing entry whose key agrees with a neighbour's
    // in the upper 27 bits, a sixth winner not separated from the best loser, float64 distances of the winners that decrease
    // or -- with an opposing entry involved -- are equal (the reference's stable sort then decides by list position).
    static PVE_HD bool walk_merge4(Sh &sh, int base, int n, int s, double ps, int t, int eb, int no, int tcell, int tcols, Regs &r,
                                   int &pr, double &pvo, double &pvd)
    {
        const auto *sidx = sh.s_idx + base;
        const int last = n - 1;
        double lraw[NNB + 1], rraw[NNB];
        int prs = -1;
#pragma unroll
        for (int i = 0; i < NNB + 1; i++) {
            const int pos = s - 1 - i, pc = pos >= 0 ? pos : 0;
            const int e = sidx_at(sidx, pc);
            lraw[i] = sh.u_vd[e];
            if (i == 0) prs = (int)sh.u_slot[e];
        }
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const int pos = s + 1 + i, pc = pos <= last ? pos : last;
            rraw[i] = sh.u_vd[sidx_at(sidx, pc)];
        }
        pr = (s > 0) ? prs : -1; pvd = (s > 0) ? lraw[0] : 0.0;                      // ref :1353-1354
        pvo = pvd;                                        // build-time distance of the predecessor (order), pvd = its current value
        double dl[NNB + 1], dr[NNB];
#pragma unroll
        for (int i = 0; i < NNB + 1; i++) dl[i] = fabs(lraw[i] - ps);                 // ref :1388
#pragma unroll
        for (int i = 0; i < NNB; i++) dr[i] = fabs(rraw[i] - ps);
        bool amb = no > 16;
#pragma unroll
        for (int i = 1; i < NNB + 1; i++) amb = amb | ((s - 1 - i >= 0) & (dl[i] == dl[i - 1]));
        unsigned kl[NNB], kq[NNB];
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const unsigned cl = (unsigned)i, cr = 8u + (unsigned)i;
            const unsigned bl = Base::f32_bits((float)dl[i]), br = Base::f32_bits((float)dr[i]);
            kl[i] = (s - 1 - i >= 0) ? ((bl & ~31u) | cl) : (0xFF800000u | (cl << 5) | cl);
            kq[i] = (s + 1 + i <= last) ? ((br & ~31u) | cr) : (0xFF800000u | (cr << 5) | cr);
        }
        unsigned w[NNB], acc = ~0u;                   // acc = the smallest XOR of a (winner, loser) pair of the merge step
#pragma unroll
        for (int i = 0; i < NNB; i++) {
            const unsigned a = kl[i], b = kq[NNB - 1 - i];
            w[i] = Base::umin(a, b);
            // (an EXACT float64 tie of a left and a right candidate is decided by the codes: left first, the reference's
            //  list order -- the frequent case, equally spaced platoons; only unequal distances need separated keys)
            acc = Base::umin(acc, (dl[i] == dr[NNB - 1 - i]) ? ~0u : (a ^ b));
        }
#define PVE_CE(A, B) { const unsigned lo_ = Base::umin(w[A], w[B]), hi_ = Base::umax(w[A], w[B]); w[A] = lo_; w[B] = hi_; }
        PVE_CE(0, 5) PVE_CE(1, 3) PVE_CE(2, 4)
        PVE_CE(1, 2) PVE_CE(3, 4)
        PVE_CE(0, 3) PVE_CE(2, 5)
        PVE_CE(0, 1) PVE_CE(2, 3) PVE_CE(4, 5)
        PVE_CE(1, 2) PVE_CE(3, 4)
#undef PVE_CE
        amb = amb | (acc < 32u);
        // the opposing entries: predecessor by build-time order, key into the sorted six (the seventh = the best loser so far)
        unsigned lose = ~0u;
        const int nq = no > 16 ? 0 : no;
        for (int q = 0; q < nq; q++) {
            const int x = sh.u_slot[eb - q];
            const double vo = sh.u_vd[eb - q];
            const double vc = sh.u_vd[tcell + mul24(q, tcols)];
            const bool before = vo < ps || (vo == ps && x < t);
            if (before && (pr < 0 || vo > pvo || (vo == pvo && x > pr))) { pvo = vo; pr = x; pvd = vc; }
            unsigned key = (Base::f32_bits((float)fabs(vc - ps)) & ~31u) | (16u + (unsigned)q);
#pragma unroll
            for (int k = 0; k < NNB; k++) {
                const unsigned lo_ = Base::umin(w[k], key);
                key = Base::umax(w[k], key);
                w[k] = lo_;
            }
            lose = Base::umin(lose, key);
        }
        // an opposing entry next to a key it is not separated from in the upper 27 bits: not decided on these keys (two window
        // keys among the six: their codes order them as in Tick::walk_window, and the float64 check below sees both)
#pragma unroll
        for (int k = 1; k < NNB; k++) amb = amb | (((w[k] ^ w[k - 1]) < 32u) & (((w[k] | w[k - 1]) & 16u) != 0u) & ((int)w[k] >= 0));
        // (the best loser is not read back: ANY loser the sixth is not separated from leaves the decision to the float64 keys)
        amb = amb | (((w[NNB - 1] ^ lose) < 32u) & ((int)w[NNB - 1] >= 0));
        // the winners' slots and values: index reads back to back, then the value reads
        int es[NNB], ev[NNB];
#pragma unroll
        for (int k = 0; k < NNB; k++) {
            const int code = (int)(w[k] & 31u);
            const bool mg = (code & 16) != 0;
            const int pos = (code & 8) ? (s - 7 + code) : (s - 1 - code);        // = s + 1 + (code - 8) on the right
            const int pc = ((int)w[k] >= 0 && !mg) ? pos : 0;                    // (the keys of absent candidates have bit 31 set)
            const int e = sidx_at(sidx, pc);
            const int q = code & 15;
            es[k] = mg ? eb - q : e;
            ev[k] = mg ? tcell + mul24(q, tcols) : e;
        }
        int sl[NNB]; double vv[NNB], dk[NNB];
#pragma unroll
        for (int k = 0; k < NNB; k++) { sl[k] = (int)sh.u_slot[es[k]]; vv[k] = sh.u_vd[ev[k]]; }
#pragma unroll
        for (int k = 0; k < NNB; k++) dk[k] = fabs(vv[k] - ps);
#pragma unroll
        for (int k = 1; k < NNB; k++)
            amb = amb | (((int)w[k] >= 0) & ((dk[k] < dk[k - 1]) | ((dk[k] == dk[k - 1]) & (((w[k] | w[k - 1]) & 16u) != 0u))));
        if (amb) return true;
#pragma unroll
        for (int k = 0; k < NNB; k++) {
            const bool ok = (int)w[k] >= 0;
            r.kr[k] = ok ? sl[k] : -1;
            r.kv[k] = ok ? vv[k] : 0.0;
        }
        return false;
    }

    // ============================================================== SCAN: list heads, predecessor, 6 nearest
    // FIX4 = false compiles the 4-lane far-conflict path out (the launcher picks it for lane_num 8 / 12: the registers
    // that path needs would otherwise be spilled in the common phases of every layout)
    template <bool FIX4 = true>
    static PVE_HD void ph_scan(const PVE_AS4 GeoConst &g, int t, Sh &sh, Regs &r)
    {
        constexpr bool KEYMERGE = FIX4 && CAP == 128;      // (walk_merge4: measured a gain at 128 slots, a loss at 64)
        r.reward = 0; r.hit = 0; r.hdr = -1;
#pragma unroll
        for (int k = 0; k < NNB; k++) { r.kr[k] = -1; r.kv[k] = 0; }
        // (a) one thread per list d: head of list d, persisted for next tick's step (ref :1517); lists are rebuilt only
        //     when their physical lane holds a vehicle (ref :234), otherwise the old head stays (stale by design)
        const bool lists = sh.pool_ok != 0;
        bool hrebuilt = false, hvalid = false;             // (combined by ballots below: cf. Tick::ph_scan)
        if (t >= CAP - ND && t - (CAP - ND) < g.dir_num) {     // the last 16 threads: the (mostly empty) tail of the last wave
            const int d = t - (CAP - ND), li = sh.tab.dir_lane[d], m = sh.tab.dir_index[d];
            if (sh.hd.lane_start[li + 1] > sh.hd.lane_start[li]) {
                hrebuilt = true;
                double best = INFINITY; int bs = -1;
                if (lists) {
                    if (sh.fill[d] > 0) { const int e0 = sidx_at(sh.s_idx, sh.lbase[d]); bs = sh.u_slot[e0]; best = sh.u_vd[e0]; }   // sorted: the first entry
                    if (FIX4) {                           // ... or one of the unsorted opposing entries (4-lane left-turn lists)
                        const int no = sh.ofill[d], eb = sh.lbase[d + 1] - 1;
                        for (int k = 0; k < no; k++) {
                            const double vo = sh.u_vd[eb - k]; const int x = sh.u_slot[eb - k];
                            if (vo < best || (vo == best && x < bs)) { best = vo; bs = x; }
                        }
                    }
                } else {
                    for (int w = 0; w < NW; w++)
                        for (u64 bits = sh.m_ctl[w]; bits; bits &= bits - 1) {
                            const int x = w * 64 + __builtin_ctzll(bits);
                            double vo;
                            if (member(g, sh, d, li, m, x, vo) && vo < best) { best = vo; bs = x; }   // ties: lower slot stays
                        }
                }
                if (bs >= 0) {
                    hvalid = true;
                    const int hl = sh.lane_of[bs];
                    sh.hd.head_lane[d] = hl;
                    sh.hd.head_j[d] = bs - sh.hd.lane_start[hl];
                }
            }
        }
#if PVE_DEVICE_CODE
        {   // lanes 48..63 of the LAST wave hold the lists 0..15: one read-modify-write of the header word by its first such lane
            const unsigned mr = (unsigned)(__ballot(hrebuilt) >> 48), mv = (unsigned)(__ballot(hvalid) >> 48);
            if (t == CAP - ND) sh.hd.head_valid = (sh.hd.head_valid & ~(int)mr) | (int)mv;
        }
#else
        if (hrebuilt) { const int d = t - (CAP - ND); if (hvalid) sh.hd.head_valid |= 1 << d; else sh.hd.head_valid &= ~(1 << d); }
#endif
        if (!(r.alive && r.ctl)) return;
        // (b) every controlled vehicle goes through the members of its route's list
        const int d = r.route, li = r.lane, m = r.intent;
        const double me = r.p;                                   // own entry: vd = p, never adjusted
        const bool fix = FIX4 && (g.lane_num == 4) && (d % 3 == 0);      // ref :1301
        const int opp = sh.tab.opp[d];
        const int ls = sh.hd.lane_start[li], le = sh.hd.lane_start[li + 1];
        const unsigned mroutes = sh.tab.mroutes[d];                  // routes that can appear in list d (cheap early reject)
        // current value of member x's entry (build-time value vo) as this ego sees it: table lookup, or the replay
        const bool tabf = fix && sh.fix_ok != 0;
        int tcol = 0;
        if (tabf) {                                           // this ego's column: its index among the egos of its route
            for (int w2 = 0; w2 < NW; w2++)
                tcol += __builtin_popcountll(sh.m_ctl[w2] & sh.m_int[m][w2] & below_sel(t - w2 * 64) & ~below_sel(ls - w2 * 64));
            tcol += sh.tstart + sh.tbase[d / 3];
        }
        const int tcols = sh.rc[d];
        auto adj = [&](int x, double vo) -> double {
            return tabf ? sh.u_vd[mad24((int)sh.rk[x] & 0x7F, tcols, tcol)] : adjusted(g, sh, m, ls, le, t, vo);
        };
        double bvo = -INFINITY, bvc = 0; int bslot = -1;
        // the 6 nearest so far, sorted by |vd - vd_self|: only (distance, slot) travel through the insertion chain;
        // the build-time / current distances of the 6 winners are re-derived at the end, and the rare exact
        // distance ties (which the reference's stable sort breaks by list position) take the full-key path.
        // Every comparison carries its tie-break explicitly, so the order in which the members arrive does not matter.
        double kd[NNB]; int ks[NNB];
#pragma unroll
        for (int k = 0; k < NNB; k++) { kd[k] = INFINITY; ks[k] = -1; }
        auto consider = [&](int x, double vo) {
            const double vc = (fix && sh.route_of[x] == opp) ? adj(x, vo) : vo;
            // predecessor in list order = stable sort by the build-time distance (ref :271, :1353)
            const bool before = vo < me || (vo == me && x < t);
            if (before && (vo > bvo || (vo == bvo && x > bslot))) { bvo = vo; bslot = x; bvc = vc; }
            // 6 nearest: stable sort of the list by |vd - vd_self| on the current values (ref :1383-1397)
            double cd = fabs(vc - me); int cs = x;
            bool tie = false;
#pragma unroll
            for (int k = 0; k < NNB; k++) tie = tie || (cd == kd[k]);
            if (!tie) {
                bool ins = false;                             // once placed, the tail shifts down by one
#pragma unroll
                for (int k = 0; k < NNB; k++) {
                    const bool sw = ins || cd < kd[k];
                    ins = sw;
                    const double td = sw ? kd[k] : cd; const int ts = sw ? ks[k] : cs;
                    kd[k] = sw ? cd : kd[k]; ks[k] = sw ? cs : ks[k];
                    cd = td; cs = ts;
                }
            } else {
                double co = vo;
#pragma unroll
                for (int k = 0; k < NNB; k++) {
                    double eo = INFINITY;                     // build-time distance of the entry (list position)
                    if (ks[k] >= 0) member(g, sh, d, li, m, ks[k], eo);
                    const bool sw = key_less(cd, co, cs, kd[k], eo, ks[k] < 0 ? 0x7fffffff : ks[k]);
                    const double td = sw ? kd[k] : cd, to = sw ? eo : co; const int ts = sw ? ks[k] : cs;
                    kd[k] = sw ? cd : kd[k]; ks[k] = sw ? cs : ks[k];
                    cd = td; co = to; cs = ts;
                }
            }
        };
        if (lists) {
            // sorted list: the 12-lane kernel's window walk (ref :1340-1405).  The lists of the 4-lane left-turn routes hold
            // everything BUT the entries of the opposing left-turn route there, whose values this ego sees re-written (ref
            // :1301-1319); those few (ofill[d], unsorted, at the back of the list's capacity) are merged into the walk's result:
            // the predecessor by build-time order (ref :1353), the 6 nearest by current distance (ref :1383-1397).  Equal
            // distances between a merged entry and a winner -- the reference's stable sort then decides by list position, which
            // the split list does not carry -- take the exact insertion below.
            int pr; double pvd, pvo;
            bool amb = false, seeded = false;
            if (KEYMERGE && g.lane_num == 4) {
                // (round 6, 128 slots: the window walk and the merge of the opposing entries as ONE selection on 32-bit keys --
                //  every ego of the layout through the same code, the egos of the other routes with no opposing entries: a wave
                //  that mixes routes walks its windows once)
                const bool mg = fix && tabf;
                amb = walk_merge4(sh, sh.lbase[d], sh.fill[d], sh.mypos[t], me, t, sh.lbase[d + 1] - 1, mg ? sh.ofill[d] : 0, tcol, tcols, r, pr,
                                  pvo, pvd);
                if (amb) {
                    // not decided on the keys (~0.8 % of the egos, mostly the ulp-level near ties of equally spaced platoons): the
                    // window exactly, then the opposing entries through the float64 insertion with its full tie-breaks; the
                    // predecessor is already known.  (the 64-slot form of that insertion, below, was measured here too: its
                    // registers cost the fast path more than this path's re-derivation of the winners: 33.9 vs 32.5 us)
                    Base::walk_window_exact(sh, sh.lbase[d], sh.fill[d], sh.mypos[t], me, r);
                    if (mg) {
#pragma unroll
                        for (int k = 0; k < NNB; k++) { ks[k] = r.kr[k]; kd[k] = r.kr[k] >= 0 ? fabs(r.kv[k] - me) : INFINITY; }
                        bslot = pr; bvo = pr >= 0 ? pvo : -INFINITY; bvc = pvd;
                        const int no = sh.ofill[d], eb = sh.lbase[d + 1] - 1;
                        for (int q = 0; q < no; q++) consider(sh.u_slot[eb - q], sh.u_vd[eb - q]);
                        seeded = true;
                    } else amb = false;                   // (no opposing entries: the exact window is the answer)
                }
                if (fix && !tabf) { amb = true; seeded = false; }   // (no room for the table behind the pool, rare: every member below)
            } else {
                // (capacity 64 -- lighter traffic: most opposing entries are farther away than the 6th winner, and the wave-level
                //  skip of the insertion below is cheaper than a key per entry: 26.5 vs 29.0 us per tick of 4096 x 64 -- and the
                //  layouts without the far-conflict path)
                Base::walk_window(sh, sh.lbase[d], sh.fill[d], sh.mypos[t], me, r, pr, pvd);
                pvo = pvd;                                        // build-time distance of the predecessor (order), pvd = its current value
                if (fix) {
                    const int no = sh.ofill[d], eb = sh.lbase[d + 1] - 1;
                    // most opposing entries are farther away than the 6th winner: the insertion chain (~80 vector instructions
                    // per entry) only runs in waves where some lane's entry can enter the 6 -- at 4 waves per SIMD every vector
                    // instruction of this loop is 16 cycles of the SIMD
                    double dk[NNB];
                    bool have_dk = false;
                    double d5 = r.kr[NNB - 1] >= 0 ? fabs(r.kv[NNB - 1] - me) : INFINITY;
                    for (int q = 0; q < no; q++) {
                        const int x = sh.u_slot[eb - q];
                        const double vo = sh.u_vd[eb - q];
                        const double vc = adj(x, vo);
                        const bool before = vo < me || (vo == me && x < t);
                        if (before && (pr < 0 || vo > pvo || (vo == pvo && x > pr))) { pvo = vo; pr = x; pvd = vc; }
                        double cd = fabs(vc - me), cv = vc; int cs = x;
#if PVE_DEVICE_CODE
                        if (__builtin_amdgcn_ballot_w64(cd <= d5) == 0) continue;
#else
                        if (!(cd <= d5)) continue;
#endif
                        if (!have_dk) {
#pragma unroll
                            for (int k = 0; k < NNB; k++) dk[k] = r.kr[k] >= 0 ? fabs(r.kv[k] - me) : INFINITY;
                            have_dk = true;
                        }
                        bool ins = false;
#pragma unroll
                        for (int k = 0; k < NNB; k++) {
                            amb = amb | (cd == dk[k]);
                            const bool sw = ins | (cd < dk[k]);
                            ins = sw;
                            const double td = sw ? dk[k] : cd, tv = sw ? r.kv[k] : cv; const int ts = sw ? r.kr[k] : cs;
                            dk[k] = sw ? cd : dk[k]; r.kv[k] = sw ? cv : r.kv[k]; r.kr[k] = sw ? cs : r.kr[k];
                            cd = td; cv = tv; cs = ts;
                        }
                        d5 = dk[NNB - 1];
                    }
                }
            }
            if (!amb) {
                r.hdr = pr;                                                                 // ref :1348-1354
                r.vir_dis = (pr >= 0) ? (me - pvd) : 100.0;
                sh.hdr[t] = (int16_t)pr;
                sh.virdis[t] = r.vir_dis;
                r.count += 1;                                                               // ref :292
                return;
            }
            if (!seeded) {                                // every member through the full-key insertion
                const int no = sh.ofill[d], eb = sh.lbase[d + 1] - 1;
                for (int q = 0; q < no; q++) consider(sh.u_slot[eb - q], sh.u_vd[eb - q]);
                const int e1 = sh.lbase[d] + sh.fill[d];
                for (int e = sh.lbase[d]; e < e1; e++) {
                    const int x = sh.u_slot[e];
                    if (x != t) consider(x, sh.u_vd[e]);
                }
            }
        } else {
            for (int w = 0; w < NW; w++)
                for (u64 bits = sh.m_ctl[w]; bits; bits &= bits - 1) {
                    const int x = w * 64 + __builtin_ctzll(bits);
                    double vo;
                    if (x == t || !((mroutes >> sh.route_of[x]) & 1u) || !member(g, sh, d, li, m, x, vo)) continue;
                    consider(x, vo);
                }
        }
#pragma unroll
        for (int k = 0; k < NNB; k++) {
            double vo = 0;
            const int x = ks[k];
            if (x >= 0) {
                member(g, sh, d, li, m, x, vo);
                if (fix && sh.route_of[x] == opp) vo = adj(x, vo);
            }
            r.kr[k] = x; r.kv[k] = vo;
        }
        r.hdr = bslot;                                                                  // ref :1348-1354
        r.vir_dis = (bslot >= 0) ? (me - bvc) : 100.0;
        sh.hdr[t] = (int16_t)bslot;
        sh.virdis[t] = r.vir_dis;
        r.count += 1;                                                                   // ref :292
    }

    // ============================================================== REWARD + XY collision test (ref :280-334)
    static PVE_HD void ph_reward(const PVE_AS4 GeoConst &g, int t, Sh &sh, Regs &r)
    {
        if (!(r.alive && r.ctl)) return;
        const PVE_AS4 Const &c = g.base;
        const double ps = r.p;
        double t_distance = 2, d_distance = 10;
        const int n0 = r.kr[0];
        if (n0 >= 0) {
            const double vdn = r.kv[0];
            d_distance = fabs(ps - vdn);
            r.closer_p = vdn;
            if (d_distance != 0) t_distance = (ps - vdn) / (r.v - sh.v[n0] + 0.0001);
        } else {
            r.closer_p = 150;
        }
        double r_ = 0;
        if (0 < t_distance && t_distance < 4) r_ += reward_coth_term(t_distance);
        // divisions by constants become multiplications in these reward-only terms (no decision reads them)
        const double jd = r.jerk * c.inv_dt;
        r_ -= jd * jd * (3.0 / 3600.0);
        if (d_distance < 10) {
            double q1 = d_distance * 0.1, q2 = q1 * q1;
            r_ += reward_log_term(q2 * q2 * q1 + 0.00001);
        }
        r_ += (r.v - c.vm) * c.inv_span * 2.0;
        r.reward = dmin(dmax(r_, -20.0), 20.0);
        r.jerk_sum += fabs(jd);
        // pre-filter in single precision: the exact FP64 positions (divisions + polynomials) are only evaluated when the
        // pair is within 5 cm of the threshold band -- the decision itself is always FP64
        bool near = false;
        if (n0 >= 0) {
            float ax, ay, bx, by;
            geo_xy_f32(g, ps, r.lane, r.intent, ax, ay);
            geo_xy_f32(g, sh.p[n0], sh.lane_of[n0], sh.intent_of[n0], bx, by);
            const float fx = bx - ax, fy = by - ay, lim = (float)c.collision_thr + 0.05f;
            near = fx * fx + fy * fy < lim * lim;
        }
        if (near) {
            double ax, ay, bx, by;
            geo_xy(g, ps, r.lane, r.intent, ax, ay);
            geo_xy(g, sh.p[n0], sh.lane_of[n0], sh.intent_of[n0], bx, by);
            const double dx = bx - ax, dy = by - ay;
            const double dxy = sqrt(dx * dx + dy * dy);
            if (fabs(dxy) < c.collision_thr) {
                r.hit = 1;
                lds_add(&sh.cnt[n0], (r.ord < sh.ord[n0]) ? 1 : (1 << 16));   // seen by n0 this tick iff we precede it
            }
        }
    }

    // ============================================================== FX: ordered effects (ref :333-359)
    static PVE_HD void ph_effects(const PVE_AS4 GeoConst &g, int t, Sh &sh, Regs &r)
    {
        const PVE_AS4 Const &c = g.base;
        r.del = 0; r.fin = 0; r.coll_seen = 0; r.coll_fin = 0;
        if (r.alive) {
            const int cc = sh.cnt[t];
            const int prev = (r.meta >> M_COLL_SHIFT) & M_COLL_MASK;
            r.coll_seen = prev + r.hit + (cc & 0xffff);
            r.coll_fin = r.coll_seen + (cc >> 16);
            if (r.ctl && r.coll_seen > 0) lds_add(&sh.acc_collisions, r.coll_seen);
            if (r.p < c.exit_p || r.coll_seen > 0) {
                r.del = 1;
                if (r.coll_seen > 0) {
                    if (r.ctl) r.reward = -10;
                    else {                                     // reward[-1]: the controlled vehicle processed last before us
                        const int po = mask_prev<NW>(sh.m_ctl_ord, r.ord);
                        if (po >= 0) sh.rew_ovr[sh.slot_at[po]] = 1;
                    }
                }
                r.meta |= M_DONE;
                r.hdr = -1; sh.hdr[t] = -1;
            } else if (r.p < 0 && (r.meta & M_CONTROL)) {
                r.fin = 1;
                r.meta |= M_DONE | M_FINISH;
                r.meta &= ~(M_CONTROL | M_LOCK);
                r.hdr = -1; sh.hdr[t] = -1;
                r.reward = 5;
                lds_add(&sh.acc_passed_steps, r.step);
            }
        }
        {                                                 // (after this thread's own resets of hdr[t] above)
            const int h = r.hdr;                          // (= hdr[t]: -1 since the top of SCAN unless SCAN set it)
            sh.chain()[t] = (uint8_t)(h < 0 ? CAP : h);
            if (t == 0) sh.chain()[CAP] = (uint8_t)CAP;
        }
        vote<NW>(sh.m_del, t, r.del);
        vote<NW>(sh.m_fin, t, r.fin);
        vote<NW>(sh.m_ctlnow, t, r.alive && !r.del && (r.meta & M_CONTROL));
        vote<NW>(sh.m_coll, t, r.alive && r.ctl && r.coll_seen > 0);
        vote<NW>(sh.m_spawn, t, t < g.lane_num && sh.hd.current_time >= sh.hd.next_arr[t < NL ? t : 0]);   // ref :379
    }

    // ============================================================== FIN: re-pack + spawn + write-back
    static PVE_HD int pack_lanej(const Sh &sh, int slot) { return Base::pack_lanej(sh, slot); }

    // RES = false: the tick kernel -- state and header go back to HBM (`O` = P.out).  RES = true: k_rollout_geo -- outputs
    // only; the persistent fields and the header updates are handed to ph_stage through `fc` and stay on the chip.
    // k_rollout_geo, `still` ticks (cf. Tick::ph_final): nobody is deleted and nobody spawns -> every vehicle keeps its slot,
    // nothing is staged, the registers carry over; `full` (uniform) forces the staged form (the last tick of a launch).
    template <bool RES, class OutT>
    static PVE_HD void ph_final(const PVE_AS4 GeoConst &g, const PVE_AS4 Params &P, const OutT &O, int env, int t, Sh &sh,
                                Regs &r, FinCarry &fc, bool full = true)
    {
        const PVE_AS4 Const &c = g.base;
        EnvHeader &gh = P.headers[env];
        const int N = sh.hd.n_alive;
        const int LN = g.lane_num;
        const size_t gpre = (size_t)env * CAP + t;
        const bool fused = RES || (P.mode == MODE_FUSED);
        fc.meta = 0;
        const unsigned want = (unsigned)(sh.m_spawn[0] & 0xFFFull);
        unsigned sp = want; int room = CAP - N;
        if (__builtin_popcount(want) > room) {            // (uniform, rare: a full intersection defers the spawns of the higher lanes)
            sp = 0;
#pragma unroll
            for (int l = 0; l < NL; l++) if ((want >> l) & 1) { if (room > 0) { sp |= 1u << l; room--; } }
        }
        const int n_over = __builtin_popcount(want) - __builtin_popcount(sp);
        u64 keep[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) keep[k] = fused ? (sh.m_alive[k] & ~sh.m_del[k]) : sh.m_alive[k];
        bool still = RES && !full && sp == 0;
#pragma unroll
        for (int k = 0; k < NW; k++) still = still && (sh.m_del[k] == 0);
        fc.still = still;
        int meta = 0, hdr_word = -1, new_slot = -1, lockf = 0;
        if (r.alive) {
            int coll = r.coll_fin > M_COLL_MASK ? M_COLL_MASK : r.coll_fin;
            meta = (r.meta & (M_CONTROL | M_FINISH | M_DONE | (M_INT_MASK << M_INT_SHIFT) | (M_LANE_MASK << M_LANE_SHIFT))) | M_ALIVE |
                   (coll << M_COLL_SHIFT);
            if (r.cyc & 1) {                                                           // ref :1493-1497
                const int len = (r.cyc >> 1) & 15, off = sh.cyc_off[r.cyc >> 9];
                double sum = 0;
#pragma unroll
                for (int q = 0; q < 10; q++) if (q < len) sum = sum + sh.u_vd[off + q];
                const int best_o = sh.lk_slot[off];
                meta |= M_LOCK;
                lockf = 1;
                if (sh.u_vd[off] < c.collision_thr || sum / (double)len < c.lock_mean_thr) {
                    if (best_o == t) meta |= M_LOCKA_POS;
                    else if (sh.hdr[best_o] == t) meta |= M_LOCKA_NEG;
                }
            }
            if (r.del) meta |= M_DEL;
            fc.meta = meta;
            hdr_word = pack_lanej(sh, r.hdr);
            if (mask_test(keep, t)) {
                new_slot = mask_below<NW>(keep, t) + __builtin_popcount(sp & ((1u << r.lane) - 1u));
                if (!RES) Base::store_slot(P, env, new_slot, r, meta, hdr_word, new_slot != t, r.ctl || new_slot != t);
                else if (!still) {         // EARLY staging (these registers die here)
                    const int s = new_slot;
                    sh.template stf<Sh::SF_JERK>()[s] = r.jerk; sh.template stf<Sh::SF_JERK_SUM>()[s] = r.jerk_sum;
                    sh.template stf<Sh::SF_VIR_DIS>()[s] = r.vir_dis; sh.template stf<Sh::SF_CLOSER_P>()[s] = r.closer_p;
                    sh.template sti<I_ID>()[s] = r.id; sh.template sti<I_SEQ>()[s] = r.seq; sh.template sti<I_VNUM>()[s] = r.vnum;
                    sh.template sti<I_STEP>()[s] = r.step; sh.template sti<I_COUNT>()[s] = r.count;
                    sh.template sti<I_META>()[s] = meta; sh.template sti<I_HDR>()[s] = hdr_word;
                }
            }
        }
        const int n_post = mask_below<NW>(keep, sh.hd.lane_start[NL]) + __builtin_popcount(sp);
        fc.new_slot = new_slot;
        fc.ls = 0; fc.sp_slot = -1; fc.sp_id = 0; fc.sp_vnum = 0; fc.sp_int = 0;
        if (t <= NL) {
            const int ls = mask_below<NW>(keep, sh.hd.lane_start[t]) + __builtin_popcount(sp & ((1u << t) - 1u));
            if (RES) fc.ls = ls; else gh.lane_start[t] = ls;
        }
        // ---- spawned vehicles (one per lane at most), ref :378-433
        if (t < LN && ((sp >> t) & 1)) {
            const int nth = __builtin_popcount(sp & ((1u << t) - 1u));               // spawns of lower lanes come first
            const int slot = mask_below<NW>(keep, sh.hd.lane_start[t + 1]) + nth;
            int intention;
            if (LN == 4) intention = (sh.hd.intention_re + nth) % 3;                   // ref :385-388
            else if (LN == 8) {                                                        // ref :389-392, draws as input
                const int ch = P.choice ? P.choice[(size_t)env * P.choice_env_stride + (size_t)sh.hd.veh_rec[t] * LN + t] : 0;
                intention = (t & 1) ? (ch ? 2 : 1) : (ch ? 1 : 0);                     // ref :125-134
            } else intention = t % 3;                                                  // ref :393-394
            if (RES) {
                fc.sp_slot = slot; fc.sp_id = sh.hd.id_seq + nth; fc.sp_vnum = sh.hd.lane_start[t + 1] - sh.hd.lane_start[t];
                fc.sp_int = intention;
            } else {
                Regs nv;
                nv.p = sel3(c.spawn_p, intention); nv.v = c.v0; nv.a = 0; nv.jerk = 0; nv.jerk_sum = 0;
                nv.vir_dis = 100; nv.closer_p = 150;
                nv.id = sh.hd.id_seq + nth;
                nv.seq = sh.hd.veh_rec[t];
                nv.vnum = sh.hd.lane_start[t + 1] - sh.hd.lane_start[t];
                nv.step = 0; nv.count = 0;
                Base::store_slot(P, env, slot, nv,
                                 M_CONTROL | M_ALIVE | (t << M_LANE_SHIFT) | (LN == 12 ? 0 : (intention << M_INT_SHIFT)), -1);   // 12-lane: lane % 3, not stored
            }
            if (O.obs_post) {                                                          // ref :380, :420
                if (P.obs_f32) { float *o = env_at<CAP * OBSW>((float *)O.obs_post, env, slot * OBSW); for (int k = 0; k < OBSW; k++) o[k] = 0.0f; }
                else { double *o = env_at<CAP * OBSW>(O.obs_post, env, slot * OBSW); for (int k = 0; k < OBSW; k++) o[k] = 0.0; }
            }
            if (!RES) {
                const int rec1 = sh.hd.veh_rec[t] + 1;
                gh.veh_rec[t] = rec1;
                gh.next_arr[t] = r.next_arr;
            }
        }
        if (!RES && t >= n_post && t < N) { P.i32[I_META][gpre] = 0; P.i32[I_ID][gpre] = -1; }
        if (!RES && t < ND) { gh.head_lane[t] = sh.hd.head_lane[t]; gh.head_j[t] = sh.hd.head_j[t]; }
        const int n_ctl = mask_count<NW>(sh.m_ctl);
        // (the header's counters are thread 0's business: cf. Tick::ph_final)
        fc.n_ctl = n_ctl; fc.n_post = n_post; fc.n_sp = __builtin_popcount(sp);
        if (t == 0) {
            const int n_lock = mask_count<NW>(sh.m_lead);
            const int n_fin = mask_count<NW>(sh.m_fin);
            const int n_del = mask_count<NW>(sh.m_del);
            double sr = 0, sj = 0;
#if PVE_DEVICE_CODE
            for (int k = 0; k < NW; k++) { sr += sh.red_reward[k]; sj += sh.red_jerk[k]; }
#else
            sr = sh.red_reward[0]; sj = sh.red_jerk[0];
#endif
            if (RES) {                                 // (the header's accumulators: cf. Tick::ph_final)
                sh.hd.passed += n_fin;
                sh.hd.passed_step_total += sh.acc_passed_steps;
                sh.hd.sum_reward = sh.hd.sum_reward + sr;
                sh.hd.sum_jerk = sh.hd.sum_jerk + sj;
                sh.hd.alive_steps += N;
                sh.hd.ctl_steps += n_ctl;
                sh.hd.ticks += 1;
                sh.hd.collided += mask_count<NW>(sh.m_coll);
                sh.hd.locks += n_lock;
                sh.hd.overflow += n_over;
            }
          if (!RES) {
            gh.current_time = sh.hd.current_time;
            gh.n_alive = n_post;
            gh.id_seq = sh.hd.id_seq + __builtin_popcount(sp);
            if (LN != 12) gh.intention_re = sh.hd.intention_re + __builtin_popcount(sp);   // ref :388, :392
            gh.passed = sh.hd.passed + n_fin;
            gh.passed_step_total = sh.hd.passed_step_total + sh.acc_passed_steps;
            gh.head_valid = sh.hd.head_valid;
            gh.sum_reward = sh.hd.sum_reward + sr;
            gh.sum_jerk = sh.hd.sum_jerk + sj;
            gh.alive_steps = sh.hd.alive_steps + N;
            gh.ctl_steps = sh.hd.ctl_steps + n_ctl;
            gh.ticks = sh.hd.ticks + 1;
            gh.collided = sh.hd.collided + mask_count<NW>(sh.m_coll);
            gh.locks = sh.hd.locks + n_lock;
            gh.overflow = sh.hd.overflow + n_over;
          }
            if (O.env_out) {
                int *eo = O.env_out + (size_t)env * 8;
                eo[0] = N; eo[1] = n_ctl; eo[2] = sh.acc_collisions;
                eo[3] = n_lock; eo[4] = n_del; eo[5] = n_fin; eo[6] = __builtin_popcount(sp); eo[7] = n_post;
            }
        }
        if (O.flags) {
            int f = 0;
            if (r.alive) {
                f = 0x01 | (r.ctl ? 0x02 : 0) | ((r.meta & M_DONE) ? 0x04 : 0) | (r.del ? 0x08 : 0) |
                    (r.fin ? 0x10 : 0) | (lockf ? 0x20 : 0) | (r.intent << 6) | (r.ctl ? (r.coll_seen << 8) : 0);
            }
            *env_at<CAP>(O.flags, env, t) = f;
        }
        if (O.reward && r.alive) *env_at<CAP>(O.reward, env, t) = r.ctl ? r.reward : 0.0;
        if (O.lanej && r.alive) *env_at<CAP>(O.lanej, env, t) = (r.lane << 16) | r.j;
        if (O.new_slot && r.alive) *env_at<CAP>(O.new_slot, env, t) = new_slot;
        if (r.alive && r.ctl && (O.nbr || O.obs_pre || (O.obs_post && new_slot >= 0))) {
            // the 6 neighbours' speed, acceleration, lane and route, then their lane starts: two batches of unconditional LDS
            // gathers on clamped slots (a guarded block per neighbour is a chain of six round trips), shared by the ids and the row
            constexpr bool ZROW = RES && PVE_GEO_ZROW;   // (the single-tick kernel at 96 VGPR would spill the longer live ranges)
            int xc[NNB], nln[NNB], nrt[NNB], nls[NNB]; double nv[NNB], na[NNB];
#pragma unroll
            for (int k = 0; k < NNB; k++) {
                xc[k] = r.kr[k] < 0 ? (ZROW ? CAP : 0) : r.kr[k];
                nln[k] = (int)sh.lane_of[xc[k]]; nrt[k] = (int)sh.route_of[xc[k]];
                nv[k] = sh.v[xc[k]]; na[k] = sh.a[xc[k]];
            }
#pragma unroll
            for (int k = 0; k < NNB; k++) { PVE_PIN(nln[k]); PVE_PIN(nrt[k]); PVE_PIN(nv[k]); PVE_PIN(na[k]); }
#pragma unroll
            for (int k = 0; k < NNB; k++) nls[k] = sh.hd.lane_start[nln[k]];
#pragma unroll
            for (int k = 0; k < NNB; k++) PVE_PIN(nls[k]);
            if (O.nbr) {                               // controlled vehicles only (PVE_F_CTL in flags)
                int *nb = env_at<CAP * NNB>(O.nbr, env, t * NNB);
#pragma unroll
                for (int k = 0; k < NNB; k++) nb[k] = r.kr[k] < 0 ? -1 : ((nln[k] << 16) | (xc[k] - nls[k]));
            }
          if (O.obs_pre || (O.obs_post && new_slot >= 0)) {
            double row[OBSW];                                                          // ref :1325-1337
            row[0] = r.p; row[1] = r.v; row[2] = r.a; row[3] = (double)r.route;
#pragma unroll
            for (int k = 0; k < NNB; k++) {
                const bool has = r.kr[k] >= 0;
                // (absent neighbour: the scan left kv = 0, the gathers came from the zero cell)
                row[4 + 4 * k] = (ZROW || has) ? r.kv[k] : 0.0; row[5 + 4 * k] = (ZROW || has) ? nv[k] : 0.0;
                row[6 + 4 * k] = (ZROW || has) ? na[k] : 0.0; row[7 + 4 * k] = (ZROW || has) ? (double)nrt[k] : 0.0;
            }
            if (O.obs_pre) {
                // obs_pre AND obs_post in ONE pass over the row: every value goes to both rows and dies (two passes keep all 28
                // values live across both: 44-110 spilled registers in the training variants).  A vehicle that leaves has no
                // obs_post row: its second store rewrites the obs_pre element (same value, same address: harmless)
                const bool wpost = O.obs_post && new_slot >= 0;
                if (P.obs_f32) {                        // (obs_pre / state_pre follow the row type, as in the 12-lane kernels)
                    float *o = env_at<CAP * OBSW>((float *)O.obs_pre, env, t * OBSW);
                    float *q = wpost ? env_at<CAP * OBSW>((float *)O.obs_post, env, new_slot * OBSW) : o;
#pragma unroll
                    for (int k = 0; k < OBSW; k++) { const float x = (float)row[k]; o[k] = x; q[k] = x; }
                } else {
                    double *o = env_at<CAP * OBSW>(O.obs_pre, env, t * OBSW);
                    double *q = wpost ? env_at<CAP * OBSW>(O.obs_post, env, new_slot * OBSW) : o;
#pragma unroll
                    for (int k = 0; k < OBSW; k++) { o[k] = row[k]; q[k] = row[k]; }
                }
            } else if (O.obs_post && new_slot >= 0) {
                if (P.obs_f32) {
                    float *o = env_at<CAP * OBSW>((float *)O.obs_post, env, new_slot * OBSW);
#pragma unroll
                    for (int k = 0; k < OBSW; k++) o[k] = (float)row[k];
                } else {
                    double *o = env_at<CAP * OBSW>(O.obs_post, env, new_slot * OBSW);
#pragma unroll
                    for (int k = 0; k < OBSW; k++) o[k] = row[k];
                }
            }
          }
        }
    }

    static PVE_HD void ph_final(const PVE_AS4 GeoConst &g, const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r)
    {
        FinCarry fc;
        ph_final<false>(g, P, P.out, env, t, sh, r, fc);
    }

    // ============================================================== k_rollout_geo: many ticks per launch (cf. Tick::ph_stage ..)
    // next tick's action of slot t stays in a register (the slot's own thread consumes it after RELOAD)
    static PVE_HD void ph_stage(const PVE_AS4 GeoConst &g, int t, Sh &sh, Regs &r, const FinCarry &fc)
    {
        const PVE_AS4 Const &c = g.base;
        if (fc.new_slot >= 0) {                          // LATE staging (the rest went at the top of FIN)
            const int s = fc.new_slot;
            sh.template stf<Sh::SF_P>()[s] = r.p; sh.template stf<Sh::SF_V>()[s] = r.v; sh.template stf<Sh::SF_A>()[s] = r.a;
        }
        if (fc.sp_slot >= 0) {                           // t < lane_num: the vehicle lane t spawns (ref :395-433)
            const int s = fc.sp_slot;
            sh.template stf<Sh::SF_P>()[s] = sel3(c.spawn_p, fc.sp_int); sh.template stf<Sh::SF_V>()[s] = c.v0;
            sh.template stf<Sh::SF_A>()[s] = 0; sh.template stf<Sh::SF_JERK>()[s] = 0; sh.template stf<Sh::SF_JERK_SUM>()[s] = 0;
            sh.template stf<Sh::SF_VIR_DIS>()[s] = 100; sh.template stf<Sh::SF_CLOSER_P>()[s] = 150;
            sh.template sti<I_ID>()[s] = fc.sp_id; sh.template sti<I_SEQ>()[s] = sh.hd.veh_rec[t];
            sh.template sti<I_VNUM>()[s] = fc.sp_vnum; sh.template sti<I_STEP>()[s] = 0; sh.template sti<I_COUNT>()[s] = 0;
            sh.template sti<I_META>()[s] = M_CONTROL | M_ALIVE | (t << M_LANE_SHIFT) | (g.lane_num == 12 ? 0 : (fc.sp_int << M_INT_SHIFT));
            sh.template sti<I_HDR>()[s] = -1;
            sh.hd.veh_rec[t] += 1;
            sh.hd.next_arr[t] = r.next_arr;
        }
        if (t <= NL) sh.hd.lane_start[t] = fc.ls;
        if (t == 0 && g.lane_num != 12) sh.hd.intention_re += fc.n_sp;               // ref :388, :392
        Base::ph_stage_header(t, sh, fc);
    }
    // a still tick: the vehicle stays in the registers, only the flags word and the next action change hands
    static PVE_HD void ph_carry_over(int t, Sh &sh, Regs &r, const FinCarry &fc)
    {
        Base::ph_stage_header(t, sh, fc);
        r.meta = fc.meta;
        r.act = r.act_nx;
    }
    // RELOAD (after barrier B): slot t's vehicle from the staging arrays, its action from the prefetch register
    static PVE_HD void ph_reload(int t, Sh &sh, Regs &r)
    {
        const int N = sh.hd.n_alive;
        r.alive = t < N;
        r.jerk = 0;
        r.p = r.v = r.a = r.jerk_sum = r.vir_dis = r.closer_p = 0;
        r.id = r.seq = r.vnum = r.step = r.count = r.meta = 0;
        if (t < N) {
            r.p = sh.template stf<Sh::SF_P>()[t]; r.v = sh.template stf<Sh::SF_V>()[t]; r.a = sh.template stf<Sh::SF_A>()[t];
            r.jerk_sum = sh.template stf<Sh::SF_JERK_SUM>()[t]; r.vir_dis = sh.template stf<Sh::SF_VIR_DIS>()[t];
            r.closer_p = sh.template stf<Sh::SF_CLOSER_P>()[t];
            r.id = sh.template sti<I_ID>()[t]; r.seq = sh.template sti<I_SEQ>()[t]; r.vnum = sh.template sti<I_VNUM>()[t];
            r.step = sh.template sti<I_STEP>()[t]; r.count = sh.template sti<I_COUNT>()[t]; r.meta = sh.template sti<I_META>()[t];
        }
        r.act = r.act_nx;
    }
    // work-array initialisation of a resident tick (what LOAD does besides loading), after the barrier behind RELOAD
    static PVE_HD void ph_tick_init(const PVE_AS4 GeoConst &g, int t, Sh &sh, Regs &r)
    {
        sh.cnt[t] = 0; sh.rew_ovr[t] = 0; sh.hdr[t] = -1;
        if (t == 0) {
            sh.acc_passed_steps = 0; sh.acc_collisions = 0; sh.lead_n = 0;
            sh.hd.current_time = sh.hd.current_time + g.base.deltaT;                  // ref :223 (repeated +=)
        }
    }
    // the per-route / per-list counters of the NEXT tick's list build (S1 counts the routes with atomics): cleared in the FX
    // phase, behind the barrier that ends their last use (WALK), so that the loop needs no barrier between TICK_INIT and S1
    static PVE_HD void ph_lists_clear(int t, Sh &sh)
    {
        if (t < ND) { sh.rc[t] = 0; sh.rfill[t] = 0; sh.fill[t] = 0; sh.cnt2[t] = 0; sh.ofill[t] = 0; }
    }

    // ============================================================== STATE: 7x28; a neighbour's row is this tick's
    // if it was processed before us (order, not slot), else the row it stored last tick (ref :1332).
    // ph_state_order (LOCK phase, only when state_pre is requested): which of the 6 neighbours precede this vehicle in the
    // processing order -- read HERE because k_rollout_geo's FIN re-uses `ord` as staging storage; the bits ride in r.mmask
    // (dead since PAIRS).
    static PVE_HD void ph_state_order(int t, const Sh &sh, Regs &r)
    {
        int fresh = 0;
        if (r.alive && r.ctl) {
#pragma unroll
            for (int q = 0; q < NNB; q++) {
                const int x = r.kr[q];
                fresh |= (x >= 0 && sh.ord[x < 0 ? 0 : x] < r.ord) ? (1 << q) : 0;
            }
        }
        r.mmask = fresh;
    }
    // COH (persistent roll-out, cf. Tick::state_rows): the stale rows of an item's first tick were stored by ANOTHER workgroup of
    // this launch (block k - 1 of the trajectory belongs to the previous item of the intersection): coherent loads
    template <class ROW, bool COH, class OutT>
    static PVE_HD void state_rows(const OutT &O, size_t base, int t, const Regs &r)
    {
        ROW *dst = (ROW *)O.state_pre + (base + t) * (size_t)((NNB + 1) * OBSW);
        const ROW *pre = (const ROW *)O.obs_pre, *prev = (const ROW *)O.obs_prev_post;
        const ROW *srcs[NNB + 1];
        unsigned coh = 0;
        srcs[0] = pre + (base + t) * OBSW;
#pragma unroll
        for (int q = 0; q < NNB; q++) {
            const int x = r.kr[q];
            const bool fresh = (r.mmask >> q) & 1;
            srcs[q + 1] = x < 0 ? (const ROW *)nullptr : ((fresh ? pre : prev) + (base + x) * OBSW);
            if (COH && x >= 0 && !fresh) coh |= 2u << q;
        }
        gather_state<ROW>(srcs, coh, dst);
    }
    template <bool COH = false, class OutT>
    static PVE_HD void ph_state(const PVE_AS4 Params &P, const OutT &O, int env, int t, Sh &sh, Regs &r)
    {
        if (!O.state_pre || !(r.alive && r.ctl)) return;
        if (P.obs_f32) state_rows<float, COH>(O, (size_t)env * CAP, t, r);
        else state_rows<double, COH>(O, (size_t)env * CAP, t, r);
    }
    static PVE_HD void ph_state(const PVE_AS4 Params &P, int env, int t, Sh &sh, Regs &r) { ph_state(P, P.out, env, t, sh, r); }
};

// ================================================================== reset / warm-up, ref :196-220
template <int CAP>
PVE_HD void reset_env_geo(const PVE_AS4 GeoConst &g, const PVE_AS4 Params &P, int env, int cap_ticks)
{
    const PVE_AS4 Const &c = g.base;
    const int LN = g.lane_num;
    EnvHeader h;
    {
        int *z = (int *)&h;
        for (int w = 0; w < (int)(sizeof(EnvHeader) / 4); w++) z[w] = 0;
    }
    for (int l = 0; l < ND; l++) { h.head_lane[l] = -1; h.head_j[l] = -1; }
    const double *arr = P.arrivals + (size_t)env * P.arr_env_stride;
    int n = 0;
    int lane_of[NL], int_of[NL];
    for (int it = 0; it < cap_ticks && n == 0; it++) {
        h.current_time += c.deltaT;
        for (int l = 0; l < LN; l++) {
            if (h.veh_rec[l] < P.rows && h.current_time >= arr[(size_t)h.veh_rec[l] * LN + l] && n < CAP) {
                int intention;
                if (LN == 4) { intention = h.intention_re % 3; h.intention_re += 1; }
                else if (LN == 8) {
                    const int ch = P.choice ? P.choice[(size_t)env * P.choice_env_stride + (size_t)h.veh_rec[l] * LN + l] : 0;
                    intention = (l & 1) ? (ch ? 2 : 1) : (ch ? 1 : 0);
                    h.intention_re += 1;
                } else intention = l % 3;
                lane_of[n] = l; int_of[n] = intention;
                n++;
                h.veh_rec[l] += 1;
            }
        }
    }
    for (int s = 0; s < CAP; s++) {
        size_t gi = (size_t)env * CAP + s;
        if (s < n) {
            P.f64[F_P][gi] = c.spawn_p[int_of[s]]; P.f64[F_V][gi] = c.v0; P.f64[F_A][gi] = 0; P.f64[F_JERK][gi] = 0;
            P.f64[F_JERK_SUM][gi] = 0; P.f64[F_VIR_DIS][gi] = 100; P.f64[F_CLOSER_P][gi] = 150;
            P.i32[I_ID][gi] = s; P.i32[I_SEQ][gi] = 0; P.i32[I_VNUM][gi] = 0; P.i32[I_STEP][gi] = 0;
            P.i32[I_COUNT][gi] = 0; P.i32[I_HDR][gi] = -1;
            P.i32[I_META][gi] = M_CONTROL | M_ALIVE | (lane_of[s] << M_LANE_SHIFT) | (LN == 12 ? 0 : (int_of[s] << M_INT_SHIFT));
        } else {
            P.f64[F_P][gi] = 0; P.f64[F_V][gi] = 0; P.f64[F_A][gi] = 0; P.f64[F_JERK][gi] = 0;
            P.f64[F_JERK_SUM][gi] = 0; P.f64[F_VIR_DIS][gi] = 0; P.f64[F_CLOSER_P][gi] = 0;
            P.i32[I_ID][gi] = -1; P.i32[I_SEQ][gi] = 0; P.i32[I_VNUM][gi] = 0; P.i32[I_STEP][gi] = 0;
            P.i32[I_COUNT][gi] = 0; P.i32[I_META][gi] = 0; P.i32[I_HDR][gi] = -1;
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
        h.next_arr[l] = (l < LN && h.veh_rec[l] < P.rows) ? arr[(size_t)h.veh_rec[l] * LN + l] : INFINITY;
    h.n_alive = n;
    h.id_seq = n;
    P.headers[env] = h;
}

}  // namespace pve
