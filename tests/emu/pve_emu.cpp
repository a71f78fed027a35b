// pve_emu.cpp -- CPU *test* emulator of the HIP kernels (NOT a product path, never shipped in
// libpveenv.so, never selected by the package on its own).
//
// It compiles the very same phase bodies (csrc/pve_tick_core.h) with g++ and executes them the
// way the GPU does -- every phase for all threads t = 0..CAP-1 of a workgroup, then the next
// phase (= a workgroup barrier) -- behind the same C ABI (csrc/pve_capi.inc) on host memory.
// The CPU test-suite uses it to check the parallel formulation of the tick against the
// sequential oracle without a GPU; the `-m gpu` tests run the real kernels.
#include <new>
#include <string>
#include <vector>

#include "../../pve-mcc_for_unsignalized_intersection_amd/csrc/pve_host.h"
#include "../../pve-mcc_for_unsignalized_intersection_amd/csrc/pve_tick_core.h"
#include "../../pve-mcc_for_unsignalized_intersection_amd/csrc/pve_tick_geo.h"
#include "../../pve-mcc_for_unsignalized_intersection_amd/csrc/pve_actor.h"

using namespace pve;

template <int CAP> static void emu_tick(const Const &c, const Params &P)
{
    typedef Tick<CAP> T;
    std::vector<Regs> regs(CAP);
    Shared<CAP> *shp = new Shared<CAP>();
    for (int env = 0; env < P.n_envs; env++) {
        Shared<CAP> &sh = *shp;
        memset(&sh, 0, sizeof(sh));
        for (int t = 0; t < CAP; t++) T::ph_load(c, P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_step1(c, P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_step2(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_lists_a(c, t, sh);
        for (int t = 0; t < CAP; t++) T::ph_step3(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_step3_publish(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_lists_b(t, sh);
        for (int t = 0; t < CAP; t++) T::ph_build(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_rank(t, sh, env);
        for (int t = 0; t < CAP; t++) T::ph_scan(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_reward(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_effects(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_prefetch_arrival(P, env, t, sh, regs[t], NL);
        for (int t = 0; t < CAP; t++) T::ph_lock(c, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_lock2(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_final(c, P, env, t, sh, regs[t]);
        if (P.out.state_pre) {                       // (the 7 x 28 states: descriptors, barrier, cooperative write)
            for (int t = 0; t < CAP; t++) T::ph_state_publish(P.out, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_state_coop(P, P.out, env, t, sh);
        }
    }
    delete shp;
}

// k_rollout: the resident multi-tick form, phase by phase exactly as the kernel orders them
// k_base: first tick of this launch / queue item within the call's output blocks (0 for a chunked launch, whose Params are
// shifted to its first block already; the item's first tick for the sequentially emulated work queue)
// ShT: Shared<CAP> (the default block), or the HOME block of k_rollout<128, 5, ..> (carried fields in LDS homes, entry pool of
// 304 entries worked in passes; PVE_EMU_HOME=1) -- PVE_EMU_HOME=2 takes a still smaller pool so that ordinary traffic runs
// through the multi-pass form
static int g_emu_max_passes = 0, g_emu_max_entries = 0; // (diagnostics of the tests: the most passes / list entries a tick has taken)
extern "C" int pve_emu_max_passes(void) { const int v = g_emu_max_passes; g_emu_max_passes = 0; return v; }
extern "C" int pve_emu_max_entries(void) { const int v = g_emu_max_entries; g_emu_max_entries = 0; return v; }
template <int CAP, class ShT = Shared<CAP>> static void emu_rollout(const Const &c, const Params &P, const RolloutArgs &R, int k_base = 0)
{
    typedef Tick<CAP, ShT> T;
    std::vector<Regs> regs(CAP);
    std::vector<FinCarry> fcs(CAP);
    std::vector<HomeRegs> hrs(CAP);
    ShT *shp = new ShT();
    for (int env = 0; env < P.n_envs; env++) {
        ShT &sh = *shp;
        memset(&sh, 0, sizeof(sh));
        int pool_idx = R.pool_tick0;
        const bool idt = R.source == 3;                      // PVE_SRC_TABLE: actions by (tick, vehicle id)
        auto tab = [&](int row, int id) { return R.pool[(size_t)row * (size_t)R.table_ids + (id < 0 ? 0 : (id < R.table_ids ? id : R.table_ids - 1))]; };
        std::vector<double> sp_act(CAP, 0.0);
        for (int t = 0; t < CAP; t++) T::ph_load(c, P, env, t, sh, regs[t]);
        if (idt) for (int t = 0; t < CAP; t++) regs[t].act = regs[t].alive ? tab(pool_idx, regs[t].id) : 0.0;
        for (int t = 0; t < CAP; t++) T::ph_home_store(t, sh, regs[t]);
        if (ShT::HOME)                                   // (the kernel's registers hold nothing of these from here on: poison them)
            for (int t = 0; t < CAP; t++) {
                Regs &q = regs[t];
                q.p = q.v = q.a = q.jerk_sum = q.vir_dis = q.closer_p = q.act = NAN; q.id = q.seq = q.vnum = q.count = -12345;
            }
        for (int k = 0; k < R.n_ticks; k++) {
            const bool last_tick = k + 1 == R.n_ticks;
            for (int w = 0; w < CAP / 64; w++)      // the emulator's vote() ORs bits: start every tick from empty masks
                sh.m_alive[w] = sh.m_ctl[w] = sh.m_del[w] = sh.m_fin[w] = sh.m_ctlnow[w] = sh.m_coll[w] = sh.m_lead[w] = sh.m_spawn[w] = 0;
            sh.emu_scan = 0;
            if (k > 0) for (int t = 0; t < CAP; t++) T::ph_tick_init(c, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_step1(c, P, env, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_step2(c, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_lists_a(c, t, sh);
            for (int t = 0; t < CAP; t++) T::ph_step3(c, t, sh, regs[t], last_tick);
            for (int t = 0; t < CAP; t++) T::ph_step3_publish(t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_lists_b(t, sh);
            if (ShT::HOME) {                         // (k_rollout<128, 5, ..>: BUILD .. WALK in passes over groups of lists)
                for (int t = 0; t < CAP; t++) T::ph_build_prep(c, t, sh, regs[t]);
                for (int t = 0; t < CAP; t++) T::ph_scan_init(regs[t]);
                int pass = 0;
                if (sh.loff[NL] <= ShT::POOL) {          // (as the kernel: ONE pass is the ungrouped code)
                    for (int t = 0; t < CAP; t++) T::template ph_build_fill<false>(c, t, sh, regs[t], 0, NL);
                    for (int t = 0; t < CAP; t++) T::template ph_rank<false>(t, sh);
                    for (int t = 0; t < CAP; t++) T::template ph_scan_lists<false>(c, t, sh, regs[t], 0, NL);
                    pass = 1;
                } else
                for (int d0 = 0; d0 < NL; pass++) {
                    const int d1 = T::group_end(sh, d0);
                    for (int t = 0; t < CAP; t++) T::template ph_build_fill<true>(c, t, sh, regs[t], d0, d1);
                    for (int t = 0; t < CAP; t++) T::template ph_rank<true>(t, sh, pass, d0, d1);
                    for (int t = 0; t < CAP; t++) T::template ph_scan_lists<true>(c, t, sh, regs[t], d0, d1);
                    d0 = d1;
                }
                if (pass > g_emu_max_passes) g_emu_max_passes = pass;
                if (sh.loff[NL] > g_emu_max_entries) g_emu_max_entries = sh.loff[NL];
            } else {
            for (int t = 0; t < CAP; t++) T::ph_build(c, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_rank(t, sh);
            for (int t = 0; t < CAP; t++) T::ph_scan(c, t, sh, regs[t]);
            }
            for (int t = 0; t < CAP; t++) T::ph_reward(c, t, sh, regs[t]);
            int nx = -1;
            if (k + 1 < R.n_ticks) { pool_idx = (pool_idx + 1 == R.n_pool) ? 0 : pool_idx + 1; nx = pool_idx; }
            if (idt) for (int t = 0; t < CAP; t++) {
                int my_id = regs[t].id;
                if constexpr (ShT::HOME) my_id = sh.h_id[t];
                regs[t].act_nx = (nx >= 0 && regs[t].alive) ? tab(nx, my_id) : 0.0;
            }
            else for (int t = 0; t < CAP; t++) T::ph_prefetch_action(P, R, env, t, nx, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_effects(c, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_prefetch_arrival(P, env, t, sh, regs[t], NL);
            if (idt) {                               // first action of the vehicles spawned at the end of this tick
                const unsigned want = (unsigned)(sh.m_spawn[0] & 0xFFFull);
                unsigned sp = 0; int room = CAP - sh.hd.n_alive;
                for (int l = 0; l < NL; l++) if ((want >> l) & 1) { if (room > 0) { sp |= 1u << l; room--; } }
                for (int t = 0; t < NL; t++)
                    sp_act[t] = (((sp >> t) & 1) && nx >= 0) ? tab(nx, sh.hd.id_seq + __builtin_popcount(sp & ((1u << t) - 1u))) : 0.0;
            }
            for (int t = 0; t < CAP; t++) T::ph_lock(c, t, sh, regs[t], last_tick);
            for (int t = 0; t < CAP; t++) T::ph_lock2(t, sh, regs[t], last_tick);
            for (int t = 0; t < CAP; t++) T::ph_keep_prefix(t, sh);
            if (!idt && !(ShT::HOME && last_tick)) for (int t = 0; t < CAP; t++) T::ph_park_action(t, sh, regs[t]);
            const Outputs O = T::tick_outputs(P, R, k_base + k);
            for (int t = 0; t < CAP; t++)
                T::template ph_final<true>(c, P, O, env, t, sh, regs[t], fcs[t], k + 1 == R.n_ticks || O.state_pre != nullptr);
            if (idt) for (int t = 0; t < CAP; t++) if (fcs[t].new_slot >= 0) sh.act_next[fcs[t].new_slot] = regs[t].act_nx;
            for (int t = 0; t < CAP; t++) T::ph_home_take(t, sh, regs[t], fcs[t], hrs[t]);
            if (fcs[0].still) {                       // (uniform) nobody moves: the registers carry over
                for (int t = 0; t < CAP; t++) T::ph_stage_header(t, sh, fcs[t]);
                for (int t = 0; t < CAP; t++) T::ph_carry_over(t, sh, regs[t], fcs[t]);
            } else {
                if (O.state_pre) {
                    if constexpr (!ShT::HOME) {
                        for (int t = 0; t < CAP; t++) T::ph_state_publish(O, t, sh, regs[t]);
                        for (int t = 0; t < CAP; t++) T::ph_state_coop(P, O, env, t, sh);
                    }
                }
                for (int t = 0; t < CAP; t++) T::ph_stage(c, t, sh, regs[t], fcs[t]);
                for (int t = 0; t < CAP; t++) T::ph_home_put(t, sh, fcs[t], hrs[t]);
                if (idt) for (int t = 0; t < NL; t++) if (fcs[t].sp_slot >= 0) sh.act_next[fcs[t].sp_slot] = sp_act[t];
                if (k + 1 < R.n_ticks) for (int t = 0; t < CAP; t++) T::ph_reload(t, sh, regs[t]);
            }
        }
        for (int t = 0; t < CAP; t++) T::ph_flush(P, env, t, sh);
    }
    delete shp;
}

template <int CAP> static void emu_tick_geo(const GeoConst &g, const Params &P)
{
    typedef TickGeo<CAP> T;
    typedef Tick<CAP, SharedGeo<CAP>> B;
    std::vector<Regs> regs(CAP);
    SharedGeo<CAP> *shp = new SharedGeo<CAP>();
    for (int env = 0; env < P.n_envs; env++) {
        SharedGeo<CAP> &sh = *shp;
        memset(&sh, 0, sizeof(sh));
        for (int t = 0; t < CAP; t++) T::ph_load(g, P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_step1(g, P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) B::ph_step2(g.base, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_order(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) B::ph_step3(g.base, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) B::ph_step3_publish(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_order2(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_pairs_mode(t, sh, P.geo_scan != 0);
        if (T::pairs_over(sh, P.geo_scan != 0)) {
            for (int t = 0; t < CAP; t++) T::ph_pairs_count(g, t, sh, P.geo_scan != 0);
            for (int t = 0; t < CAP; t++) T::ph_pairs_exact(t, sh, P.geo_scan != 0);
            for (int t = 0; t < CAP; t++) T::ph_pairs_apply(t, sh);
        }
        for (int t = 0; t < CAP; t++) T::ph_pairs_fill(g, t, sh);
        for (int t = 0; t < CAP; t++) T::ph_fix_table(g, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_load_late(P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_rank(t, sh, env);
        for (int t = 0; t < CAP; t++) T::ph_scan(g, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_reward(g, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_effects(g, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) B::ph_prefetch_arrival(P, env, t, sh, regs[t], g.lane_num);
        for (int t = 0; t < CAP; t++) B::ph_lock_slot(g.base, t, sh, regs[t]);
        if (P.out.state_pre) for (int t = 0; t < CAP; t++) T::ph_state_order(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) B::ph_lock2_slot(t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_final(g, P, env, t, sh, regs[t]);
        if (P.out.state_pre)
            for (int t = 0; t < CAP; t++) T::ph_state(P, env, t, sh, regs[t]);
    }
    delete shp;
}

// k_rollout_geo: the resident multi-tick form of the general-geometry tick, phase by phase as the kernel orders them
template <int CAP> static void emu_rollout_geo(const GeoConst &g, const Params &P, const RolloutArgs &R, int k_base = 0)
{
    typedef TickGeo<CAP> T;
    typedef Tick<CAP, SharedGeo<CAP>> B;
    std::vector<Regs> regs(CAP);
    std::vector<FinCarry> fcs(CAP);
    SharedGeo<CAP> *shp = new SharedGeo<CAP>();
    const bool fix4 = g.lane_num == 4;
    for (int env = 0; env < P.n_envs; env++) {
        SharedGeo<CAP> &sh = *shp;
        memset(&sh, 0, sizeof(sh));
        int pool_idx = R.pool_tick0;
        const bool idt = R.source == 3;                      // PVE_SRC_TABLE: actions by (tick, vehicle id)
        auto tab = [&](int row, int id) { return R.pool[(size_t)row * (size_t)R.table_ids + (id < 0 ? 0 : (id < R.table_ids ? id : R.table_ids - 1))]; };
        std::vector<double> sp_act(CAP, 0.0);
        for (int t = 0; t < CAP; t++) T::ph_load(g, P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) T::ph_load_late(P, env, t, sh, regs[t]);
        if (idt) for (int t = 0; t < CAP; t++) regs[t].act = regs[t].alive ? tab(pool_idx, regs[t].id) : 0.0;
        for (int k = 0; k < R.n_ticks; k++) {
            for (int w = 0; w < CAP / 64; w++) {     // the emulator's vote() ORs bits: start every tick from empty masks
                sh.m_alive[w] = sh.m_ctl[w] = sh.m_del[w] = sh.m_fin[w] = sh.m_ctlnow[w] = sh.m_coll[w] = sh.m_lead[w] = sh.m_spawn[w] = 0;
                sh.m_keep[w] = sh.m_ctl_ord[w] = sh.m_int[0][w] = sh.m_int[1][w] = sh.m_int[2][w] = 0;
            }
            sh.red_reward[0] = 0; sh.red_jerk[0] = 0;
            if (k > 0) for (int t = 0; t < CAP; t++) T::ph_tick_init(g, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_step1(g, P, env, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) B::ph_step2(g.base, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_order(t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) B::ph_step3(g.base, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) B::ph_step3_publish(t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_order2(t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_pairs_mode(t, sh, P.geo_scan != 0);
            if (T::pairs_over(sh, P.geo_scan != 0)) {
                for (int t = 0; t < CAP; t++) T::ph_pairs_count(g, t, sh, P.geo_scan != 0);
                for (int t = 0; t < CAP; t++) T::ph_pairs_exact(t, sh, P.geo_scan != 0);
                for (int t = 0; t < CAP; t++) T::ph_pairs_apply(t, sh);
            }
            for (int t = 0; t < CAP; t++) T::ph_pairs_fill(g, t, sh);
            for (int t = 0; t < CAP; t++) T::ph_fix_table(g, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_rank(t, sh);
            if (fix4) for (int t = 0; t < CAP; t++) T::template ph_scan<true>(g, t, sh, regs[t]);
            else for (int t = 0; t < CAP; t++) T::template ph_scan<false>(g, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_reward(g, t, sh, regs[t]);
            int nx = -1;
            if (k + 1 < R.n_ticks) { pool_idx = (pool_idx + 1 == R.n_pool) ? 0 : pool_idx + 1; nx = pool_idx; }
            if (idt) for (int t = 0; t < CAP; t++) regs[t].act_nx = (nx >= 0 && regs[t].alive) ? tab(nx, regs[t].id) : 0.0;
            else for (int t = 0; t < CAP; t++) B::ph_prefetch_action(P, R, env, t, nx, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_effects(g, t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) T::ph_lists_clear(t, sh);
            for (int t = 0; t < CAP; t++) B::ph_prefetch_arrival(P, env, t, sh, regs[t], g.lane_num);
            for (int t = 0; t < CAP; t++) B::ph_lock_slot(g.base, t, sh, regs[t]);
            const Outputs O = B::template tick_outputs<true>(P, R, k_base + k);
            if (O.state_pre) for (int t = 0; t < CAP; t++) T::ph_state_order(t, sh, regs[t]);
            for (int t = 0; t < CAP; t++) B::ph_lock2_slot(t, sh, regs[t]);
            for (int t = 0; t < CAP; t++)
                T::template ph_final<true>(g, P, O, env, t, sh, regs[t], fcs[t], k + 1 == R.n_ticks || O.state_pre != nullptr);
            if (idt) {
                for (int t = 0; t < CAP; t++) if (!fcs[t].still && fcs[t].new_slot >= 0) sh.p[fcs[t].new_slot] = regs[t].act_nx;
                for (int t = 0; t < CAP; t++)
                    sp_act[t] = (fcs[t].sp_slot >= 0 && nx >= 0) ? tab(nx, fcs[t].sp_id) : 0.0;
            }
            if (fcs[0].still) {                       // (uniform) nobody moves: the registers carry over
                for (int t = 0; t < CAP; t++) T::ph_carry_over(t, sh, regs[t], fcs[t]);
            } else {
                if (O.state_pre) for (int t = 0; t < CAP; t++) T::ph_state(P, O, env, t, sh, regs[t]);
                for (int t = 0; t < CAP; t++) T::ph_stage(g, t, sh, regs[t], fcs[t]);
                if (idt) for (int t = 0; t < CAP; t++) if (fcs[t].sp_slot >= 0) sh.p[fcs[t].sp_slot] = sp_act[t];
                if (k + 1 < R.n_ticks) {
                    for (int t = 0; t < CAP; t++) T::ph_reload(t, sh, regs[t]);
                    if (idt) for (int t = 0; t < CAP; t++) regs[t].act = sh.p[t];
                }
            }
        }
        for (int t = 0; t < CAP; t++) B::ph_flush(P, env, t, sh);
    }
    delete shp;
}

template <int CAP> static void emu_compact(const Params &P)
{
    std::vector<CRegs> regs(CAP);
    Shared<CAP> *shp = new Shared<CAP>();
    for (int env = 0; env < P.n_envs; env++) {
        Shared<CAP> &sh = *shp;
        memset(&sh, 0, sizeof(sh));
        for (int t = 0; t < CAP; t++) Tick<CAP>::ph_c_load(P, env, t, sh, regs[t]);
        for (int t = 0; t < CAP; t++) Tick<CAP>::ph_c_store(P, env, t, sh, regs[t]);
    }
    delete shp;
}

struct Backend {
    static int set_device(int, std::string &) { return 0; }
    static int enter_device(int) { return 0; }
    static void leave_device(int) {}
    static void *dmalloc(size_t n) { return calloc(1, n); }
    static void dfree(void *p) { free(p); }
    static int memset0(void *p, size_t n, void *) { memset(p, 0, n); return 0; }
    static int n_xcc(int) { return 1; }
    static int d2h(void *dst, const void *src, size_t n, void *) { memcpy(dst, src, n); return 0; }
    static int sync(void *, std::string &) { return 0; }
    static int launch_tick(const Const &c, const Params &P, int cap, void *, std::string &)
    {
        if (cap == 64) emu_tick<64>(c, P); else emu_tick<128>(c, P);
        return 0;
    }
    // The persistent form (R.queue): the kernel's workgroups pull (intersection, chunk) items from a queue; here the items run
    // sequentially, chunk-major -- same item schedule (rollout_item), same first tick / pool row / output block per item, so the
    // schedule arithmetic of pve_capi.inc is under CPU parity.
    template <typename G, typename F>
    static int run_launch(const G &g, const Params &P_in, const RolloutArgs &R, int cap, F emu)
    {
        Params P = P_in;
        RolloutArgs Rk = R;
        if (!R.queue) {
            if (R.source == 1) {
                Rk.pool_tick0 = R.pool_tick0 % R.n_pool;
                P.actions = R.pool + (size_t)Rk.pool_tick0 * (size_t)P.n_envs * (size_t)cap;
            } else P.actions = nullptr;
            if (R.source == 3) Rk.pool_tick0 = R.pool_tick0 % R.n_pool;
            emu(g, P, Rk, 0);
            return 0;
        }
        for (int chunk = 0; chunk < R.n_full + R.n_taper; chunk++) {
            int kb, nt;
            rollout_item(R, chunk, kb, nt);
            if (nt < 1 || kb + nt > R.call_ticks) return -1;
            Rk.n_ticks = nt;
            Rk.pool_tick0 = (R.source == 1 || R.source == 3) ? (R.pool_tick0 + kb) % R.n_pool : 0;
            P.actions = R.source == 1 ? R.pool + (size_t)Rk.pool_tick0 * (size_t)P.n_envs * (size_t)cap : nullptr;
            emu(g, P, Rk, kb);
        }
        return 0;
    }
    static int launch_rollout(const Const &c, const Params &P_in, const RolloutArgs &R, int cap, void *, std::string &err)
    {
        if (getenv("PVE_NO_ROLLOUT_KERNEL") || R.source == 2) return 1;
        // PVE_EMU_HOME (tests): the HOME block of k_rollout<128, 5, ..> for every 128-slot roll-out without training outputs
        // (1 = the kernel's pool of 304 entries, 2 = a pool of 296 entries, the smallest the staging overlays admit)
        const char *hm = getenv("PVE_EMU_HOME");
        const int home = (hm && cap == 128 && !P_in.out.obs_pre && !P_in.out.state_pre) ? atoi(hm) : 0;
        const int rc = run_launch(c, P_in, R, cap, [&](const Const &cc, const Params &P, const RolloutArgs &Rk, int kb) {
            if (cap == 64) emu_rollout<64>(cc, P, Rk, kb);
            else if (home == 1) emu_rollout<128, Shared<128, false, true>>(cc, P, Rk, kb);
            else if (home >= 2) emu_rollout<128, Shared<128, false, true, 296>>(cc, P, Rk, kb);
            else emu_rollout<128>(cc, P, Rk, kb); });
        if (rc < 0) err = "emulated work queue: inconsistent item schedule";
        return rc;
    }
    static int launch_rollout_geo(const GeoConst &g, const Params &P_in, const RolloutArgs &R, int cap, void *, std::string &err)
    {
        const bool train = P_in.out.obs_pre || P_in.out.state_pre;
        if (getenv("PVE_NO_ROLLOUT_KERNEL") || R.source == 2 || (train && R.source == 3)) return 1;
        const int rc = run_launch(g, P_in, R, cap, [&](const GeoConst &gg, const Params &P, const RolloutArgs &Rk, int kb) {
            if (cap == 64) emu_rollout_geo<64>(gg, P, Rk, kb); else emu_rollout_geo<128>(gg, P, Rk, kb); });
        if (rc < 0) err = "emulated work queue: inconsistent item schedule";
        return rc;
    }
    static int launch_compact(const Params &P, int cap, void *, std::string &)
    {
        if (cap == 64) emu_compact<64>(P); else emu_compact<128>(P);
        return 0;
    }
    static int launch_tick_geo(const GeoConst &g, const Params &P, int cap, void *, std::string &)
    {
        if (cap == 64) emu_tick_geo<64>(g, P); else emu_tick_geo<128>(g, P);
        return 0;
    }
    static int launch_reset_geo(const GeoConst &g, const Params &P, int cap, void *, std::string &)
    {
        for (int env = 0; env < P.n_envs; env++) {
            if (cap == 64) reset_env_geo<64>(g, P, env, 200000); else reset_env_geo<128>(g, P, env, 200000);
        }
        return 0;
    }
    // the canonical float32 evaluation order of csrc/pve_actor.h (what the matrix-core kernel computes)
    static int pack_actor(const float *W, float *flat, unsigned char *, void *, std::string &)
    {
        memcpy(flat, W, sizeof(float) * AW_TOTAL);
        return 0;
    }
    static int launch_actor(const float *W, const unsigned char *, const void *obs_v, int obs_f32, const int32_t *meta,
                            double *actions, int n_envs, int cap, void *, std::string &)
    {
        const double *obs = (const double *)obs_v;
        const float *obsf = (const float *)obs_v;
        for (size_t s = 0; s < (size_t)n_envs * cap; s++) {
            if ((meta[s] & (M_ALIVE | M_CONTROL)) != (M_ALIVE | M_CONTROL)) { actions[s] = 0.0; continue; }
            float x[ACT_IN];
            for (int k = 0; k < ACT_IN; k++) x[k] = (obs_f32 & 1) ? obsf[s * OBSW + k] : (float)obs[s * OBSW + k];
            actions[s] = (double)actor_canonical(W, x);
        }
        return 0;
    }
    static int launch_probe(const Params &, int, int *, void *, std::string &) { return 0; }
    static int launch_reset(const Const &c, const Params &P, int cap, void *, std::string &)
    {
        for (int env = 0; env < P.n_envs; env++) {
            if (cap == 64) reset_env<64>(c, P, env, 200000); else reset_env<128>(c, P, env, 200000);
        }
        return 0;
    }
};

#include "../../pve-mcc_for_unsignalized_intersection_amd/csrc/pve_capi.inc"
