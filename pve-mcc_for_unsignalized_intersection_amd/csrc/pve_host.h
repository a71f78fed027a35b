// pve_host.h -- backend-independent host side of the C ABI (handle, constants, layout).
// Included by pve_hip.hip (product, HIP backend) and by tests/emu/pve_emu.cpp (CPU test emulator).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "../../include/pve_env.h"
#include "pve_types.h"

namespace pve {

// Geometry and limits exactly as the reference constructor derives them
// (ref traffic_interaction_scene.py:21-45, :148-152, :153-166, :182-186; libm = CPython's math).
inline Const make_const(const pve_config &cfg)
{
    Const c;
    memset(&c, 0, sizeof(c));
    const double cw = cfg.lane_cw, dc = cfg.dis_ctl;
    c.deltaT = cfg.deltaT; c.dt2 = pow(cfg.deltaT, 2);
    c.vm = cfg.vm; c.vM = cfg.vM; c.am = cfg.am; c.aM = cfg.aM; c.v0 = cfg.v0;
    c.abs_am = fabs(cfg.am); c.two_abs_am = 2 * fabs(cfg.am);
    c.inv_abs_am = 1.0 / c.abs_am; c.inv_two_abs_am = 1.0 / c.two_abs_am;
    c.aM_minus_am = cfg.aM - cfg.am;
    c.inv_dt = 1.0 / cfg.deltaT; c.inv_span = 1.0 / (cfg.aM - cfg.am);
    c.collision_thr = cfg.collision_thr; c.lock_mean_thr = cfg.collision_thr + 3;
    c.exit_p = -dc + (double)((NL + 1) / 2) * cw;
    c.cw = cw;
    const double approach = dc - 6 * cw;
    c.inbox[0] = 3.1415 / 2 * 7 * cw; c.inbox[1] = 12 * cw; c.inbox[2] = 3.1415 / 2 * cw;
    for (int m = 0; m < 3; m++) c.spawn_p[m] = 0 + approach + c.inbox[m];
    const double pi = 3.141592653589793;
    const double cita = (2 * sqrt(10.0) - 6) * cw;
    const double alpha = atan((6 * cw + cita) / (3 * cw));
    const double beta = pi / 2 - alpha;
    const double gama = atan((sqrt(13.0) * cw) / (6 * cw));
    const double _gama = pi / 2 - gama;
    // ego = left turn (ref :771-799): delta = p1 - A + B, vd = C + delta
    c.vdA[0][0] = 6 * cw;          c.vdB[0][0] = cita;  c.vdC[0][0] = alpha * 7 * cw;
    c.vdA[0][1] = gama * 7 * cw;   c.vdB[0][1] = 0;     c.vdC[0][1] = _gama * 7 * cw;
    c.vdA[0][2] = _gama * 7 * cw;  c.vdB[0][2] = 0;     c.vdC[0][2] = gama * 7 * cw;
    c.vdA[0][3] = 6 * cw;          c.vdB[0][3] = -cita; c.vdC[0][3] = beta * 7 * cw;
    // ego = straight (ref :733-766)
    c.vdA[1][0] = 3 * cw;          c.vdB[1][0] = 0;     c.vdC[1][0] = 9 * cw;
    c.vdA[1][1] = beta * 7 * cw;   c.vdB[1][1] = 0;     c.vdC[1][1] = 6 * cw + cita;
    c.vdA[1][2] = alpha * 7 * cw;  c.vdB[1][2] = 0;     c.vdC[1][2] = 6 * cw - cita;
    c.vdA[1][3] = 9 * cw;          c.vdB[1][3] = 0;     c.vdC[1][3] = 3 * cw;
    for (int k = 0; k < 4; k++) {
        double ang = 3.141593 / 2 * (double)k;            // ref :1251
        c.rot_cos[k] = cos(ang); c.rot_sin[k] = sin(ang);
    }
    c.arc_k = 3.141593;
    static const int8_t l2l[NL][4] = {
        {10, 3, 9, 7}, {10, 6, 3, 4}, {-1, -1, -1, -1},
        {1, 6, 0, 10}, {1, 9, 6, 7},  {-1, -1, -1, -1},
        {4, 9, 3, 1},  {4, 0, 9, 10}, {-1, -1, -1, -1},
        {7, 0, 6, 4},  {7, 3, 0, 1},  {-1, -1, -1, -1}};
    for (int i = 0; i < NL; i++)
        for (int k = 0; k < 4; k++) c.l2l[i][k] = l2l[i][k];
    for (int i = 0; i < NL; i++)
        for (int k = 0; k < 4; k++) {
            c.l2l_inv[i][k] = -1;
            int o = l2l[i][k];
            if (o < 0) continue;
            for (int q = 0; q < 4; q++) if (l2l[o][q] == i) c.l2l_inv[i][k] = (int8_t)q;
        }
    return c;
}

// General geometry: constructor tables of the 4- / 8-lane branches (ref :66-145), get_virtual_distance as a table
// (ref :453-660), get_p as canonical path + quarter turns (ref :896-1249).  lane_num = 12 re-expresses make_const's
// tables in the same form (used to cross-check the general kernel against the fast path).
inline GeoConst make_geo_const(const pve_config &cfg)
{
    GeoConst g;
    memset(&g, 0, sizeof(g));
    g.base = make_const(cfg);
    Const &c = g.base;
    const int LN = cfg.lane_num;
    const double cw = cfg.lane_cw, dc = cfg.dis_ctl;
    g.lane_num = LN;
    memset(g.l2l, -1, sizeof(g.l2l));
    memset(g.direction, -1, sizeof(g.direction));
    memset(g.tab.dir_lane, 0, sizeof(g.tab.dir_lane));
    auto set = [&](int ty, int k, double A, double B, double C, double C2) {
        g.tab.vd[ty][k][0] = A; g.tab.vd[ty][k][1] = B; g.tab.vd[ty][k][2] = C; g.tab.vd[ty][k][3] = C2;
    };
    if (LN == 4) {
        g.dir_num = 12; g.tmod = 3; g.RL = 3; g.H = 2 * cw;
        const double approach = dc - 2 * cw;                                     // ref :69-71
        c.inbox[0] = 3.1415 / 2 * 3 * cw; c.inbox[1] = 4 * cw; c.inbox[2] = 3.1415 / 2 * cw;
        for (int m = 0; m < 3; m++) c.spawn_p[m] = 0 + approach + c.inbox[m];
        static const int8_t l2l[12][7] = {                                       // ref :74-87
            {10, 6, 9, 3, 7, 4, 8}, {10, 6, 3, 4, 9, 5, -1}, {6, 10, -1, -1, -1, -1, -1},
            {1, 9, 0, 6, 10, 7, 11}, {1, 9, 6, 7, 0, 8, -1}, {9, 1, -1, -1, -1, -1, -1},
            {4, 0, 3, 9, 1, 10, 2},  {4, 0, 9, 10, 3, 11, -1}, {0, 4, -1, -1, -1, -1, -1},
            {7, 3, 6, 0, 4, 1, 5},   {7, 3, 0, 1, 6, 2, -1},  {3, 7, -1, -1, -1, -1, -1}};
        for (int d = 0; d < 12; d++) for (int k = 0; k < 7; k++) g.l2l[d][k] = l2l[d][k];
        static const int8_t dir[4][3] = {{6, 7, 8}, {0, 1, 2}, {9, 10, 11}, {3, 4, 5}};   // ref :88-93
        for (int i = 0; i < 4; i++) for (int k = 0; k < 3; k++) g.direction[i][k] = dir[i][k];
        static const int8_t turn[4] = {0, 2, 1, 3};                              // ref :897, :939, :981, :1022
        for (int i = 0; i < 4; i++) g.turn[i] = turn[i];
        const double alpha = atan((4 - sqrt(2.0)) / (4 + sqrt(2.0)));            // ref :94-98
        const double _alpha = atan((4 + sqrt(2.0)) / (4 - sqrt(2.0)));
        const double beta = atan(2 / sqrt(5.0)), _beta = atan(sqrt(5.0) / 2);
        const double gama = atan(1.0 / 2 * sqrt(2.0));
        const double half = 0.5 * 3.1415, q = 1.5 * 3.1415 * cw, cg = cos(gama);
        auto F = [&](double x) { return (q * x) / half; };                       // 1.5*3.1415*cw*x/(0.5*3.1415)
        auto G = [&](double x) { return q * (x / half); };                       // (1.5*3.1415)*cw*(x/(0.5*3.1415))
        set(0, 0, 4 * cw - 3 * cw * cg, 0, 3 * cw * (0.5 * 3.1415 - gama), 0);   // ref :455-459
        set(0, 1, G(_alpha), 0, G(alpha), 0);                                    // ref :476-480
        set(0, 2, F(beta), 0, F(_beta), 0);                                      // ref :466-470
        set(0, 3, F(_beta), 0, F(beta), 0);                                      // ref :471-475
        set(0, 4, 3 * cw * cg, 0, G(gama), 0);                                   // ref :481-485
        set(1, 0, cw, 0, 3 * cw, 0);                                             // ref :497-516
        set(1, 1, F(gama), 0, 3 * cw * cg, 0);
        set(1, 2, F(0.5 * 3.1415 - gama), 0, 4 * cw - 3 * cw * cg, 0);
        set(1, 3, 3 * cw, 0, cw, 0);
        g.fix_d = (_alpha - alpha) * 3 * cw;                                     // ref :1304
        g.fix_hi = _alpha * 3 * cw; g.fix_lo = alpha * 3 * cw;                   // ref :1309, :1316
    } else if (LN == 8) {
        g.dir_num = 16; g.tmod = 4; g.RL = 5; g.H = 4 * cw;
        const double approach = dc - 4 * cw;                                     // ref :103-105
        c.inbox[0] = 3.1415 / 2 * 5 * cw; c.inbox[1] = 8 * cw; c.inbox[2] = 3.1415 / 2 * cw;
        for (int m = 0; m < 3; m++) c.spawn_p[m] = 0 + approach + c.inbox[m];
        static const int8_t l2l[16][7] = {                                       // ref :107-124
            {14, 4, 13, 12, 9, 10, 5}, {14, 13, 8, 4, 5, 6, 12}, {14, 13, 8, 4, 5, 6, 7}, {14, -1, -1, -1, -1, -1, -1},
            {2, 8, 1, 0, 13, 14, 9},   {2, 1, 12, 8, 9, 10, 0},  {2, 1, 12, 8, 9, 10, 11}, {2, -1, -1, -1, -1, -1, -1},
            {6, 12, 5, 4, 1, 2, 13},   {6, 5, 0, 12, 13, 14, 4}, {6, 5, 0, 12, 13, 14, 15}, {6, -1, -1, -1, -1, -1, -1},
            {10, 0, 9, 8, 5, 6, 1},    {10, 9, 4, 0, 1, 2, 8},   {10, 9, 4, 0, 1, 2, 3},   {10, -1, -1, -1, -1, -1, -1}};
        for (int d = 0; d < 16; d++) for (int k = 0; k < 7; k++) g.l2l[d][k] = l2l[d][k];
        static const int8_t dir[8][3] = {{0, 1, -1}, {-1, 2, 3}, {4, 5, -1}, {-1, 6, 7},
                                         {8, 9, -1}, {-1, 10, 11}, {12, 13, -1}, {-1, 14, 15}};   // ref :135-144
        for (int i = 0; i < 8; i++) for (int k = 0; k < 3; k++) g.direction[i][k] = dir[i][k];
        static const int8_t turn[4] = {2, 3, 0, 1};                              // lane pairs, ref :1062-1249
        for (int i = 0; i < 8; i++) g.turn[i] = turn[i / 2];
        const double s24 = sqrt(24.0);
        set(0, 0, 8 * cw - s24 * cw, 0, atan(s24) * 5 * cw, 0);                  // ref :540-575
        set(0, 1, atan(3.0 / 4) * 5 * cw, 0, atan(4.0 / 3) * 5 * cw, 0);
        set(0, 2, 4 * cw, 0, atan(4.0 / 3) * 5 * cw, 0);
        set(0, 3, atan(4.0 / 3) * 5 * cw, 0, atan(3.0 / 4) * 5 * cw, 0);
        set(0, 4, 4 * cw, 0, atan(3.0 / 4) * 5 * cw, 0);
        set(0, 5, s24 * cw, 0, atan(1 / s24) * 5 * cw, 0);
        set(1, 0, 3 * cw, 0, 7 * cw, 0);                                         // ref :578-613
        set(1, 1, 3 * cw, 0, 5 * cw, 0);
        set(1, 2, atan(3.0 / 4) * 5 * cw, 0, 4 * cw, 0);
        set(1, 3, atan(4.0 / 3) * 5 * cw, 0, 4 * cw, 0);
        set(1, 4, 5 * cw, 0, 3 * cw, 0);
        set(1, 5, 5 * cw, 0, cw, 0);
        set(2, 0, cw, 0, 7 * cw, 0);                                             // ref :615-652
        set(2, 1, cw, 0, 5 * cw, 0);
        set(2, 2, atan(1 / s24) * 5 * cw, 0, s24 * cw, 0);
        set(2, 3, atan(s24) * 5 * cw, 0, 8 * cw, s24 * cw);                      // abs(d) + 8cw - sqrt(24)cw, ref :634
        set(2, 4, 7 * cw, 0, 3 * cw, 0);
        set(2, 5, 7 * cw, 0, cw, 0);
    } else {
        g.dir_num = 12; g.tmod = 3; g.RL = 7; g.H = 6 * cw;
        for (int d = 0; d < 12; d++) for (int k = 0; k < 4; k++) g.l2l[d][k] = c.l2l[d][k];
        for (int i = 0; i < 12; i++) g.direction[i][i % 3] = (int8_t)i;          // ref :168-181
        for (int m = 0; m < 2; m++) for (int k = 0; k < 4; k++) set(m, k, c.vdA[m][k], c.vdB[m][k], c.vdC[m][k], 0);
    }
    c.exit_p = -dc + (double)((LN + 1) / 2) * cw;                                // ref :341-342
    for (int i = 0; i < LN; i++)
        for (int m = 0; m < 3; m++)
            if (g.direction[i][m] >= 0) { g.tab.dir_lane[g.direction[i][m]] = (int8_t)i; g.tab.dir_index[g.direction[i][m]] = (int8_t)m; }
    // membership tables of the virtual-lane lists (ref :240-270): which routes can appear in list d, and where
    memset(g.tab.pos, -1, sizeof(g.tab.pos));
    memset(g.tab.mroutes, 0, sizeof(g.tab.mroutes));
    memset(g.tab.lroutes, 0, sizeof(g.tab.lroutes));
    for (int d = 0; d < ND; d++) {
        g.tab.opp[d] = g.l2l[d][1];
        if (d >= g.dir_num) continue;
        for (int q = 0; q < MAXK; q++) if (g.l2l[d][q] >= 0) g.tab.pos[d][g.l2l[d][q]] = (int8_t)q;
        unsigned mr = 0;
        const int li = g.tab.dir_lane[d];
        for (int q = 0; q < 3; q++) if (g.direction[li][q] >= 0) mr |= 1u << g.direction[li][q];
        for (int q = 0; q < MAXK; q++) if (g.l2l[d][q] >= 0) mr |= 1u << g.l2l[d][q];
        g.tab.mroutes[d] = (uint16_t)mr;
    }
    for (int d = 0; d < g.dir_num; d++)
        for (int rt = 0; rt < ND; rt++) if ((g.tab.mroutes[d] >> rt) & 1u) g.tab.lroutes[rt] |= (uint16_t)(1u << d);
    memset(g.tab.lst, 0, sizeof(g.tab.lst));
    for (int rt = 0; rt < ND; rt++) {
        int n = 0;
        for (int d = 0; d < g.dir_num; d++) if ((g.tab.lroutes[rt] >> d) & 1u) g.tab.lst[rt][n++] = (int8_t)d;
        g.tab.nl[rt] = (int8_t)n;
        g.tab.ninv[rt] = (uint16_t)(n ? (32768 + n - 1) / n : 0);
    }
    for (int d = 0; d < ND; d++) g.tab.dty[d] = (int8_t)(d % g.tmod);
    for (int m = 0; m < 3; m++) g.tab.inbox[m] = c.inbox[m];
    g.turn_pk = 0; g.dir_pk[0] = g.dir_pk[1] = g.dir_pk[2] = 0;
    for (int i = 0; i < NL; i++) {
        g.turn_pk |= (unsigned long long)(g.turn[i] & 15) << (4 * i);
        for (int m = 0; m < 3; m++) g.dir_pk[m] |= (unsigned long long)((g.direction[i][m] + 1) & 31) << (5 * i);
    }
    return g;
}

inline std::string &last_error_ref()
{
    static thread_local std::string e;
    return e;
}
inline int fail(int code, const std::string &msg)
{
    last_error_ref() = msg;
    return code;
}

}  // namespace pve

struct pve_handle_s {
    pve::Const c;
    pve::GeoConst g;                  // general-geometry path (lane_num 4 / 8, or 12 with PVE_CFG_GENERAL_PATH)
    bool geo;
    int lane_num;
    const int32_t *choice;
    long long choice_stride;
    int choice_rows;
    pve_config cfg;
    int n_envs, cap, device;
    pve::Layout L;
    char *ws;
    bool own_ws;
    const double *arrivals;
    int rows;
    long long arr_stride;
    void *stream;
    bool has_arrivals, is_reset;
    long long ticks_since_reset;
    unsigned long long *phase_cycles;
    bool has_actor;                   // pve_set_actor installed an actor in the workspace
    int stop_phase;                   // pve_debug_stop_phase (diagnostics), -1 = off
    unsigned q_done_base;             // persistent roll-out: items completed per intersection since pve_reset (cumulative)
    int last_launch_kind;             // PVE_LAUNCH_*: what the last stepping call launched (pve_debug_last_launch)
};
