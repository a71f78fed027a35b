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
    c.aM_minus_am = cfg.aM - cfg.am;
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
};
