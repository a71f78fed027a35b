#!/usr/bin/env python3
"""Per-phase wave-cycle breakdown of k_tick at the bench workload (diagnostics, GPU only).
Usage: python tools/phase_profile.py [--capacity 128] [--envs 4096] [--ticks 200]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import pve_mcc_amd  # noqa: E402
from pve_mcc_amd.arrivals import synthetic_arrivals  # noqa: E402

PHASES = ("load", "step1", "step2+listsA", "step3+listsB", "build", "rank", "reward+xy", "effects", "lock",
          "final", "state", "walk(merge)")
# k_rollout (--many): column 0 = barrier behind RELOAD (first tick: LOAD), 9 = FIN, 10 = barrier A + STAGE + barrier B + RELOAD
PHASES_MANY = ("reload barrier", "init+step1", "step2+listsA", "step3+listsB", "build", "rank", "reward+xy", "effects",
               "lock", "final", "stage+reload", "walk(merge)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--capacity", type=int, default=128)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--outputs", default="obs_post,reward,flags,nbr,new_slot,env_out")
    ap.add_argument("--lane-num", type=int, default=12, choices=(12, 8, 4), help="4 / 8: k_tick_geo (column 4 = FILL)")
    ap.add_argument("--geo-scan", action="store_true", help="k_tick_geo with the membership scan (PVE_CFG_GEO_SCAN)")
    ap.add_argument("--many", action="store_true", help="profile k_rollout (pve_step_many: all ticks in one launch)")
    ap.add_argument("--rate", type=float, default=None, help="veh/h/lane (default: the bench rate of the layout)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rate = a.rate or {12: 1100.0 if a.capacity == 128 else 350.0, 8: 1500.0, 4: 1800.0}[a.lane_num]
    arr = synthetic_arrivals(a.envs, rate=rate, horizon_s=(a.ticks + a.warmup) * 0.1 + 20, lane_num=a.lane_num)
    from pve_mcc_amd.arrivals import synthetic_intentions
    ch = synthetic_intentions(a.envs, arr.shape[1], seed=1) if a.lane_num == 8 else None
    env = pve_mcc_amd.BatchedIntersections(a.envs, a.capacity, arr, device=dev, lane_num=a.lane_num, intentions=ch,
                                           geo_scan=a.geo_scan, outputs=tuple(x for x in a.outputs.split(",") if x))
    pool = torch.as_tensor(bench.action_pool(a.envs, a.capacity, 99), device=dev)
    env.reset()
    for t in range(a.warmup):
        env.step(pool[t % bench.N_POOL])
    buf = torch.zeros(a.envs * (a.capacity // 64), 16, dtype=torch.int64, device=dev)
    env.lib.pve_debug_phase_cycles(env._h, C.c_void_p(buf.data_ptr()))
    if a.many:
        env.set_action_pool(pool)
        env.step_many(a.ticks)
    else:
        for t in range(a.warmup, a.warmup + a.ticks):
            env.step(pool[t % bench.N_POOL])
    torch.cuda.synchronize()
    env.lib.pve_debug_phase_cycles(env._h, None)
    cyc = buf.sum(0).cpu().numpy().astype(float)
    if cyc[13] > 0:
        print("shader clock during the launch: %.0f MHz (s_memtime / 100 MHz wall clock)" % (100.0 * cyc[12] / cyc[13]))
    cyc[12:] = 0
    waves = a.envs * (a.capacity // 64) * a.ticks
    tot = cyc.sum()
    print("phase                cycles/wave   share")
    names = PHASES_MANY if a.many else PHASES
    for k, name in enumerate(names):
        if cyc[k] > 0:
            print("%-18s %12.0f  %6.1f%%" % (name, cyc[k] / waves, 100 * cyc[k] / tot))
    print("%-18s %12.0f  (wall_clock64 ticks, 100 MHz constant clock => x10 ns)" % ("total", tot / waves))
    import numpy as np
    per_wave = buf[:, :12].sum(1).cpu().numpy().astype(float) / a.ticks
    per_env = per_wave.reshape(a.envs, -1).max(1)
    n_alive = env.state_field("meta").ne(0).sum(1).cpu().numpy()
    q = np.percentile(per_env, [0, 10, 50, 90, 99, 100])
    print("per-env wave time (avg over ticks), ticks of 10ns: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(q))
    ph = buf.cpu().numpy().astype(float).reshape(a.envs, -1, 16).max(1) / a.ticks
    slow = per_env >= np.percentile(per_env, 99)
    mid = (per_env >= np.percentile(per_env, 40)) & (per_env <= np.percentile(per_env, 60))
    print("phase means  (slowest 1%% | middle 20%%):")
    for k, name in enumerate(names):
        if ph[:, k].sum() > 0:
            print("   %-16s %7.0f | %7.0f" % (name, ph[slow, k].mean(), ph[mid, k].mean()))
    print("corr(time, n_alive) = %.3f ; n_alive min/mean/max = %d / %.1f / %d" % (
        np.corrcoef(per_env, n_alive)[0, 1], n_alive.min(), n_alive.mean(), n_alive.max()))


if __name__ == "__main__":
    main()
