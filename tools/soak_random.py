"""Randomised-configuration parity soak on the GPU (not part of the test-suite): layout, capacity, arrival rate, action
scale and quantisation are drawn per run; every env against its own sequential oracle, every tick.
python tools/soak_random.py [--runs 40] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import scenarios  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--many", type=int, default=0, help="additional pve_step_many == single-tick runs")
ap.add_argument("--backend", default="hip")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
# (layout, capacity) -> arrival rates that keep the intersections below their capacity for ~1000 ticks with most tapes
RATES = {(12, 128): (300.0, 1250.0), (12, 64): (150.0, 480.0), (8, 128): (400.0, 1700.0), (8, 64): (200.0, 750.0),
         (4, 128): (600.0, 3000.0), (4, 64): (300.0, 2000.0)}
ok = stopped = 0
for k in range(a.runs):
    ln = int(rng.choice([12, 12, 12, 8, 8, 4]))
    cap = int(rng.choice([64, 128, 128]))
    lo, hi = RATES[(ln, cap)]
    rate = float(rng.uniform(lo, hi))
    scale = float(rng.choice([0.2, 0.5, 1.0, 2.0, 3.0]))
    quant = rng.choice([0.0, 0.0, 0.1, 0.25, 0.5, 1.0, 3.0])
    quant = None if quant == 0.0 or quant > scale * 2 else float(quant)
    ticks = int(rng.integers(500, 1300))
    n_envs = int(rng.choice([8, 16, 24]))
    seed = int(rng.integers(1, 1 << 30))
    what = "run %2d: %2d lanes cap %3d rate %6.0f |a|<=%.1f quant %-5s %4d ticks x %2d envs seed %d" % (
        k, ln, cap, rate, scale, quant, ticks, n_envs, seed)
    t0 = time.time()
    try:
        if ln == 12:
            c, l = scenarios.check_fuzz_vs_oracle(a.backend, n_envs=n_envs, capacity=cap, ticks=ticks, rate=rate, seed=seed,
                                                  action_scale=scale, quantize=quant)
        else:
            c, l = scenarios.check_geo_fuzz_vs_oracle(a.backend, ln, n_envs=n_envs, capacity=cap, ticks=ticks, rate=rate,
                                                      seed=seed, action_scale=scale, quantize=quant)
        ok += 1
        print("%s OK (collisions %d, locks %d) %.0f s" % (what, c, l, time.time() - t0), flush=True)
    except scenarios.CapacityOverflow as ex:
        stopped += 1
        print("%s stopped by a full intersection (%s) %.0f s" % (what, str(ex)[:60], time.time() - t0), flush=True)
print("%d runs: %d green, %d ended at a deferred spawn, 0 differences" % (a.runs, ok, stopped))
# pve_step_many (k_rollout: still ticks, staged ticks, chunked launches, trajectory outputs) == single ticks, bit for bit
for k in range(a.many):
    kind = str(rng.choice(["k_rollout", "k_rollout", "actor", "geo", "geo", "state", "persistent", "persistent", "geo_state", "geo_table",
                           "state_pers", "state_pers", "closed_state", "closed_geo", "closed_geo", "home", "home", "home",
                           "closed_geo_state", "closed_geo_state", "geo_state"]))
    seed = int(rng.integers(1, 1 << 30))
    t0 = time.time()
    if kind in ("k_rollout", "actor"):
        cap = int(rng.choice([64, 128, 128]))
        lo, hi = RATES[(12, cap)]
        rate = float(rng.uniform(lo * 0.3, hi * 1.15))          # (deferred spawns are deterministic: both paths agree on them too)
        n_envs = int(rng.choice([3, 8, 17]))
        chunks = tuple(int(x) for x in rng.integers(1, 90, size=int(rng.integers(2, 6))))
        src = "actor" if kind == "actor" else str(rng.choice(["pool", "pool", "zero"]))
        if src == "actor":
            rate = min(rate, hi * 0.85)                         # (the pretrained policy keeps more vehicles in the box)
        scenarios.check_step_many(a.backend, src, n_envs=n_envs, capacity=cap, rate=rate, prefill=int(rng.choice([0, 150, 320])),
                                  chunks=chunks, trajectory_chunk=int(rng.integers(2, 30)), seed=seed)
        what = "cap %3d rate %6.0f %s chunks %s x %2d envs" % (cap, rate, src, chunks, n_envs)
    elif kind == "home":                                        # round 6: the HOME build of the 128-slot queue kernel under random constructor
        # arguments and loads up to a FULL intersection (more list entries than its pool holds: several BUILD .. WALK passes; deferred
        # spawns: the table source's late gather)
        src = str(rng.choice(["table", "table", "pool", "zero"]))
        cfg = dict(vm=float(rng.choice([0.5, 2.0, 3.0, 5.0, 6.0])), collision_thr=float(rng.choice([0.01, 2.0, 3.0])),
                   deltaT=float(rng.choice([0.1, 0.1, 0.2])), dis_ctl=float(rng.choice([150.0, 120.0])))
        rate = float(rng.uniform(600.0, 3200.0))
        n_envs = int(rng.choice([9, 40, 130, 300]))
        chunks = tuple(int(x) for x in rng.integers(8, 120, size=int(rng.integers(2, 5))))
        lo_a = float(rng.choice([-3.0, -3.0, -1.0]))
        st = scenarios.check_step_many(a.backend, src, n_envs=n_envs, capacity=128, rate=rate, prefill=int(rng.choice([0, 150, 250])),
                                       chunks=chunks, trajectory_chunk=int(rng.integers(2, 30)), seed=seed, persistent=True, cfg=cfg,
                                       act_lo=lo_a, act_hi=float(rng.choice([-2.0, 0.0, 3.0])) if lo_a < -2 else 3.0)
        what = "HOME %s rate %6.0f cfg %s chunks %s x %3d envs: max alive %d overflow %d" % (src, rate, cfg, chunks, n_envs, st["max_alive"], st["overflow"])
    elif kind == "persistent":                                  # the work-queue launch (round 4): intersections change hands inside the launch
        cap = int(rng.choice([64, 128, 128]))
        lo, hi = RATES[(12, cap)]
        src = str(rng.choice(["pool", "zero", "table", "actor"]))
        rate = float(rng.uniform(lo * 0.5, hi * (0.85 if src == "actor" else 1.1)))
        n_envs = int(rng.choice([9, 40, 130, 300, 700]))
        chunks = tuple(int(x) for x in rng.integers(1, 70, size=int(rng.integers(2, 5))))
        scenarios.check_step_many(a.backend, src, n_envs=n_envs, capacity=cap, rate=rate, prefill=int(rng.choice([0, 150])),
                                  chunks=chunks, trajectory_chunk=int(rng.integers(2, 30)), seed=seed, persistent=True)
        what = "cap %3d rate %6.0f %s chunks %s x %3d envs (persistent)" % (cap, rate, src, chunks, n_envs)
    elif kind == "state_pers":                                  # round 5: the trainer's roll-out through the work queue, pool / zero / table sources
        import torch
        dt = torch.float32 if rng.random() < 0.5 else torch.float64
        cap = int(rng.choice([64, 128, 128]))
        lo, hi = RATES[(12, cap)]
        rate = float(rng.uniform(lo * 1.2, hi * 0.95))
        calls = tuple(int(x) for x in rng.integers(5, 60, size=int(rng.integers(2, 5))))
        src = str(rng.choice(["pool", "zero", "table"]))
        n_envs = int(rng.choice([3, 20, 70]))
        scenarios.check_step_many_state_rows(a.backend, n_envs=n_envs, capacity=cap, calls=calls, rate=rate, seed=seed, obs_dtype=dt,
                                             chunk=int(rng.choice([3, 7, 16])), source=src, min_ctl_per_tick=0, persistent=True)
        what = "cap %3d state rows %s calls %s rate %6.0f %s x %2d envs (persistent)" % (cap, str(dt).split(".")[-1], calls, rate, src, n_envs)
    elif kind == "closed_state":                                # round 5: closed loop + training outputs (k_rollout<ACT, TRAIN[, PERS]>)
        import torch
        cap = int(rng.choice([64, 128]))
        lo, hi = RATES[(12, cap)]
        rate = float(rng.uniform(lo * 1.2, hi * 0.8))
        calls = tuple(int(x) for x in rng.integers(5, 50, size=int(rng.integers(2, 4))))
        pers = bool(rng.random() < 0.6)
        scenarios.check_closed_loop_state_rows(a.backend, n_envs=int(rng.choice([4, 30])), capacity=cap, rate=rate, calls=calls,
                                               chunk=int(rng.choice([3, 7, 16])), seed=seed, persistent=pers,
                                               obs_dtype=torch.float32 if rng.random() < 0.5 else torch.float64)
        what = "cap %3d closed-loop state rows calls %s rate %6.0f persistent %s" % (cap, calls, rate, pers)
    elif kind == "closed_geo":                                  # round 5: closed loop for 4 / 8 lanes inside k_rollout_geo<.., ACT[, PERS]>
        import torch
        ln = int(rng.choice([4, 8]))
        cap = int(rng.choice([64, 128]))
        lo, hi = RATES[(ln, cap)]
        rate = float(rng.uniform(lo, hi * 0.7))
        pers = bool(rng.random() < 0.6)
        chunks = tuple(int(x) for x in rng.integers(1, 60, size=int(rng.integers(2, 5))))
        scenarios.check_step_many_geo_actor(a.backend, ln, n_envs=int(rng.choice([3, 9, 40, 300])) if pers else int(rng.choice([3, 8, 12])),
                                            capacity=cap, chunks=chunks, rate=rate, trajectory_chunk=int(rng.integers(2, 20)), seed=seed,
                                            obs_dtype=torch.float32 if rng.random() < 0.5 else torch.float64, persistent=pers, strict=False)
        what = "%d lanes cap %3d rate %6.0f closed loop chunks %s persistent %s" % (ln, cap, rate, chunks, pers)
    elif kind == "closed_geo_state":                            # round 6: closed loop + training outputs for 4 / 8 lanes (k_rollout_geo<.., TRAIN, .., ACT>)
        import torch
        ln = int(rng.choice([4, 8]))
        cap = int(rng.choice([64, 128]))
        lo, hi = RATES[(ln, cap)]
        rate = float(rng.uniform(lo, hi * 0.7))
        calls = tuple(int(x) for x in rng.integers(5, 50, size=int(rng.integers(2, 4))))
        pers = bool(rng.random() < 0.4)
        scenarios.check_closed_loop_state_rows(a.backend, n_envs=int(rng.choice([4, 30])), capacity=cap, rate=rate, calls=calls,
                                               chunk=int(rng.choice([0, 3, 7, 16])), seed=seed, persistent=pers, lane_num=ln,
                                               want_launch=("resident",), obs_dtype=torch.float32 if rng.random() < 0.5 else torch.float64)
        what = "%d lanes cap %3d closed-loop state rows calls %s rate %6.0f persistent %s" % (ln, cap, calls, rate, pers)
    elif kind in ("geo_state", "geo_table"):                    # f3 x f4 (round 4): training outputs / id-indexed table for 4 / 8 lanes
        import torch
        ln = int(rng.choice([4, 8]))
        cap = int(rng.choice([64, 128]))
        lo, hi = RATES[(ln, cap)]
        rate = float(rng.uniform(lo, hi * 0.8))
        if kind == "geo_state":
            dt = torch.float32 if rng.random() < 0.5 else torch.float64
            calls = tuple(int(x) for x in rng.integers(5, 60, size=int(rng.integers(2, 5))))
            pers = bool(rng.random() < 0.5)                     # (8 lanes through the work queue: round 5; 4 lanes: round 6)
            scenarios.check_step_many_state_rows(a.backend, n_envs=int(rng.choice([2, 5, 30])) if pers else int(rng.choice([2, 5])),
                                                 capacity=cap, calls=calls, rate=rate, seed=seed, obs_dtype=dt,
                                                 chunk=int(rng.choice([3, 7, 16])) if pers else int(rng.choice([0, 7, 16])), lane_num=ln,
                                                 min_ctl_per_tick=0, persistent=pers)
            what = "%d lanes cap %3d state rows %s calls %s rate %6.0f persistent %s" % (ln, cap, str(dt).split(".")[-1], calls, rate, pers)
        else:
            chunks = tuple(int(x) for x in rng.integers(1, 70, size=int(rng.integers(2, 5))))
            pers = bool(rng.random() < 0.5)                     # (round 5: the table source through the work queue)
            scenarios.check_step_many_geo(a.backend, ln, n_envs=int(rng.choice([3, 12, 60, 300])) if pers else int(rng.choice([3, 8, 12])),
                                          capacity=cap, chunks=chunks, rate=rate, trajectory_chunk=int(rng.integers(2, 20)), seed=seed,
                                          source="table", persistent=pers)
            what = "%d lanes cap %3d rate %6.0f table chunks %s persistent %s" % (ln, cap, rate, chunks, pers)
    elif kind == "geo":                                         # k_rollout_geo == k_tick_geo ticks
        ln = int(rng.choice([4, 8]))
        cap = int(rng.choice([64, 128]))
        lo, hi = RATES[(ln, cap)]
        rate = float(rng.uniform(lo, hi * 0.9))
        chunks = tuple(int(x) for x in rng.integers(1, 70, size=int(rng.integers(2, 5))))
        quant = rng.choice([0.0, 0.5, 1.0])
        scenarios.check_step_many_geo(a.backend, ln, n_envs=int(rng.choice([3, 8, 12])), capacity=cap, chunks=chunks, rate=rate,
                                      trajectory_chunk=int(rng.integers(2, 20)), seed=seed, quantize=None if quant == 0.0 else float(quant))
        what = "%d lanes cap %3d rate %6.0f chunks %s quant %s" % (ln, cap, rate, chunks, quant)
    else:                                                       # training outputs of pve_step_many vs the oracle, every tick
        import torch
        dt = torch.float32 if rng.random() < 0.5 else torch.float64
        calls = tuple(int(x) for x in rng.integers(5, 60, size=int(rng.integers(2, 5))))
        rate = float(rng.uniform(400.0, 1200.0))
        scenarios.check_step_many_state_rows(a.backend, n_envs=int(rng.choice([2, 5])), calls=calls, rate=rate, seed=seed,
                                             obs_dtype=dt, chunk=int(rng.choice([0, 7, 16])), source=str(rng.choice(["pool", "zero"])),
                                             min_ctl_per_tick=0)
        what = "state rows %s calls %s rate %6.0f" % (str(dt).split(".")[-1], calls, rate)
    print("many %2d [%s]: %s seed %d OK %.0f s" % (k, kind, what, seed, time.time() - t0), flush=True)
if a.many:
    print("%d pve_step_many runs (12-lane pool / zero / actor, 4- / 8-lane, training outputs): all bit-identical to single ticks / equal to the oracle" % a.many)
