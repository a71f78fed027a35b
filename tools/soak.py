"""Long randomised parity soak on the GPU (not part of the test-suite): every env against its own sequential oracle,
every tick.  python tools/soak.py [--ticks 3000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import scenarios  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ticks", type=int, default=3000)
ap.add_argument("--seed", type=int, default=0, help="offset added to every tape / arrival seed")
a = ap.parse_args()


def guarded(what, fn):
    """A tape that fills every slot of an intersection ends the comparison there (deferred spawn: documented deviation)."""
    t0 = time.time()
    try:
        r = fn()
    except scenarios.CapacityOverflow as ex:
        print("%s: stopped by a full intersection (%s), %.0f s" % (what, ex, time.time() - t0))
        return None
    print("%s OK %s, %.0f s" % (what, "" if r is None else "(collisions %d, locks %d)" % tuple(r), time.time() - t0))
    return r


guarded("12 lanes, quantised 0.5: %d ticks x 32 envs" % a.ticks,
        lambda: scenarios.check_fuzz_vs_oracle("hip", n_envs=32, capacity=128, ticks=a.ticks, rate=1100.0, seed=101 + a.seed, quantize=0.5))
for seed, rate, cap, scale, quant in ((201, 1100.0, 128, 3.0, None), (202, 1350.0, 128, 0.3, None), (203, 450.0, 64, 1.0, None),
                                      (204, 1300.0, 128, 1.0, 0.25)):
    # unquantised tapes (the 32-bit key path of WALK), dense traffic (more than 64 controlled vehicles: the second wave
    # joins the dense-mapped phases), one-wave workgroups
    guarded("12 lanes, %.0f veh/h/lane, cap %d, |a| <= %.1f: %d ticks x 32 envs" % (rate, cap, scale, a.ticks),
            lambda: scenarios.check_fuzz_vs_oracle("hip", n_envs=32, capacity=cap, ticks=a.ticks, rate=rate, seed=seed + a.seed,
                                                   action_scale=scale, quantize=quant))
    print("   most controlled vehicles in one intersection: %d" % scenarios.check_fuzz_vs_oracle.max_ctl)
for src, rate, cap in (("pool", 1100.0, 128), ("pool", 300.0, 128), ("pool", 1400.0, 128), ("pool", 350.0, 64), ("zero", 900.0, 128)):
    # pve_step_many (still ticks, staged ticks, chunked launches) == single ticks, bit for bit
    t0 = time.time()
    scenarios.check_step_many("hip", src, n_envs=24, capacity=cap, rate=rate, prefill=300, chunks=(1, 2, 25, 60, 7, 100),
                              trajectory_chunk=20, seed=300 + int(rate) + a.seed)
    print("pve_step_many == single ticks (%s, %.0f veh/h/lane, cap %d) OK, %.0f s" % (src, rate, cap, time.time() - t0))
for ln, rate, cap in ((4, 2000.0, 64), (8, 1600.0, 128)):
    guarded("%d lanes, quantised 1.0: %d ticks x 24 envs" % (ln, a.ticks),
            lambda: scenarios.check_geo_fuzz_vs_oracle("hip", ln, n_envs=24, capacity=cap, ticks=a.ticks, rate=rate,
                                                       seed=102 + ln + a.seed, quantize=1.0))
    guarded("%d lanes, unquantised: %d ticks x 24 envs" % (ln, a.ticks),
            lambda: scenarios.check_geo_fuzz_vs_oracle("hip", ln, n_envs=24, capacity=cap, ticks=a.ticks, rate=rate * 0.8,
                                                       seed=202 + ln + a.seed))
