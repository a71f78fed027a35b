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
a = ap.parse_args()
t0 = time.time()
c, l = scenarios.check_fuzz_vs_oracle("hip", n_envs=32, capacity=128, ticks=a.ticks, rate=1100.0, seed=101, quantize=0.5)
print("12 lanes: %d ticks x 32 envs OK, collisions %d, locks %d, %.0f s" % (a.ticks, c, l, time.time() - t0))
for ln, rate, cap in ((4, 2000.0, 64), (8, 1600.0, 128)):
    t0 = time.time()
    c, l = scenarios.check_geo_fuzz_vs_oracle("hip", ln, n_envs=24, capacity=cap, ticks=a.ticks, rate=rate, seed=102 + ln,
                                              quantize=1.0)
    print("%d lanes: %d ticks x 24 envs OK, collisions %d, locks %d, %.0f s" % (ln, a.ticks, c, l, time.time() - t0))
