"""Randomised pin of the CPU oracles to the LIVE reference (build container only: needs /root/reference): layout, density,
action scale and quantisation drawn per run; every field, every tick, 1e-12.   python tools/soak_reference.py [--runs 60]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.oracle_geo import OracleGeoEnv  # noqa: E402
from oracle.record import compare_records  # noqa: E402
from tests.golden import ref_harness as rh  # noqa: E402
from tests.golden.gen_golden_geo import make_stream  # noqa: E402
from tests.test_oracle_vs_reference_fuzz import random_tape  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
MEAN = {12: (2.4, 6.0), 8: (1.7, 4.0), 4: (1.1, 3.0)}          # mean seconds between arrivals per lane
tot = 0
for k in range(a.runs):
    ln = int(rng.choice([12, 12, 8, 4]))
    mean = float(rng.uniform(*MEAN[ln]))
    scale = float(rng.choice([0.3, 1.0, 2.0, 3.0]))
    quant = rng.choice([0.0, 0.0, 0.25, 0.5, 1.0, 3.0])
    quant = None if quant == 0.0 or quant > 2 * scale else float(quant)
    ticks = int(rng.integers(250, 600))
    seed = int(rng.integers(1, 1 << 30))
    arr, choice = make_stream(ln, 400, mean, seed)
    ref = rh.GeoRefRunner(arr, ln, random_tape(seed, scale, quant), choice=choice, want_state=True)
    t0 = time.time()
    try:
        orc = OracleGeoEnv(arr, ln, choice=choice)
        coll = lock = 0
        for t in range(ticks):
            ra = ref.tick()
            rb = orc.tick(ref.tape, want_state=True)
            compare_records(ra, rb, tol=1e-12, label="run %d" % k)
            coll += int(ra["collisions"]); lock += int(ra["lock"])
    finally:
        ref.close()
    tot += ticks
    print("run %2d: %2d lanes mean %.1f s |a|<=%.1f quant %-5s %3d ticks seed %d OK (collisions %d, locks %d) %.0f s"
          % (k, ln, mean, scale, quant, ticks, seed, coll, lock, time.time() - t0), flush=True)
print("%d runs, %d ticks: the oracles equal the live reference in every field" % (a.runs, tot))
