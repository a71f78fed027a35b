#!/usr/bin/env python3
"""How many vehicle slots does a rate need?  The CPU oracle (no capacity, like the reference: ref traffic_interaction_scene.py:378-433)
runs n_envs synthetic streams for `ticks` ticks under BASELINE.md 3's sin tape and reports the peak number of vehicles alive at
once per env -- the evidence behind the capacity a bench leg is run with (BASELINE config 2's stated 500 veh/h/lane
peaks above 64 slots in some of 4096 envs; does it stay below 96 / 128?).  Test infrastructure (imports oracle/).
  python tools/peak_population.py [--rate 500] [--envs 4096] [--ticks 2300] [--threads 8]"""
import argparse
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import OracleEnv  # noqa: E402
from pve_mcc_amd.arrivals import synthetic_arrivals  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rate", type=float, default=500.0)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--ticks", type=int, default=2300)
ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
ap.add_argument("--seed", type=int, default=20250213 + 15485863)      # bench.py's cap64_on_spec streams (rank 0)
a = ap.parse_args()
arr = synthetic_arrivals(a.envs, rate=a.rate, horizon_s=a.ticks * 0.1 + 20.0, seed=a.seed, lane_num=12)
peak = np.zeros(a.envs, np.int64)


def work(k):
    for e in range(k, a.envs, a.threads):
        o = OracleEnv(arr[e])
        o.run(a.ticks, 1, 1.0, 0)
        peak[e] = o.peak_alive


ths = [threading.Thread(target=work, args=(k,)) for k in range(a.threads)]
[t.start() for t in ths]
[t.join() for t in ths]
print("%d envs x %d ticks at %.0f veh/h/lane: peak alive per env  max %d  p99.9 %d  p99 %d  p50 %d;  envs above 64: %d, above 96: %d, above 128: %d"
      % (a.envs, a.ticks, a.rate, peak.max(), np.percentile(peak, 99.9), np.percentile(peak, 99), np.percentile(peak, 50),
         (peak > 64).sum(), (peak > 96).sum(), (peak > 128).sum()))
