#!/usr/bin/env python3
"""Why is a short timed region slower per tick than a long one?  Times regions of K ticks (persistent roll-out, one batch
of 4096 x 128) for several K, with and without an idle gap in front, and prints us per tick.  Diagnostics only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pve_mcc_amd
from pve_mcc_amd.arrivals import synthetic_arrivals

dev = torch.device("cuda", 0)
n, cap = 4096, 128
arr = synthetic_arrivals(n, rate=1100.0, horizon_s=400.0, seed=20250213)
pers = int(os.environ.get("PROBE_PERSISTENT", "1"))
chunk = int(os.environ.get("PROBE_CHUNK", "5"))
if pers:
    env = pve_mcc_amd.BatchedIntersections(n, cap, arr, device=dev)
else:
    env = pve_mcc_amd.PipelinedIntersections(n, cap, arr, n_sub=2, device=dev)
env.reset()
env.set_action_pool(torch.as_tensor(bench.action_pool(n, cap, 99), device=dev))
calls = {}


def run(k):
    if k not in calls:
        calls[k] = env.prepare_step_many(k, chunk=chunk, persistent=bool(pers))
    calls[k]()


run(300)
torch.cuda.synchronize()
for gap_ms in (0.0, 1.0, 20.0):
    for K in (20, 20, 40, 80, 160, 20):
        run(5)
        torch.cuda.synchronize()
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("gap %5.1f ms  K %4d  %.2f us/tick  (%.0f us)" % (gap_ms, K, dt / K * 1e6, dt * 1e6), flush=True)
# back-to-back: two regions of 20 enqueued without a gap, timed separately by events
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
run(5); torch.cuda.synchronize()
s = torch.cuda.current_stream(dev) if pers else None
if pers:
    e[0].record(); run(20); e[1].record(); run(20); e[2].record(); run(20); e[3].record()
    torch.cuda.synchronize()
    print("back-to-back regions of 20 ticks: %.1f  %.1f  %.1f us" % tuple(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(3)))
