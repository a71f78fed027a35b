#!/usr/bin/env python3
"""Launches of k_tick<128> truncated behind one phase (pve_debug_stop_phase) on a frozen steady-state batch: run under
`rocprofv3 --pmc ...` by tools/phase_counters.sh; the counters of stop = n minus those of stop = n - 1 are phase n's.
usage: python tools/phase_probe.py <stop phase, -1 = full tick> [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pve_mcc_amd
from pve_mcc_amd.arrivals import synthetic_arrivals

stop = int(sys.argv[1])
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
n, cap = 2048, 128
arr = synthetic_arrivals(n, rate=1100.0, horizon_s=80.0, seed=20250213)
env = pve_mcc_amd.BatchedIntersections(n, cap, arr, device=dev)
env.reset()
pool = torch.as_tensor(bench.action_pool(n, cap, 99), device=dev)
env.set_action_pool(pool)
env.step_many(400, chunk=25)                               # steady state (k_rollout: not the kernel that is counted)
torch.cuda.synchronize()
# the counted launches are every SECOND k_tick launch: a full tick in between advances the state and the tick counter (RANK's
# claim tag is a stamp of the tick: on a frozen counter every claim would meet its own left-over stamp in LDS)
for k in range(launches):
    env.lib.pve_debug_stop_phase(env._h, -1)
    env.step(pool[k % 16])
    env.lib.pve_debug_stop_phase(env._h, stop)
    env.step(pool[(k + 1) % 16])
torch.cuda.synchronize()
env.lib.pve_debug_stop_phase(env._h, -1)
m = env.metrics()
print("stop", stop, "launches", launches, "mean alive", m["alive_steps"] / m["ticks"])
