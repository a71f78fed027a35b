#!/usr/bin/env python3
"""Launch the traffic-probe kernel (known bytes: 72 B read per slot) N times; run under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` to calibrate FETCH_SIZE for 8 B / 4 B per-lane loads."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pve_mcc_amd  # noqa: E402
from pve_mcc_amd.arrivals import synthetic_arrivals  # noqa: E402

n_envs, cap = 4096, 128
env = pve_mcc_amd.BatchedIntersections(n_envs, cap, synthetic_arrivals(n_envs, 1100.0, 30.0), device="cuda:0",
                                       outputs=("env_out",))
env.reset()
sink = torch.zeros(n_envs, dtype=torch.int32, device="cuda:0")
# evict the caches between launches with a 1 GiB fill so every probe reads from HBM
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")
for _ in range(10):
    big.fill_(1.0)
    env.lib.pve_debug_traffic_probe(env._h, C.c_void_p(sink.data_ptr()))
torch.cuda.synchronize()
print("known read bytes per probe launch: %d (%.1f KiB)" % (n_envs * cap * 72, n_envs * cap * 72 / 1024))
