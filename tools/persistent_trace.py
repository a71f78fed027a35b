#!/usr/bin/env python3
"""Per-item timeline of one persistent pve_step_many call (pve_debug_phase_cycles armed -> k_rollout<.., PERS> stamps every
(intersection, chunk) item: dequeue start, item known, predecessor done, state loaded, flush issued, handed on; 100 MHz
clock).  Diagnostics only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pve_mcc_amd
from pve_mcc_amd.arrivals import synthetic_arrivals

dev = torch.device("cuda", 0)
n, cap = 4096, 128
K = int(os.environ.get("TRACE_K", "20"))
chunk = int(os.environ.get("TRACE_CHUNK", "6"))
arr = synthetic_arrivals(n, rate=1100.0, horizon_s=400.0, seed=20250213)
env = pve_mcc_amd.BatchedIntersections(n, cap, arr, device=dev)
env.reset()
env.set_action_pool(torch.as_tensor(bench.action_pool(n, cap, 99), device=dev))
env.step_many(300, chunk=25, persistent=True)
env.step_many(K, chunk=chunk, persistent=True)
torch.cuda.synchronize()
n_chunks = K                                          # upper bound (the library tapers the last items of a call)
buf = torch.zeros(n_chunks * n * 8, dtype=torch.int64, device=dev)
env.lib.pve_debug_phase_cycles(env._h, C.c_void_p(buf.data_ptr()))
env.step_many(K, chunk=chunk, persistent=True)
torch.cuda.synchronize()
env.lib.pve_debug_phase_cycles(env._h, None)
t = buf.cpu().numpy().reshape(n_chunks, n, 8).astype(np.float64)
n_chunks = int((t[:, 0, 0] != 0).sum())
t = t[:n_chunks]
t0 = t[:, :, 0].min()
us = (t[:, :, :6] - t0) / 100.0                      # 100 MHz -> us
who = t[:, :, 6].astype(np.int64)
wg, xcc = who & 0xFFFFFFFF, who >> 32
print("call span %.1f us; %d items, %d workgroups, XCDs seen %s" % (us[:, :, 5].max(), n_chunks * n, len(np.unique(wg)), np.unique(xcc)))
for c in range(n_chunks):
    u = us[c]
    print("chunk %d: dequeue %.1f..%.1f  start(mean) %.1f  end(mean) %.1f end(max) %.1f | dequeue %.2f wait %.2f load %.2f ticks %.2f drain %.2f total %.2f us"
          % (c, u[:, 0].min(), u[:, 0].max(), u[:, 0].mean(), u[:, 5].mean(), u[:, 5].max(), (u[:, 1] - u[:, 0]).mean(),
             (u[:, 2] - u[:, 1]).mean(), (u[:, 3] - u[:, 2]).mean(), (u[:, 4] - u[:, 3]).mean(), (u[:, 5] - u[:, 4]).mean(),
             (u[:, 5] - u[:, 0]).mean()))
# per-XCD shard affinity and balance
for x in np.unique(xcc):
    m = xcc == x
    envs = np.unique(np.nonzero(m)[1] % 8)
    print("XCD %d: %d items, env %% 8 in %s, workgroups %d, last end %.1f" % (x, m.sum(), envs, len(np.unique(wg[m])), us[:, :, 5][m].max()))
# item duration by start time (does the tick rate change over the call?)
dur = (us[:, :, 4] - us[:, :, 3]).ravel()
st = us[:, :, 3].ravel()
edges = np.linspace(0, st.max() + 1, 9)
for a, b in zip(edges[:-1], edges[1:]):
    m = (st >= a) & (st < b)
    if m.any():
        print("items starting in [%.0f, %.0f) us: %5d, ticks phase mean %.1f us (%.2f per tick) p10 %.1f p90 %.1f" %
              (a, b, m.sum(), dur[m].mean(), dur[m].mean() / chunk, np.percentile(dur[m], 10), np.percentile(dur[m], 90)))
print("(per-tick figures above assume items of %d ticks; the library tapers the last items)" % chunk)
# idle time per workgroup between its items
idle = []
for w in np.unique(wg)[:256]:
    m = wg == w
    s, e = np.sort(us[:, :, 0][m]), np.sort(us[:, :, 5][m])
    idle.append((s[1:] - e[:-1]).sum() if len(s) > 1 else 0.0)
print("gap between a workgroup's items (sum over its items, mean over 256 workgroups): %.2f us" % np.mean(idle))
busy_end = np.array([us[:, :, 5][wg == w].max() for w in np.unique(wg)])
print("workgroup finish times: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(busy_end, [10, 50, 90, 100])))
