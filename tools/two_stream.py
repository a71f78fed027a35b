"""Experiment: the 4096-env batch as two 2048-env handles on two HIP streams (the two populations desynchronise, so the
LOAD / FIN bursts of one overlap the compute phases of the other).  python tools/two_stream.py [--split 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pve_mcc_amd  # noqa: E402,F401
from pve_mcc_amd.arrivals import synthetic_arrivals  # noqa: E402
from pve_mcc_amd.batched import BatchedIntersections  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--split", type=int, default=2)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--warmup", type=int, default=300)
a = ap.parse_args()
n = a.envs // a.split
arr = synthetic_arrivals(a.envs, 1100.0, (a.steps + a.warmup) * 0.1 + 20)
pool = torch.as_tensor(bench.action_pool(a.envs, 128, 99), device="cuda")
streams = [torch.cuda.Stream() for _ in range(a.split)]
hs = []
for k in range(a.split):
    with torch.cuda.stream(streams[k]):
        h = BatchedIntersections(n, 128, arr[k * n:(k + 1) * n], stream=streams[k])
        h.reset()
        hs.append(h)
pools = [pool[:, k * n:(k + 1) * n].contiguous() for k in range(a.split)]
torch.cuda.synchronize()


def run(steps, t0):
    for t in range(t0, t0 + steps):
        for k in range(a.split):
            hs[k].step(pools[k][t % bench.N_POOL])


run(a.warmup, 0)
torch.cuda.synchronize()
w0 = time.perf_counter()
run(a.steps, a.warmup)
torch.cuda.synchronize()
dt = time.perf_counter() - w0
print("split %d x %d envs: %.2f us per tick of all %d envs, %.3e env-steps/s" % (a.split, n, dt / a.steps * 1e6, a.envs,
                                                                              a.envs * 128 * a.steps / dt))
