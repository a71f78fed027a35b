#!/usr/bin/env python3
"""Same-process A/B of launch shapes for a short call (default 20 ticks, the driver's timed region): the persistent
work-queue launch with several item schedules against the two stream-pipelined sub-batches, alternating, median of R
repetitions.  Diagnostics only.  PVE_TAPER_TAIL is an A/B knob of the KNOB build of the library (the product library
reads no environment variable on its launch path): `make -C pve-mcc_for_unsignalized_intersection_amd/csrc knobs`, then
PVE_LIBRARY_PATH=build/libpveenv_knobs.so python tools/ab_launch_shapes.py."""
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
n, cap = 4096, int(os.environ.get("AB_CAP", "128"))
K = int(os.environ.get("AB_K", "20"))
R = int(os.environ.get("AB_REPS", "15"))
rate = 1100.0 if cap == 128 else 350.0
arr = synthetic_arrivals(n, rate=rate, horizon_s=(300 + (K + 5) * 12 * R) * 0.1 + 60, seed=20250213)
pool = torch.as_tensor(bench.action_pool(n, cap, 99), device=dev)
one = pve_mcc_amd.BatchedIntersections(n, cap, arr, device=dev)
two = pve_mcc_amd.PipelinedIntersections(n, cap, arr, n_sub=2, device=dev)
for e in (one, two):
    e.reset()
    e.set_action_pool(pool)
torch.cuda.synchronize()
variants = [("2 streams, launches of 5", two, dict(chunk=5), None),
            ("persistent 9,8 + 3", one, dict(chunk=9, persistent=True), "3"),
            ("one launch of 4096 x 20", one, dict(chunk=0), None)]
# more item schedules: AB_SHAPES="12:5,3;10:6,4;..." = chunk_ticks : PVE_TAPER_TAIL (knob build of the library)
for spec in filter(None, os.environ.get("AB_SHAPES", "12:5,3;11:6,3;14:3,3;17:3;7:;6:3").split(";")):
    ch, tail = spec.split(":")
    variants.insert(-1, ("persistent chunk %s + tail [%s]" % (ch, tail), one, dict(chunk=int(ch), persistent=True), tail))
calls = {}


def run(env, k, kw, tail):
    if tail is not None:
        os.environ["PVE_TAPER_TAIL"] = tail
    key = (id(env), k, tuple(sorted(kw.items())))
    if key not in calls:
        calls[key] = env.prepare_step_many(k, **kw)
    calls[key]()


for e in (one, two):
    run(e, 300, dict(chunk=25), None)
torch.cuda.synchronize()
res = {v[0]: [] for v in variants}
for rep in range(R):
    for name, env, kw, tail in variants:
        run(env, 5, dict(chunk=5), None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(env, K, kw, tail)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) * 1e6)
for name, *_ in variants:
    v = np.array(res[name])
    print("%-28s median %.1f us (%.2f us/tick)  min %.1f  p25 %.1f  p75 %.1f" % (name, np.median(v), np.median(v) / K, v.min(),
                                                                            np.percentile(v, 25), np.percentile(v, 75)))
m1, m2 = one.metrics(), two.metrics()
print("ticks", m1["ticks"] / n, m2["ticks"] / n, "overflow", m1["overflow"], m2["overflow"])
