#!/usr/bin/env python3
"""Experiment (GPU): closed loop with the sub-batches staggered by half a step: even sub-batches run actor -> tick, odd
ones tick -> actor (their first actor is issued up front), so that one sub-batch's actor is enqueued beside another's tick.
Usage: python tools/actor_stagger.py [--pipeline 2] [--steps 300]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pve_mcc_amd
from pve_mcc_amd.arrivals import synthetic_arrivals
ap = argparse.ArgumentParser()
ap.add_argument("--pipeline", type=int, default=2)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--stagger", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, cap = 4096, 128
arr = synthetic_arrivals(n, rate=1000.0, horizon_s=(a.steps + 400) * 0.1 + 20, seed=20250213)
env = pve_mcc_amd.PipelinedIntersections(n, cap, arr, n_sub=a.pipeline, device=dev, obs_dtype=torch.float32)
env.reset()
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "actor_66.npz"))
env.set_actor({k: z[k] for k in z.files})
for _ in range(350):
    env.step_with_actor()
torch.cuda.synchronize()
subs = env.subs
def run(K):
    if a.stagger:
        for k, s in enumerate(subs):
            if k & 1: s.act()
        for _ in range(K):
            for k, s in enumerate(subs):
                if k & 1:
                    s.step(s._actor_actions); s.act()
                else:
                    s.step_with_actor()
    else:
        for _ in range(K):
            env.step_with_actor()
run(20)
torch.cuda.synchronize(); t0 = time.perf_counter()
run(a.steps)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pipeline %d stagger %d: %.1f us per step" % (a.pipeline, a.stagger, dt / a.steps * 1e6))
