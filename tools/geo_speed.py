"""Throughput of the general-geometry kernel (k_tick_geo; lane_num 4 / 8) on synthetic streams:
env-steps/s = envs x capacity x ticks / time, HIP-event timed.  Usage: python tools/geo_speed.py [--envs 4096]"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pve_mcc_amd  # noqa: E402
from pve_mcc_amd.arrivals import synthetic_arrivals, synthetic_intentions  # noqa: E402
from pve_mcc_amd.batched import BatchedIntersections  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=300)
    args = ap.parse_args()
    for lane_num, cap, rate, kw in ((4, 64, 1800.0, {}), (8, 128, 1500.0, {}), (12, 128, 1100.0, dict(general_path=True)),
                                    (12, 128, 1100.0, {})):
        horizon = (args.steps + args.warmup) * 0.1 + 30
        arr = synthetic_arrivals(256, rate, horizon, lane_num=lane_num)
        arr = torch.as_tensor(arr).repeat((args.envs + 255) // 256, 1, 1)[:args.envs]
        ch = None
        if lane_num == 8:
            ch = torch.as_tensor(synthetic_intentions(256, arr.shape[1])).repeat((args.envs + 255) // 256, 1, 1)[:args.envs]
        b = BatchedIntersections(args.envs, cap, arr, intentions=ch, lane_num=lane_num, **kw)
        b.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        pool = [(torch.rand(args.envs, cap, generator=g, device="cuda", dtype=torch.float64) * 2 - 1) for _ in range(8)]
        for t in range(args.warmup):
            b.step(pool[t % 8])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(args.steps):
            b.step(pool[t % 8])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        m = b.metrics()
        print(json.dumps(dict(lane_num=lane_num, path="general" if (lane_num != 12 or kw) else "fast", capacity=cap,
                              envs=args.envs, ms_per_tick=ms, env_steps_per_s=args.envs * cap / (ms * 1e-3),
                              mean_alive=m["alive_steps"] / m["ticks"], mean_ctl=m["ctl_steps"] / m["ticks"],
                              overflow=m["overflow"])))


if __name__ == "__main__":
    main()
