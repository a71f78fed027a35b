#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused HIP tick at BASELINE.json's headline config
(4096 parallel 12-lane intersections x 128 vehicle slots per GPU, synthetic Poisson arrivals at
1100 veh/h/lane), weak-scaled env-parallel over N GPUs (one process per GPU, no data-path collective;
one RCCL all-gather of the metrics vector after the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one fused tick (all step() calls + scene_update() + delete_vehicle()) of every env of the
rank = one kernel launch.  Inputs (arrival streams, action pool) are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_FP64 = 380.0      # algorithmic bytes per vehicle-slot-step, FP64 layout
B_ALG_OBS_F32 = 268.0   # the same with float32 observation rows (--obs-f32): 380 - 28 x 4 (SURVEY.md §8d, FP32 output)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (guides/MI355X_MICROARCH.md)
N_POOL = 16


def action_pool(n_envs, cap, seed):
    """Synthetic action tape resident in HBM: pool[k][env][slot] = float32(sin(0.37*u + 0.05*k*7))-like
    values in [-1, 1] (the SURVEY §8d 'sin, A=1' pattern, indexed by slot so that it needs no feedback)."""
    rng = np.random.default_rng(seed)
    phase = rng.uniform(0, 2 * np.pi, size=(1, n_envs, cap))
    k = np.arange(N_POOL, dtype=np.float64)[:, None, None]
    a = np.sin(phase + 0.37 * np.arange(cap)[None, None, :] + 0.05 * 7 * k)
    return a.astype(np.float32).astype(np.float64)


def cpu_baseline(arr, pool, cap, warm, ticks, lane_num=12, choice=None):
    """The CPU oracle (oracle/pve_oracle.c, a plain-C port of the reference algorithm) timed on this
    host's cores on a bounded sample of the same workload: the first `n` envs of the same arrival
    tensor with the same action pool, one env per thread-task, all cores busy."""
    from oracle.oracle import OracleEnv
    cores = os.cpu_count() or 1
    n = min(arr.shape[0], max(cores * 8, 16))
    if lane_num == 12:
        envs = [OracleEnv(arr[e]) for e in range(n)]
    else:
        from oracle.oracle_geo import OracleGeoEnv
        envs = [OracleGeoEnv(arr[e], lane_num, choice=None if choice is None else choice[e]) for e in range(n)]
    res = [None] * n
    nxt = [0]
    lock = threading.Lock()

    def worker(phase, nt, t0):
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= n:
                return
            res[i] = envs[i].run_pool(nt, pool[:, i, :], t0)

    def run(nt, t0):
        nxt[0] = 0
        ths = [threading.Thread(target=worker, args=(0, nt, t0)) for _ in range(cores)]
        t = time.perf_counter()
        [x.start() for x in ths]
        [x.join() for x in ths]
        return time.perf_counter() - t

    run(warm, 0)
    dt = run(ticks, warm)
    alive = sum(r[0] for r in res)
    return dict(value=n * cap * ticks / dt, unit="env-steps/s", cores=cores, kind="port",
                alive_steps_per_s=alive / dt,
                sample="%d envs x %d ticks (after %d warm-up ticks) of the same synthetic workload, %d threads, "
                       "%.2f s wall" % (n, ticks, warm, cores, dt))


def pmc_traffic(n_envs, cap, outputs, actor):
    """HBM bytes per k_tick launch from the PMC counters (FETCH_SIZE x gfx950 correction + WRITE_SIZE). Counters
    cannot be read from inside the process, so they come from the committed rocprofv3 passes of this very
    command (tools/collect_profiles.sh -> profiles/r*_traffic.json); null when the config differs."""
    import glob
    default_outputs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")
    if actor or cap != 128 or tuple(outputs) != default_outputs:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    t = json.load(open(files[-1]))
    if int(t.get("envs_per_launch", 4096)) != n_envs:      # measured on launches of another size
        return None, None
    return t["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)


def main(argv=None, env_factory=None):
    """env_factory: tests inject a factory (device, backend, builder) to exercise the rank plumbing and the
    JSON contract without a GPU; the product run never passes it."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--capacity", type=int, default=128)
    ap.add_argument("--rate", type=float, default=None, help="veh/h/lane (default 1100 at cap 128, 500 at cap 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--outputs", default="obs_post,reward,flags,nbr,new_slot,env_out")
    ap.add_argument("--lane-num", type=int, default=12, choices=(12, 8, 4),
                    help="intersection layout; 12 = BASELINE metric (k_tick), 4 / 8 = SURVEY 8 f4 (k_tick_geo)")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="free-running sub-batches per GPU, each on its own HIP stream (PipelinedIntersections); 1 = one "
                         "launch over all envs per step")
    ap.add_argument("--obs-f32", action="store_true",
                    help="float32 observation rows (PVE_CFG_OBS_F32; SURVEY 8d's FP32-output variant, 268 B algorithmic); "
                         "the headline / BASELINE metric is the float64 parity layout (380 B)")
    ap.add_argument("--actor", action="store_true",
                    help="BASELINE config 5: close the loop on the device (k_actor -> k_tick per step) instead of the action pool")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    import torch.distributed as dist
    emu = env_factory is not None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if emu else torch.device("cuda", local_rank)
    if not emu:
        torch.cuda.set_device(dev)

    def sync():
        if not emu:
            torch.cuda.synchronize(dev)

    import pve_mcc_amd
    from pve_mcc_amd.arrivals import synthetic_arrivals, synthetic_intentions
    from pve_mcc_amd.distributed import gather_metrics

    cap, n_envs, lane_num = args.capacity, args.envs, args.lane_num
    rate = args.rate or {12: (1100.0 if cap == 128 else 500.0), 8: 1500.0, 4: 1800.0}[lane_num]
    K, W = args.steps, args.warmup
    horizon = (K + W) * 0.1 + 20.0
    # weak scaling: every rank owns its own n_envs environments (global env index = rank*n_envs + e)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=horizon, seed=20250213 + rank * n_envs, lane_num=lane_num)
    choice = synthetic_intentions(n_envs, arr.shape[1], seed=20250213 + rank * n_envs) if lane_num == 8 else None
    pool_np = action_pool(n_envs, cap, seed=99 + rank)
    outputs = tuple(x for x in args.outputs.split(",") if x)
    n_sub = max(1, min(args.pipeline, n_envs))
    if emu:
        env = env_factory(n_envs, cap, arr, outputs)
        n_sub = getattr(env, "n_sub", 1)
    elif n_sub == 1:
        env = pve_mcc_amd.BatchedIntersections(n_envs, cap, arr, device=dev, outputs=outputs, lane_num=lane_num,
                                               intentions=choice,
                                               obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
    else:
        # the envs are independent: n_sub free-running sub-batches on their own streams pipeline the ticks (the chip-wide
        # LOAD / FIN bursts of one sub-batch overlap the compute phases of the other), DESIGN.md 5
        env = pve_mcc_amd.PipelinedIntersections(n_envs, cap, arr, n_sub=n_sub, device=dev, outputs=outputs,
                                                 lane_num=lane_num, intentions=choice,
                                                 obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
    pool = torch.as_tensor(pool_np, device=dev)
    env.reset()
    sub_streams = getattr(env, "streams", None) if not emu else None
    if args.actor:
        wpath = os.path.join(ROOT, "tests", "golden", "actor_66.npz")
        z = np.load(wpath)
        env.set_actor({k: z[k] for k in z.files})       # the reference's pretrained actor (model_data/baseline/66.cptk)

    def one_step(t):
        if args.actor:
            env.step_with_actor()
        else:
            env.step(pool[t % N_POOL])

    for t in range(W):
        one_step(t)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    m0 = env.metrics()
    # HIP events on the streams the kernels are launched on: torch's current stream for one batch, every sub-batch
    # stream for the pipelined form (start / end of the K launches of that stream)
    if not emu:
        ev_streams = sub_streams if sub_streams else [torch.cuda.current_stream(dev)]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ev_streams]
    t0 = time.perf_counter()
    if not emu:
        for (e0, _), st in zip(evs, ev_streams):
            e0.record(st)
    for t in range(W, W + K):
        one_step(t)
    if not emu:
        for (_, e1), st in zip(evs, ev_streams):
            e1.record(st)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    # average duration of one launch on its stream (K back-to-back launches per stream)
    gpu_ms = (sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)) if not emu else wall * 1e3
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    m1 = env.metrics()
    delta = {k: m1[k] - m0[k] for k in m1}
    per_rank, tot = gather_metrics(delta, dev)          # the single RCCL all-gather (metrics only)

    if rank == 0:
        slot_steps = float(cap) * n_envs * K * world
        value = slot_steps / wall
        kern_s = gpu_ms * 1e-3 / K                      # per launch (n_envs / n_sub envs), launches of the n_sub streams overlap
        b_alg = B_ALG_OBS_F32 if args.obs_f32 else B_ALG_FP64
        envs_per_launch = n_envs / float(n_sub)
        per_launch = b_alg * cap * envs_per_launch / kern_s / 1e9
        achieved = per_launch * n_sub                    # n_sub launches of the kernel are in flight at any time
        traffic, traffic_src = pmc_traffic(int(envs_per_launch), cap, outputs, args.actor or lane_num != 12 or args.obs_f32)
        line = {
            "metric": "env-steps/sec (vehicles x envs x steps/s) at 128 veh x 4096 envs",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not emu else "synthetic (injected test environment: timings meaningless)",
            "config": {"workload": "%d parallel %d-lane intersections x %d vehicle slots per GPU, synthetic Poisson "
                                   "arrivals %.0f veh/h/lane, %s, fused step+scene_update+delete tick"
                                   % (n_envs, lane_num, cap, rate, "on-device MADDPG actor (pretrained 66.cptk weights) closing the loop"
                                      if args.actor else "sin action pool"),
                       "envs_per_gpu": n_envs, "capacity": cap,
                       "parallelism": "env-parallel x%d" % world + (", %d stream-pipelined sub-batches of %d envs per GPU"
                                                                   % (n_sub, int(envs_per_launch)) if n_sub > 1 else ""),
                       "outputs": list(outputs), "obs_dtype": "f32" if args.obs_f32 else "f64"},
            "alive_steps_per_s": tot["alive_steps"] / wall,
            "ctl_steps_per_s": tot["ctl_steps"] / wall,
            "mean_alive_per_env": tot["alive_steps"] / (K * n_envs * world),
            "overflow": tot["overflow"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "kernel": ("k_tick<%d>" if lane_num == 12 else "k_tick_geo<%d>") % cap, "kernel_ms": kern_s * 1e3,
                         "alg_bytes_per_slot_step": b_alg, "envs_per_launch": int(envs_per_launch),
                         "concurrent_launches": n_sub, "per_launch_achieved": per_launch,
                         "definition": "achieved = algorithmic bytes per launch / average launch duration (HIP events on "
                                       "the launching stream) x launches in flight; with one sub-batch this is the plain "
                                       "per-launch figure"},
        }
        if args.actor:
            line["roofline"]["note"] = "kernel_ms = k_actor + k_tick per step; achieved uses the tick's algorithmic bytes only"
        if not args.no_cpu_baseline and not args.actor and world == 1:      # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(arr, pool_np, cap, min(W, 300), min(K, 200), lane_num, choice)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                # every rank leaves together (rank 0 may still be timing the CPU baseline)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
