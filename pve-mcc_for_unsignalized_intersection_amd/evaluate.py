"""`python -m pve_mcc_amd.evaluate --mat data/test/arvTimeNewVeh_new_1000_12.mat --weights actor.npz`

The evaluation protocol of the reference's `main.py:test()` / `batch_test()` (main.py:367-441, 530-584) run
entirely on the GPU: the pretrained MADDPG actor (on-device inference) closes the loop around the fused tick, and
the aggregate lines of main.py:523-526 / 576-581 are printed from the metrics vector:

    vehicle number, collisions occurred number, collisions rate, pT-m (mean passing time), mean jerk, lock_num

One arrival stream can be replicated over many environments (`--envs`, independent copies are identical) or a
synthetic batch can be evaluated (`--synthetic RATE`)."""
import argparse
import json

import numpy as np

from .arrivals import load_arrival_mat, pad_stream, synthetic_arrivals, synthetic_intentions
from .batched import BatchedIntersections


def evaluate(arrivals, weights, ticks=1000, n_envs=1, capacity=128, device="cuda", intentions=None, **config):
    """-> dict with the quantities main.py prints at the end of test() (main.py:523-526).
    config: reference constructor arguments (vm, lane_num = 12 | 8 | 4, ...; main.py --lane_num, :101)."""
    env = BatchedIntersections(n_envs, capacity, arrivals, device=device, intentions=intentions,
                               outputs=("obs_post", "reward", "flags", "env_out"), **config)
    env.set_actor(weights)
    env.reset()
    for _ in range(ticks):
        env.step_with_actor()
    m = env.metrics()
    dt = float(env.cfg.deltaT)
    return {
        "vehicles": m["spawned"], "passed": m["passed"], "collisions": m["collided"],
        "collisions_rate": m["collided"] / max(m["spawned"], 1.0),
        "pT_m": m["passed_steps"] / (m["passed"] + 1e-4) * dt,            # main.py:526
        "jerk_mean": m["sum_jerk"] / max(m["passed"], 1.0),                # main.py:524
        "lock_num": m["locks"], "reward_mean": m["sum_reward"] / max(m["ctl_steps"], 1.0),
        "ticks": ticks, "n_envs": n_envs, "overflow": m["overflow"],
    }


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mat", help="MATLAB v5 arrival stream (variable arvTimeNewVeh)")
    ap.add_argument("--synthetic", type=float, help="instead of --mat: synthetic Poisson arrivals at RATE veh/h/lane")
    ap.add_argument("--weights", required=True, help=".npz with the 12 actor tensors (tools/extract_actor.py)")
    ap.add_argument("--ticks", type=int, default=1000)
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--capacity", type=int, default=128)
    ap.add_argument("--vm", type=float, default=5.0)
    ap.add_argument("--lane-num", type=int, default=12, choices=(12, 8, 4), help="main.py --lane_num (:101)")
    ap.add_argument("--seed", type=int, default=20250213, help="synthetic streams / 8-lane intention draws")
    a = ap.parse_args()
    z = np.load(a.weights)
    weights = {k: z[k] for k in z.files}
    if a.mat:
        arr = pad_stream(load_arrival_mat(a.mat))
    else:
        arr = synthetic_arrivals(a.envs, rate=a.synthetic or 1000.0, horizon_s=a.ticks * 0.1 + 30, seed=a.seed,
                                 lane_num=a.lane_num)
    if arr.shape[-1] != a.lane_num:
        ap.error("the arrival stream has %d columns, --lane-num is %d" % (arr.shape[-1], a.lane_num))
    draws = None
    if a.lane_num == 8:                        # the reference draws them with random.randint (ref :390)
        draws = synthetic_intentions(a.envs, arr.shape[-2], seed=a.seed)
        draws = draws if arr.ndim == 3 else draws[0]
    res = evaluate(arr, weights, ticks=a.ticks, n_envs=a.envs, capacity=a.capacity, vm=a.vm, lane_num=a.lane_num,
                   intentions=draws)
    print("vehicle number: %d; collisions occurred number: %d; collisions rate: %s" % (
        res["vehicles"], res["collisions"], res["collisions_rate"]))
    print("pT-m: %.3f s; mean jerk: %.3f; lock_num: %d; mean reward: %.4f" % (
        res["pT_m"], res["jerk_mean"], res["lock_num"], res["reward_mean"]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
