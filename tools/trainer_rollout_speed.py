"""Diagnostics: the trainer's roll-out through the work queue (obs_pre + state_pre of every tick into trajectory blocks, float32
rows): us per tick of 4096 x 128 intersections.  python tools/trainer_rollout_speed.py [label]
LANE_NUM=4 / 8 / 12: the layout;  SOURCE=actor: the closed loop (the on-device actor chooses the actions) instead of the pool;
PERSISTENT=0: chunked launches.  With PVE_LIBRARY_PATH=build/libpveenv_knobs.so, PVE_NO_ROLLOUT_ACTOR=1 gives the per-tick form
(actor launch + tick launch) for an A/B."""
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench, pve_mcc_amd
from pve_mcc_amd.arrivals import synthetic_arrivals
import os
n, cap, K = 4096, 128, 20
LN = int(os.environ.get("LANE_NUM", "12"))            # LANE_NUM=4 / 8: k_rollout_geo<.., TRAIN[, PERS]>
rate = {12: 1100.0, 8: 1500.0, 4: 1800.0}[LN]
arr = synthetic_arrivals(n, rate=rate, horizon_s=200.0, seed=20250213, lane_num=LN)
from pve_mcc_amd.arrivals import synthetic_intentions
ch = synthetic_intentions(n, arr.shape[1], seed=20250213) if LN == 8 else None
env = pve_mcc_amd.BatchedIntersections(n, cap, arr, device="cuda:0", outputs=("obs_post", "obs_pre", "state_pre", "reward", "flags", "env_out"), obs_dtype=torch.float32,
                                       lane_num=LN, intentions=ch)
SRC = os.environ.get("SOURCE", "pool")
PERS = os.environ.get("PERSISTENT", "1") != "0"
if SRC == "actor":
    env.set_actor(bench.actor_weights())
env.reset(); env.set_action_pool(torch.as_tensor(bench.action_pool(n, cap, 99), device="cuda:0"))
ring = [env.alloc_trajectory(K) for _ in range(2)]
for rep in range(15):                                 # prefill to steady state (state_pre needs trajectory roll-outs)
    env.step_many(K, source=SRC, trajectory=ring[rep & 1], chunk=10, persistent=PERS)
ts = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step_many(K, source=SRC, trajectory=ring[rep & 1], chunk=10, persistent=PERS)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
print("trainer roll-out (%d lanes, %s, state_pre, f32 rows, %s): %.2f us per tick (median of 8), launch %s" % (LN, SRC, sys.argv[1] if len(sys.argv) > 1 else "", np.median(ts), env.last_launch()))
