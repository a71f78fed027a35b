"""Worker of the world_size-2 gloo test: each rank simulates its shard of a global set of envs
(env-parallel, no data-path collective) and the ranks all-gather the metrics vector once."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def run(rank, world, port, n_total, ticks, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from pve_mcc_amd.distributed import gather_metrics, shard_range
    from tests.hip_adapter import make_batch
    arr = synthetic_arrivals(n_total, rate=500.0, horizon_s=ticks * 0.1 + 30, seed=3)
    lo, hi = shard_range(n_total, rank, world)
    b = make_batch(arr[lo:hi], hi - lo, 64, "emu", outputs=("obs_post", "reward", "flags", "env_out"))
    b.reset()
    g = torch.Generator().manual_seed(1234)
    acts_all = torch.rand(ticks, n_total, 64, generator=g, dtype=torch.float64) * 2 - 1
    for t in range(ticks):
        b.step(acts_all[t, lo:hi].contiguous())
    per_rank, tot = gather_metrics(b.metrics())
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"per_rank": per_rank.tolist(), "total": tot, "shards": [shard_range(n_total, r, world)
                                                                                 for r in range(world)]}, f)
    dist.destroy_process_group()
