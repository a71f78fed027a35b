"""The RCCL branch of the multi-GPU plumbing on real hardware: backend "nccl" (= RCCL on ROCm) with world_size 1 --
process-group creation on the device, the metrics all-gather of distributed.gather_metrics and the MAX all-reduce of the
bench timing execute on the MI355X (the world_size-2 logic is covered on CPU with gloo, tests/test_distributed_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import pve_mcc_amd
from pve_mcc_amd.arrivals import synthetic_arrivals
from pve_mcc_amd.distributed import gather_metrics
arr = synthetic_arrivals(8, rate=1100.0, horizon_s=20.0, seed=3)
b = pve_mcc_amd.BatchedIntersections(8, 128, arr, device="cuda:0", outputs=("flags", "env_out"))
b.reset()
for _ in range(50):
    b.step(None)
m = b.metrics()
per_rank, tot = gather_metrics(m, torch.device("cuda", 0))
tw = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")
dist.all_reduce(tw, op=dist.ReduceOp.MAX)
dist.barrier()
print(json.dumps(dict(backend=dist.get_backend(), shape=list(per_rank.shape), tot=tot, m=m, tw=float(tw.item()))))
dist.destroy_process_group()
"""


def test_nccl_world_size_1_gathers_metrics_on_the_device():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.check_output([sys.executable, "-c", WORKER % ROOT], env=env, text=True, timeout=600)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["shape"] == [1, 12] and d["tw"] == 1.25
    assert d["tot"] == d["m"] and d["m"]["ticks"] == 400 and d["m"]["alive_steps"] > 0


def test_bench_gpus_1_through_the_self_launcher_path():
    """`bench.py --gpus 1` under an outer torchrun-style environment (WORLD_SIZE=1): the nccl process group is not created
    for one rank, the line carries the steady-state protocol fields."""
    out = subprocess.check_output([sys.executable, "bench.py", "--envs", "256", "--steps", "20", "--warmup", "5",
                                   "--no-cpu-baseline", "--no-copy-peak"], cwd=ROOT, text=True, timeout=900)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d["population"] == "steady" and d["prefill_ticks"] >= 300 and d["mean_alive_per_env"] > 60
    r = d["roofline"]
    per_tick = 128 * 256 / (d["ms_per_step"] * 1e-3) / 1e9
    assert abs(r["nominal"]["achieved"] - 380.0 * per_tick) / r["nominal"]["achieved"] < 1e-9
    # pve_step_many launches of T ticks move the persistent state (124 of the 380 B) once per launch, not once per tick
    T = d["config"]["ticks_per_state_move"]            # (= ticks per launch; per queue item of a persistent launch)
    assert d["config"]["mode"] == "rollout" and abs(r["alg_bytes_per_slot_step"] - (380.0 - 124.0 * (1 - 1.0 / T))) < 1e-9
    assert abs(r["achieved"] - r["alg_bytes_per_slot_step"] * per_tick) / r["achieved"] < 1e-9
    assert d["verified"] is True and d["ranks"]["seen"] == 1
