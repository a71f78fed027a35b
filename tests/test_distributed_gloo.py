"""Multi-rank path on CPU: world_size 2, gloo backend, kernels run by the test emulator.
Checks sharding arithmetic, that the single all-gather carries every rank's metrics vector, and
that the sharded job simulates exactly the same envs as one process (env-parallel => no data-path
collective, results independent of the partition)."""
import json
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from pve_mcc_amd.arrivals import synthetic_arrivals
from pve_mcc_amd.distributed import gather_metrics, shard_range
from pve_mcc_amd import _capi
from tests.hip_adapter import make_batch


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 4096, 32768, 10):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 3, 8) == (12288, 16384)      # BASELINE config 4: 4096 envs per GPU


def test_gather_metrics_without_process_group():
    m = {k: float(i) for i, k in enumerate(_capi.METRIC_NAMES)}
    per_rank, tot = gather_metrics(m)
    assert per_rank.shape == (1, 12) and tot == m


def test_two_rank_gloo_matches_single_process(tmp_path):
    from tests import dist_worker
    n_total, ticks = 6, 120
    out = str(tmp_path / "dist.json")
    mp.spawn(dist_worker.run, args=(2, free_port(), n_total, ticks, out), nprocs=2, join=True)
    res = json.load(open(out))
    assert res["shards"] == [[0, 3], [3, 6]]
    # single-process run over all envs with the same inputs
    arr = synthetic_arrivals(n_total, rate=500.0, horizon_s=ticks * 0.1 + 30, seed=3)
    b = make_batch(arr, n_total, 64, "emu", outputs=("obs_post", "reward", "flags", "env_out"))
    b.reset()
    g = torch.Generator().manual_seed(1234)
    acts_all = torch.rand(ticks, n_total, 64, generator=g, dtype=torch.float64) * 2 - 1
    for t in range(ticks):
        b.step(acts_all[t].contiguous())
    m = b.metrics()
    per_rank = np.array(res["per_rank"])
    assert per_rank.shape == (2, 12)
    for i, k in enumerate(_capi.METRIC_NAMES):
        if k in ("sum_reward", "sum_jerk"):
            assert abs(res["total"][k] - m[k]) <= 1e-9 * max(1.0, abs(m[k])), k
        else:
            assert res["total"][k] == m[k], (k, res["total"][k], m[k])
        assert abs(per_rank[:, i].sum() - res["total"][k]) <= 1e-9 * max(1.0, abs(m[k]))
    assert per_rank[0, _capi.METRIC_NAMES.index("ticks")] == 3 * ticks
