"""bench.py contract on CPU: one JSON line with the required keys, single process and world_size 2 (gloo) through
tests/bench_emulated_launcher.py (bench.main with the kernel emulator injected; the multi-rank plumbing, barrier/max-over-ranks timing and the metrics
all-gather are the real code paths)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def check_line(out, n, envs=12):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == 6 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if n == 1:                                     # reported on rank 0 at N = 1 only
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    else:
        assert "cpu_baseline" not in d
    assert abs(d["value"] - 64 * envs * 6 * n / (d["ms_per_step"] * 6 / 1e3)) / d["value"] < 1e-6
    # roofline.achieved follows from the line's own wall clock: B_alg x slots of ONE GPU / time per tick
    assert abs(r["achieved"] - 380.0 * 64 * envs / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert d["population"] in ("steady", "cold") and "mean_alive_per_env" in d and "prefill_ticks" in d
    # the line certifies itself: sampled envs of the timed run replayed by the oracle
    assert d["verified"] is True and d["verification"]["ticks_replayed"] == 9 and len(d["verification"]["envs"]) == min(8, envs)
    # ... and shows that n ranks took part, each with its own envs (item 2d / 7)
    rk = d["ranks"]
    assert rk["seen"] == n and len(rk["ms"]) == n and rk["ticks"] == [6.0] * n
    assert rk["first_env"] == [envs * k for k in range(n)] and rk["envs_per_rank"] == envs      # disjoint env ranges = disjoint seeds
    # the job's time IS the slowest rank's own time for its K ticks (taken before the closing barrier; MAX over ranks)
    assert all(ms > 0 for ms in rk["ms"]) and abs(max(rk["ms"]) - d["ms_per_step"] * 6) <= 1e-6 * max(rk["ms"])
    assert r["nominal"]["alg_bytes_per_slot_step"] == 380.0 and "binding" in r
    # multi-rank runs carry BASELINE config 4 beside the weak-scaled headline: 64-slot intersections, rank k owns the global
    # envs shard_range(envs x n, k, n), its own barrier-bracketed timed region, MAX over ranks, one all-gather, verified
    c4 = d.get("config4")
    if n == 1:
        assert c4 is None
    else:
        assert c4["n_gpus"] == n and c4["steps"] == 6 and c4["verified"] is True and c4["overflow"] == 0
        assert c4["ranks"]["seen"] == n and c4["ranks"]["first_env"] == [envs * k for k in range(n)]
        assert abs(c4["value"] - 64 * envs * n * 6 / (c4["ms_per_step"] * 6 / 1e3)) / c4["value"] < 1e-9
        assert max(c4["ranks"]["ms"]) <= c4["ms_per_step"] * 6 * (1 + 1e-9) and "32452843" in c4["ranks"]["arrival_seeds"]
    return d


def test_bench_single_process_emulated():
    out = subprocess.check_output([sys.executable, "tests/bench_emulated_launcher.py", "--envs", "12", "--capacity", "64",
                                   "--steps", "6", "--warmup", "3", "--prefill", "0"], cwd=ROOT, text=True, timeout=600)
    d = check_line(out, 1)
    assert d["population"] == "cold" and d["prefill_ticks"] == 0
    c = d["cpu_baseline"]
    assert c["single_thread"]["value"] > 0 and c["cpu_model"] and "300 un-timed warm-up" in c["sample"]


def test_bench_prefill_is_untimed_and_reported():
    """--prefill runs before (and is not part of) --warmup; a short prefill marks the line cold."""
    out = subprocess.check_output([sys.executable, "tests/bench_emulated_launcher.py", "--envs", "4", "--capacity", "64",
                                   "--steps", "6", "--warmup", "3", "--prefill", "20", "--no-cpu-baseline"],
                                  cwd=ROOT, text=True, timeout=600)
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    d = json.loads(lines[0])
    assert d["prefill_ticks"] == 20 and d["population"] == "cold" and d["warmup"] == 3 and d["steps"] == 6


def test_bench_two_ranks_gloo_emulated():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "tests/bench_emulated_launcher.py", "--gpus", "2",
           "--envs", "12", "--capacity", "64", "--steps", "6", "--warmup", "3", "--prefill", "0"]
    out = subprocess.check_output(cmd, cwd=ROOT, text=True, timeout=900, stderr=subprocess.DEVNULL)
    d = check_line(out, 2)
    assert d["config"]["parallelism"] == "env-parallel x2"


def test_bench_self_launches_two_ranks():
    """`bench.py --gpus 2` with no outer launcher starts its own ranks (the driver calls it that way) and forwards
    rank 0's line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "tests/bench_emulated_launcher.py", "--gpus", "2", "--envs", "12", "--capacity", "64",
                        "--steps", "6", "--warmup", "3", "--prefill", "0"], cwd=ROOT, text=True, timeout=900, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert p.returncode == 0
    d = check_line(p.stdout, 2)
    assert d["config"]["parallelism"] == "env-parallel x2"


def test_bench_self_launches_eight_ranks_config_4_sharding():
    """BASELINE config 4's shape on CPU: `bench.py --gpus 8` starts eight ranks itself, every rank owns its own envs
    (weak scaling: rank k simulates global envs [k E, (k + 1) E), arrival seeds 20250213 + global env index), the single
    all-gather carries eight metric vectors, the timing is the MAX over ranks and every rank's sampled envs verify.  With
    E = 4096 that partition is exactly shard_range(32768, k, 8) -- 32 768 intersections x 64 slots over 8 GPUs."""
    from pve_mcc_amd.distributed import shard_range
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, "tests/bench_emulated_launcher.py", "--gpus", "8", "--envs", "5", "--capacity", "64",
                        "--steps", "6", "--warmup", "3", "--prefill", "0"], cwd=ROOT, text=True, timeout=1200, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert p.returncode == 0
    d = check_line(p.stdout, 8, envs=5)
    assert d["config"]["parallelism"] == "env-parallel x8" and d["ranks"]["seen"] == 8
    assert d["ranks"]["arrival_seeds"] == "20250213 + first_env + e"
    # the same rule at config 4's size: rank k's first env = k x 4096 = the start of shard k of 32 768
    assert [shard_range(32768, k, 8) for k in range(8)] == [(4096 * k, 4096 * (k + 1)) for k in range(8)]
    assert d["value"] > 0 and d["overflow"] == 0


def test_bench_verification_detects_a_wrong_tape():
    """verify_against_oracle is a real check: the same envs replayed with another action pool do not verify."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import pve_mcc_amd
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from tests.hip_adapter import emulator_lib
    n_envs, cap, ticks = 4, 64, 80
    arr = synthetic_arrivals(n_envs, rate=500.0, horizon_s=40.0, seed=5)
    pool_np = bench.action_pool(n_envs, cap, seed=3)
    b = pve_mcc_amd.BatchedIntersections(n_envs, cap, arr, device="cpu", outputs=("obs_post", "reward", "flags", "env_out"),
                                         _lib=emulator_lib())
    b.reset()
    pool = torch.as_tensor(pool_np)
    for t in range(ticks):
        b.step(pool[t % bench.N_POOL])
    locate = lambda e: (b, e)
    last = lambda e: {n: b.out[n][e].numpy() for n in ("flags", "reward", "env_out")}
    good = bench.verify_against_oracle(locate, last, arr, pool_np, ticks, 12, None, n_sample=4)
    assert good["verified"] is True and good["envs"] == [0, 1, 2, 3]
    bad = bench.verify_against_oracle(locate, last, arr, -pool_np, ticks, 12, None, n_sample=4)
    assert bad["verified"] is False and "env 0" in bad["mismatch"]
    off = bench.verify_against_oracle(locate, last, arr, pool_np, ticks - 1, 12, None, n_sample=4)
    assert off["verified"] is False
