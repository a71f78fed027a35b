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


def check_line(out, n):
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
    assert abs(d["value"] - 64 * 12 * 6 * n / (d["ms_per_step"] * 6 / 1e3)) / d["value"] < 1e-6
    # roofline.achieved follows from the line's own wall clock: B_alg x slots of ONE GPU / time per tick
    assert abs(r["achieved"] - 380.0 * 64 * 12 / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert d["population"] in ("steady", "cold") and "mean_alive_per_env" in d and "prefill_ticks" in d
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
