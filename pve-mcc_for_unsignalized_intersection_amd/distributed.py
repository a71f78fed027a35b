"""Env-parallel sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm). Environments are fully independent, so the data
path has NO collective: each rank simulates its own shard; the only exchange is one all-gather of
the fixed-size metrics vector per reporting interval."""
import torch
import torch.distributed as dist

from . import _capi


def shard_range(n_total, rank, world):
    """Contiguous shard [lo, hi) of n_total envs for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def metrics_vector(metrics, device, extra=()):
    return torch.tensor([metrics[k] for k in _capi.METRIC_NAMES] + [float(x) for x in extra], dtype=torch.float64,
                        device=device)


def gather_metrics(metrics, device=None, extra=()):
    """All-gather each rank's metrics dict (12 x f64 = 96 B per rank) and return
    (per_rank [world, 12 + len(extra)] tensor on CPU, global dict of sums). `extra`: a few more per-rank numbers
    that ride in the same vector (e.g. the rank's own wall-clock), so that reporting stays ONE collective.
    Works without an initialised process group (world = 1)."""
    if not (dist.is_available() and dist.is_initialized()):
        v = metrics_vector(metrics, "cpu", extra)
        return v[None, :], dict(metrics)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    v = metrics_vector(metrics, device, extra)
    out = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(out, v)
    per_rank = torch.stack(out).cpu()
    tot = per_rank.sum(0)
    return per_rank, {k: float(tot[i]) for i, k in enumerate(_capi.METRIC_NAMES)}
