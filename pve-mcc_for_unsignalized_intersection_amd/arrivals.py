"""Arrival streams: the synthetic generator of BASELINE.md §3 and the loader of the reference's
MATLAB v5 `arvTimeNewVeh` files (main.py:387-389, data/test/*.mat)."""
import numpy as np


def synthetic_arrivals(n_envs, rate, horizon_s, seed=20250213, rows=None):
    """float64 [n_envs, rows, 12]: per env e and lane, cumulative sums of inter-arrival times
    max(1.0, Exponential(3600/rate)) from numpy.random.default_rng(seed + e), padded with +inf
    (the shipped streams have the same clipped-exponential shape)."""
    mean = 3600.0 / float(rate)
    if rows is None:
        rows = 64 + int(np.ceil(horizon_s / mean)) * 2
    out = np.full((n_envs, rows, 12), np.inf, dtype=np.float64)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + e)
        dt = np.maximum(1.0, rng.exponential(mean, size=(rows - 1, 12)))
        out[e, :rows - 1, :] = np.cumsum(dt, axis=0)
    return out


def pad_stream(arr, extra_rows=1):
    """Replace the zero padding at the tail of a shipped stream by +inf so that an
    exhausted lane simply stops spawning (the reference would spawn every tick, ref :379)."""
    a = np.array(arr, dtype=np.float64, copy=True)
    for l in range(a.shape[1]):
        col = a[:, l]
        nz = np.flatnonzero(col > 0)
        last = nz[-1] if len(nz) else -1
        col[last + 1:] = np.inf
    if extra_rows:
        a = np.concatenate([a, np.full((extra_rows, a.shape[1]), np.inf)], axis=0)
    return a
