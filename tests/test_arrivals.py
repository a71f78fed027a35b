"""Arrival-stream I/O: the MAT-v5 reader against SciPy on the reference's own data files
(committed as data fixtures under tests/golden/streams), and the synthetic generator of BASELINE.md §3."""
import os

import numpy as np
import pytest

from pve_mcc_amd.arrivals import load_arrival_mat, pad_stream, synthetic_arrivals
from tests.parity_util import GOLDEN_DIR, GoldenCase

STREAMS = os.path.join(GOLDEN_DIR, "streams")


@pytest.mark.parametrize("name", ["arvTimeNewVeh_new_200_12.mat", "arvTimeNewVeh_new_1000_12.mat"])
def test_mat_reader_matches_scipy_bit_for_bit(name):
    scio = pytest.importorskip("scipy.io")
    path = os.path.join(STREAMS, name)
    mine = load_arrival_mat(path)
    ref = scio.loadmat(path)["arvTimeNewVeh"]
    assert mine.dtype == np.float64 and mine.shape == ref.shape and mine.flags["C_CONTIGUOUS"]
    assert np.array_equal(mine, ref)


def test_mat_reader_feeds_the_golden_runs():
    """The golden tapes were generated from these very files: their stored arrival rows are a prefix."""
    a200 = load_arrival_mat(os.path.join(STREAMS, "arvTimeNewVeh_new_200_12.mat"))
    a1000 = load_arrival_mat(os.path.join(STREAMS, "arvTimeNewVeh_new_1000_12.mat"))
    assert a200.shape == (322, 12) and a1000.shape == (1400, 12)
    g = GoldenCase("s200_sin1").arrive
    assert np.array_equal(g, a200[:g.shape[0]])
    g = GoldenCase("s1000_zero").arrive
    assert np.array_equal(g, a1000[:g.shape[0]])


def test_mat_reader_errors(tmp_path):
    p = tmp_path / "x.mat"
    p.write_bytes(b"not a mat file")
    with pytest.raises(ValueError):
        load_arrival_mat(str(p))
    with pytest.raises(KeyError):
        load_arrival_mat(os.path.join(STREAMS, "arvTimeNewVeh_new_200_12.mat"), name="nope")


def test_pad_stream_replaces_zero_tail():
    a = load_arrival_mat(os.path.join(STREAMS, "arvTimeNewVeh_new_1000_12.mat"))
    p = pad_stream(a)
    assert p.shape == (1401, 12) and np.isinf(p[-1]).all()
    for l in range(12):
        col = a[:, l]
        k = np.flatnonzero(col > 0)[-1]
        assert np.array_equal(p[:k + 1, l], col[:k + 1]) and np.isinf(p[k + 1:, l]).all()
        assert np.all(np.diff(col[:k + 1]) > 0)


def test_synthetic_arrivals_definition():
    a = synthetic_arrivals(3, rate=1100.0, horizon_s=130.0, seed=20250213)
    assert a.shape == (3, 64 + 40 * 2, 12) and np.isinf(a[:, -1]).all()
    rng = np.random.default_rng(20250213 + 1)
    dt = np.maximum(1.0, rng.exponential(3600.0 / 1100.0, size=(a.shape[1] - 1, 12)))
    assert np.array_equal(a[1, :-1], np.cumsum(dt, axis=0))
    d = np.diff(a[0, :-1], axis=0)
    assert d.min() >= 1.0 - 1e-9 and 3.0 < d.mean() < 4.2   # clipped exponential (diff of cumsum rounds), mean ~3.3-3.8 s
    assert np.array_equal(a, synthetic_arrivals(3, rate=1100.0, horizon_s=130.0, seed=20250213))
