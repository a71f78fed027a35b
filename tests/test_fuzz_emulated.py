"""Random-tape fuzzing of the parallel tick (emulated kernels) against the sequential oracle."""
import pytest

from tests import scenarios


@pytest.mark.parametrize("seed,rate,cap,quant", [(1, 900.0, 128, None), (2, 500.0, 64, None), (3, 1100.0, 128, 1.0),
                                                 (4, 700.0, 128, 3.0)])
def test_fuzz_random_tapes(seed, rate, cap, quant):
    coll, lock = scenarios.check_fuzz_vs_oracle("emu", n_envs=3, capacity=cap, ticks=700, rate=rate, seed=seed,
                                                quantize=quant)
    assert coll > 0 and lock > 0, "the fuzz tapes are meant to provoke collisions and dead-locks"


def test_fuzz_more_than_64_controlled_vehicles():
    """Dense traffic: the controlled vehicles of an intersection no longer fit the first wave, so the second wave takes
    part in the dense-mapped phases (BUILD / WALK / REWARD / dead-lock walk / observation rows)."""
    scenarios.check_fuzz_vs_oracle("emu", n_envs=3, capacity=128, ticks=420, rate=1400.0, seed=5, action_scale=0.3)
    assert scenarios.check_fuzz_vs_oracle.max_ctl > 64


@pytest.mark.parametrize("groups,scale,quant", [([list(range(12))], 0.0, None),
                                                ([[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]], 3.0, 3.0)])
def test_fuzz_symmetric_lanes_equal_distances(groups, scale, quant):
    """Lanes that spawn in the same tick: runs of equal virtual distances in every list, every tick (RANK's claim / fix-up
    of equal keys, WALK's exact path), with symmetric collisions inside the box."""
    arr = scenarios.symmetric_arrivals(2, gap_s=3.4, rows=40, lane_groups=groups)
    scenarios.check_fuzz_vs_oracle("emu", n_envs=2, capacity=128, ticks=320, rate=0.0, seed=9, action_scale=scale,
                                   quantize=quant, arrivals=arr)
