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
