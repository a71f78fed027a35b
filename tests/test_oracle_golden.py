"""Pins the CPU oracle (oracle/pve_oracle.c) against the golden vectors generated from the
unmodified reference (tests/golden/gen_golden.py). Runs on CPU, no reference needed."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv
from tests.parity_util import CASE_NAMES, GOLDEN_DIR, GoldenCase, replay_case


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_matches_golden(name):
    case = GoldenCase(name)
    env = OracleEnv(case.arrive, **case.ctor)
    n = replay_case(case, env)
    assert n == case.ticks
    assert env.ref_would_raise == 0


def test_oracle_constructor_warmup_pins():
    """SURVEY App. D first-tick pins (1000 stream): t_init, veh_num after ctor, tick-0 rewards."""
    case = GoldenCase("s1000_zero")
    env = OracleEnv(case.arrive)
    assert repr(env.current_time) == "1.0999999999999999"
    assert env.lane_counts().tolist() == [0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0]
    rec = env.tick(np.zeros(env.n_alive))
    assert np.allclose(rec["reward"], [-1.1473, -0.4973], atol=5e-5)


def test_oracle_geometry_known_answers():
    import os
    g = np.load(os.path.join(GOLDEN_DIR, "geometry.npz"))
    case = GoldenCase("s200_sin1")
    env = OracleEnv(case.arrive)
    ps = g["ps"]
    for lane in range(12):
        for k, p in enumerate(ps):
            q = env.get_p(p, lane)
            assert np.allclose(q, g["get_p"][lane, k], rtol=0, atol=1e-12), (lane, p)
    vd = g["vd"]
    for ego in range(12):
        for other in range(12):
            if np.all(np.isnan(vd[ego, other])):
                continue
            for k, p in enumerate(ps):
                got = env.get_virtual_distance(other, ego, p)
                exp = vd[ego, other, k]
                if np.isnan(exp):
                    assert got is None, (ego, other, p)
                else:
                    assert got == exp, (ego, other, p, got, exp)
    # SURVEY App. B known answers
    assert np.allclose(env.get_p(20, 0), [7.738093550595712, 0.92214541825631], atol=1e-12)
    assert np.allclose(env.get_p(2, 2), [13.258275389267348, 13.206568824747324], atol=1e-12)
    assert np.allclose(env.get_p(160, 3), [-2.5000255498095116, 147.51187456698503], atol=1e-12)
    assert np.allclose(env.get_p(-10, 4), [-7.4999956698723045, -25.0000012990379], atol=1e-12)
