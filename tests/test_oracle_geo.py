"""Pins the general-geometry CPU oracle (oracle/pve_oracle_geo.c, lane_num 4 / 8 / 12) against golden vectors
generated from the unmodified reference (tests/golden/gen_golden_geo.py).  Runs on CPU, no reference needed."""
import os

import numpy as np
import pytest

from oracle.oracle_geo import OracleGeoEnv
from tests.parity_util import CASE_NAMES, GEO_CASE_NAMES, GOLDEN_DIR, GoldenCase, replay_case


@pytest.mark.parametrize("name", GEO_CASE_NAMES)
def test_geo_oracle_matches_golden(name):
    case = GoldenCase(name)
    env = OracleGeoEnv(case.arrive, case.lane_num, choice=case.choice, **case.ctor)
    assert replay_case(case, env) == case.ticks
    assert env.ref_would_raise == 0


@pytest.mark.parametrize("name", ["s1000_sin1", "s400_sin2", "s1000_sin3"])
def test_geo_oracle_12_lanes_equals_the_pinned_12_lane_vectors(name):
    """The generalised loop structure (directions per physical lane, (lane, intention, j) order) run with
    lane_num = 12 must reproduce the 12-lane golden tapes that pin oracle/pve_oracle.c."""
    assert name in CASE_NAMES
    case = GoldenCase(name)
    env = OracleGeoEnv(case.arrive, 12, **case.ctor)
    assert replay_case(case, env) == case.ticks


@pytest.mark.parametrize("lane_num", [4, 8])
def test_geo_oracle_geometry_known_answers(lane_num):
    g = np.load(os.path.join(GOLDEN_DIR, "geometry_geo.npz"))
    ps, gp, vd = g["ps%d" % lane_num], g["get_p%d" % lane_num], g["vd%d" % lane_num]
    arr = np.cumsum(np.full((4, lane_num), 50.0), axis=0)
    env = OracleGeoEnv(arr, lane_num)
    n_xy = 0
    for lane in range(lane_num):
        for m in range(3):
            if np.all(np.isnan(gp[lane, m])):
                continue
            for k, p in enumerate(ps):
                q = env.get_p(p, lane, m)
                assert np.allclose(q, gp[lane, m, k], rtol=0, atol=1e-12), (lane, m, p, q, gp[lane, m, k])
                n_xy += 1
    assert n_xy == (12 if lane_num == 4 else 16) * len(ps)
    nd = vd.shape[0]
    n_vd = 0
    for ego in range(nd):
        for other in range(nd):
            if np.all(np.isnan(vd[ego, other])):
                continue
            for k, p in enumerate(ps):
                got = env.get_virtual_distance(other, ego, p)
                exp = vd[ego, other, k]
                if np.isnan(exp):
                    assert got is None, (ego, other, p)
                else:
                    assert got == exp, (ego, other, p, got, exp)      # table arithmetic: bit-exact
                    n_vd += 1
    assert n_vd > 1000
