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


def test_actor_tape_from_the_live_reference_reproduces_appendix_d():
    """the pretrained actor (the reference's own graph) driving the LIVE reference env for 1000 ticks
    on the 1000 stream (main.py:397-441; fixture s1000_actor.npz from tests/golden/gen_golden.py).  The aggregates main.py:test()
    prints (main.py:523-526) are asserted from the fixture AND from the oracle's replay of the recorded action tape."""
    case = GoldenCase("s1000_actor")
    agg = case.aggregates
    assert (agg["id_seq"], agg["passed"], agg["collided"], agg["locks"], agg["ctl_steps"]) == (323, 281, 0, 548, 37295)
    assert abs(agg["pT_m"] - 12.294) < 5e-4 and abs(agg["reward_mean"] - 1.30294) < 5e-6 and abs(agg["jerk_per_veh"] - 208.799) < 5e-4
    env = OracleEnv(case.arrive)
    ctl = coll = locks = 0
    rew = []
    for t in range(case.ticks):
        vid, c, obs0 = env.alive_view()
        rec = env.tick(case.policy(t, vid, c, obs0))
        ctl += len(rec["ids"]); coll += int((rec["coll_pv"] > 0).sum()); locks += rec["lock"]; rew += list(rec["reward"])
    assert (rec["id_seq"], rec["passed"], coll, locks, ctl) == (323, 281, 0, 548, 37295)
    assert abs(rec["passed_step_total"] / (rec["passed"] + 1e-4) * 0.1 - agg["pT_m"]) < 1e-12
    assert abs(np.mean(rew) - agg["reward_mean"]) < 1e-9
    # the tape IS what the committed weights produce: the NumPy restatement of the actor on the oracle's rows gives the same
    # float32 actions to 2 ulp (tests/test_actor_graph.py pins that restatement against the decoded graph)
    from oracle.actor_np import actor_forward, load_weights
    w = load_weights()
    env = OracleEnv(case.arrive)
    for t in range(60):
        vid, c, obs0 = env.alive_view()
        a = case.policy(t, vid, c, obs0)
        if c.any():
            mine = actor_forward(w, obs0[c != 0].astype(np.float32)).astype(np.float64).reshape(-1)
            assert np.allclose(mine, a[c != 0], rtol=0, atol=2e-6), t
        env.tick(a)
