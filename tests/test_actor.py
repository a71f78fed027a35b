"""f1: MADDPG actor inference. CPU part: the NumPy restatement (oracle/actor_np.py) is pinned by the
closed-loop known answers of SURVEY App. D; the C-ABI plumbing is exercised through the emulator."""
import numpy as np

from oracle.actor_np import actor_forward, flat_weights, load_weights
from oracle.oracle import OracleEnv
from tests import actor_scenarios as A


def test_weight_fixture_shapes():
    w = load_weights()
    assert {k: v.shape for k, v in w.items()} == {
        "ln0_beta": (28,), "ln0_gamma": (28,), "w1": (28, 64), "b1": (64,), "ln1_beta": (64,), "ln1_gamma": (64,),
        "w2": (64, 64), "b2": (64,), "ln2_beta": (64,), "ln2_gamma": (64,), "w3": (64, 1), "b3": (1,)}
    assert flat_weights(w).shape == (6393,) and flat_weights(w).dtype == np.float32


def test_numpy_actor_closed_loop_known_answers():
    """SURVEY App. D, last row: 1000 stream, 1000 ticks, pretrained actor (main.py:test() protocol)."""
    w = load_weights()
    env = OracleEnv(A.stream_1000())
    alive = ctl = coll = locks = 0
    rew = []
    for t in range(1000):
        vid, c, obs0 = env.alive_view()
        a = np.where(c != 0, actor_forward(w, obs0).astype(np.float64), 0.0)
        alive += len(vid)
        ctl += int(c.sum())
        rec = env.tick(a)
        coll += int((rec["coll_pv"] > 0).sum())
        locks += rec["lock"]
        rew += list(rec["reward"])
    assert (alive, ctl, rec["id_seq"], rec["passed"], coll, locks) == (72416, 37295, 323, 281, 0, 548)
    assert abs(rec["passed_step_total"] / (rec["passed"] + 1e-4) * 0.1 - 12.294) < 1e-3
    assert abs(np.mean(rew) - 1.30294) < 1e-4
    assert np.all(np.abs(actor_forward(w, np.zeros((3, 28)))) <= 3.0)


def test_actor_through_c_abi_emulated():
    assert A.check_actions_on_oracle_states("emu", ticks=120) <= A.ACTION_TOL


def test_closed_loop_through_c_abi_emulated():
    A.check_closed_loop_on_device("emu", ticks=1000)


def test_evaluate_protocol_known_answers(monkeypatch):
    """pve_mcc_amd.evaluate = main.py:test() on the device; through the emulator it reproduces SURVEY App. D's
    pretrained-actor row (323 vehicles, 281 passed, 0 collisions, 548 locks, pT-m 12.294 s, 208.799 jerk/veh)."""
    import os
    from pve_mcc_amd import evaluate as E
    from pve_mcc_amd.arrivals import pad_stream
    from tests.hip_adapter import emulator_lib
    orig = E.BatchedIntersections
    monkeypatch.setattr(E, "BatchedIntersections",
                        lambda *a, **k: orig(*a, **{**k, "device": "cpu", "_lib": emulator_lib()}))
    res = E.evaluate(pad_stream(A.stream_1000()), load_weights(), ticks=1000, device="cpu")
    assert (res["vehicles"], res["passed"], res["collisions"], res["lock_num"]) == (323, 281, 0, 548)
    assert abs(res["pT_m"] - 12.294) < 1e-3 and abs(res["jerk_mean"] - 208.799) < 1e-2
    assert abs(res["reward_mean"] - 1.30294) < 1e-4


def test_closed_loop_on_float32_observations_is_bit_identical():
    A.check_closed_loop_f32_obs_equals_f64("emu", ticks=150, n_envs=2)
