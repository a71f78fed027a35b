"""CPU tests of the *parallel formulation* of the tick: the phase bodies of the HIP kernel
(csrc/pve_tick_core.h) are executed by the test emulator (tests/emu, same C ABI on host memory)
and compared with the sequential oracle and the golden vectors. The `-m gpu` twin of this file
(test_gpu_parity.py) runs the real kernels through libpveenv.so."""
import pytest
import torch

from tests.hip_adapter import SplitEnv, make_batch
from tests.parity_util import CASE_NAMES, GoldenCase, replay_case
from tests import scenarios

BACKEND = "emu"


@pytest.mark.parametrize("name", CASE_NAMES)
def test_split_protocol_matches_golden(name):
    case = GoldenCase(name)
    b = make_batch(case.arrive, 1, 128, BACKEND, **case.ctor)
    env = SplitEnv(b)
    replay_case(case, env, ftol=1e-9, dtol=1e-9, want_state=False)
    assert b.metrics()["overflow"] == 0


@pytest.mark.parametrize("name", ["s1000_sin1", "s200_sin1", "s1000_sin3"])
def test_split_protocol_matches_oracle_every_field(name):
    scenarios.check_split_vs_oracle(GoldenCase(name), BACKEND, ticks=400)


def test_capacity_64_on_sparse_stream():
    scenarios.check_split_vs_oracle(GoldenCase("s400_sin2"), BACKEND, ticks=600, capacity=64)


def test_fused_equals_split():
    scenarios.check_fused_equals_split(GoldenCase("s1000_sin3"), BACKEND, ticks=300)
    scenarios.check_fused_equals_split(GoldenCase("s200_sin1"), BACKEND, ticks=300, capacity=64)


def test_batch_of_independent_envs():
    scenarios.check_batch_independent(BACKEND, n_envs=5, capacity=64, ticks=150)


def test_overflow_defers_spawns():
    scenarios.check_overflow(BACKEND)


def test_empty_env_and_exhausted_stream():
    scenarios.check_empty_and_exhausted(BACKEND)


def test_reset_replays_the_same_episode():
    scenarios.check_reset_reproducible(BACKEND)


def test_float32_observation_rows():
    scenarios.check_obs_f32(BACKEND, lane_num=12)
    scenarios.check_obs_f32(BACKEND, lane_num=8, ticks=80)


def test_pipelined_sub_batches_equal_one_batch():
    scenarios.check_pipelined_equals_single(BACKEND, n_envs=7, n_sub=3, ticks=120)
    scenarios.check_pipelined_equals_single(BACKEND, n_envs=4, n_sub=2, ticks=80, actor=True)


@pytest.mark.parametrize("source", ["pool", "zero", "actor"])
def test_step_many_equals_single_ticks(source):
    scenarios.check_step_many(BACKEND, source, n_envs=3, chunks=(1, 7, 40, 3), trajectory_chunk=6)


def test_step_many_capacity_64_and_pipelined():
    scenarios.check_step_many(BACKEND, "pool", n_envs=3, capacity=64, rate=450.0, chunks=(5, 30), trajectory_chunk=4)
    scenarios.check_step_many_pipelined(BACKEND)


def test_emulated_driver_launch_shape_vs_oracle():
    """Small-batch CPU version of the driver-shape test (the GPU test runs it at 4096 x 128)."""
    for traj in (False, True):
        m, _ = scenarios.check_driver_shape_vs_oracle(BACKEND, n_envs=7, n_sub=2, n_sample=7, calls=(30, 30, 5, 20), trajectory=traj)
        assert m["ctl_steps"] > 0


def test_emulated_closed_loop_rollout_vs_two_launch():
    """Small-batch CPU version of the full-size config-5 certification (the GPU test runs it at 4096 x 128)."""
    m, worst = scenarios.check_closed_loop_rollout_vs_two_launch(BACKEND, n_envs=5, n_sub=2, n_sample=3, chunk=5,
                                                                 calls=(30, 30, 5, 25), obs_dtype=torch.float32)
    assert m["ctl_steps"] > 0 and worst <= 5e-4


def test_emulated_step_many_emits_training_states():
    """state_pre / obs_pre / 7-action vectors over pve_step_many trajectories, every tick vs the oracle (f64 and f32 rows)."""
    scenarios.check_step_many_state_rows(BACKEND, n_envs=2, calls=(30, 12, 25), chunk=0)
    scenarios.check_step_many_state_rows(BACKEND, n_envs=2, calls=(20, 15), chunk=7, obs_dtype=torch.float32, seed=83)


def test_emulated_prepared_step_many_equals_step_many():
    """prepare_step_many (one ctypes call per re-issue) == step_many, incl. the position in the action pool and pipelined
    sub-batches."""
    import numpy as np
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from pve_mcc_amd.batched import PipelinedIntersections
    from tests.hip_adapter import emulator_lib
    arr = synthetic_arrivals(5, rate=900.0, horizon_s=40.0, seed=4)
    outs = ("obs_post", "reward", "flags", "env_out", "new_slot", "nbr")
    a = make_batch(arr, 5, 128, BACKEND, outputs=outs)
    b = PipelinedIntersections(5, 128, arr, n_sub=2, outputs=outs, device="cpu", _lib=emulator_lib())
    pool = torch.as_tensor(np.random.default_rng(1).uniform(-2, 2, size=(5, 5, 128)))
    a.reset(); b.reset()
    a.set_action_pool(pool); b.set_action_pool(pool)
    f = b.prepare_step_many(7, chunk=3)
    for _ in range(6):
        a.step_many(7, chunk=3)
        f()
    for k in scenarios.STATE_F + scenarios.STATE_I:
        x = a.state_field(k).numpy()
        y = np.concatenate([sub.state_field(k).numpy() for sub in b.subs], 0)
        live = a.state_field("meta").numpy() != 0
        assert np.array_equal(x[live], y[live]), k
    assert a.ticks == 42 and all(sub.ticks == 42 for sub in b.subs)
    # a prepared call holds the address of the pool it was built on: replacing the pool must make it refuse, not read
    # recycled memory; a call prepared on another source kind is unaffected
    import pytest as _pytest
    from pve_mcc_amd._capi import PveError
    g = a.prepare_step_many(3)
    z = a.prepare_step_many(2, source="zero")
    assert g._keep[2] is a._pool
    a.set_action_pool(pool.clone())
    with _pytest.raises(PveError, match="stale"):
        g()
    z()
    a.prepare_step_many(3)()
    assert a.ticks == 47


def test_emulated_step_many_table_source_equals_single_ticks():
    """PVE_SRC_TABLE (actions by (tick, vehicle id), gathered by the vehicle's own thread inside the resident loop; the
    spawned vehicles' first actions; still ticks) == single ticks with the same table applied on the host side."""
    scenarios.check_step_many(BACKEND, "table", n_envs=4, chunks=(1, 7, 40, 3, 60), trajectory_chunk=12)
    scenarios.check_step_many(BACKEND, "table", n_envs=3, capacity=64, rate=350.0, chunks=(5, 30, 50), trajectory_chunk=8, seed=7)


def test_emulated_step_many_symmetric_lanes_equal_distances():
    """Runs of equal virtual distances (lanes that spawn in the same tick, zero actions) through the resident loop."""
    arr = scenarios.symmetric_arrivals(2, gap_s=3.4, rows=40, lane_groups=[[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]])
    scenarios.check_step_many(BACKEND, "zero", n_envs=2, chunks=(1, 30, 90), trajectory_chunk=10, arrivals=arr)


@pytest.mark.parametrize("source,cap,rate", [("pool", 128, 1100.0), ("zero", 128, 1100.0), ("table", 128, 1100.0), ("pool", 64, 420.0)])
def test_work_queue_item_schedule_emulated(source, cap, rate):
    """pve_rollout.persistent through the emulator: the item schedule pve_step_many lays out (full items, taper, 3-tick tail;
    k_base / pool row / table row / trajectory block per item) run sequentially, item by item, with the kernel's own
    rollout_item() -- == single ticks bit for bit, and the persistent path IS the one taken (pve_debug_last_launch)."""
    scenarios.check_step_many(BACKEND, source, n_envs=3, capacity=cap, rate=rate, chunks=(1, 7, 40, 20, 9, 33), trajectory_chunk=12,
                              persistent=True, seed=29)


@pytest.mark.parametrize("source,dtype", [("pool", torch.float64), ("table", torch.float32), ("zero", torch.float64)])
def test_trainer_rollout_through_the_work_queue_emulated(source, dtype):
    """Round 5: the training outputs (obs_pre, state_pre with fresh / stale neighbour rows, 7-action vectors) of persistent
    trajectory roll-outs -- items of the emulated work queue in sequence, the stale rows of an item's first tick taken from
    the previous item's last block -- against the oracle at every tick; PVE_SRC_TABLE with the training outputs."""
    scenarios.check_step_many_state_rows(BACKEND, n_envs=2, calls=(30, 12, 25), chunk=7, obs_dtype=dtype, source=source, persistent=True,
                                         min_ctl_per_tick=(1 if source == "zero" else 5))


def test_closed_loop_training_rollout_emulated():
    scenarios.check_closed_loop_state_rows(BACKEND, n_envs=2, calls=(12, 9), chunk=5)


def test_id_tape_driver_shape_small_emulated():
    """The table-source form of the driver-shape scenario through the emulator (small batch): the scenario's oracle side of the
    id-indexed tape, the emulated work queue's item schedule."""
    scenarios.check_driver_shape_vs_oracle(BACKEND, n_envs=6, n_sub=1, chunk=12, n_sample=6, calls=(30, 20, 5, 20), persistent=True, table=True,
                                           rate=900.0)
    scenarios.check_driver_shape_vs_oracle(BACKEND, n_envs=4, n_sub=1, chunk=7, n_sample=4, calls=(25, 20), persistent=True, table=True,
                                           trajectory=True, rate=900.0)
