"""CPU parity of the HOME block of k_rollout<128, 5, ..> (csrc/pve_tick_core.h: Shared<128, false, true>): the 96-register /
10-workgroups-per-CU build of the resident kernel keeps the carried per-slot fields (jerk_sum, closer_p, id, seq | vnum,
count, vir_dis, p / v / a, the next action) in LDS instead of registers and works the virtual-lane lists in passes over an
entry pool of 304 entries.  The emulator (tests/emu) runs the same phase bodies with that block when PVE_EMU_HOME is set
(1 = the kernel's pool of 304 entries, 2 = 296) -- every roll-out must still
equal single ticks of k_tick bit for bit (ref traffic_interaction_scene.py:1501-1539, :222-376, :435-444)."""
import ctypes as C

import pytest

from tests import scenarios
from tests.hip_adapter import emulator_lib

BACKEND = "emu"


def _passes():
    lib = emulator_lib()
    lib.pve_emu_max_passes.restype = C.c_int
    return int(lib.pve_emu_max_passes())


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("source", ["table", "pool", "zero"])
def test_home_block_step_many_equals_single_ticks(monkeypatch, mode, source):
    monkeypatch.setenv("PVE_EMU_HOME", str(mode))
    _passes()
    # (mode 2: > 296 list entries need ~85 controlled vehicles: slow traffic that does not collide fills the intersection)
    kw = dict(prefill=200, rate=1600.0, cfg=dict(vm=2.0, collision_thr=0.01)) if mode == 2 else {}
    scenarios.check_step_many(BACKEND, source, n_envs=3, chunks=(1, 7, 40, 3, 60), trajectory_chunk=12, seed=171 + mode, **kw)
    p = _passes()
    assert p >= 1, "the HOME block did not run"
    if mode == 2:
        assert p >= 2, "a pool of 296 entries must take several passes with a full intersection"


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("source", ["table", "pool"])
def test_home_block_through_the_work_queue(monkeypatch, mode, source):
    """the persistent form's item schedule (items of different lengths: the `last tick` of every item stages the jerks)"""
    monkeypatch.setenv("PVE_EMU_HOME", str(mode))
    _passes()
    kw = dict(prefill=200, rate=1600.0, cfg=dict(vm=2.0, collision_thr=0.01)) if mode == 2 else {}
    scenarios.check_step_many(BACKEND, source, n_envs=4, chunks=(20, 9, 33, 50), trajectory_chunk=14, seed=181, persistent=True, **kw)
    assert _passes() >= mode


def test_home_block_full_intersection_takes_several_passes_of_the_kernel_pool(monkeypatch):
    """3000 veh/h/lane, vm = 3 m/s, full braking and (almost) no collisions: the 128 slots fill up with controlled vehicles
    and the lists need more than 304 entries -- the multi-pass form with the pool size the HIP kernel has (and deferred
    spawns / a full intersection on the table source's late spawn-action gather)."""
    monkeypatch.setenv("PVE_EMU_HOME", "1")
    _passes()
    st = scenarios.check_step_many(BACKEND, "table", n_envs=2, chunks=(250, 40, 7, 60), trajectory_chunk=10, seed=191, rate=3000.0,
                                   cfg=dict(vm=3.0, collision_thr=0.01), act_lo=-3.0, act_hi=-2.0)
    assert st["max_alive"] >= 120 and st["overflow"] > 0, st
    assert _passes() >= 2, "a full intersection must overflow the 304-entry pool"


def test_home_block_symmetric_lanes_equal_distances(monkeypatch):
    """runs of equal virtual distances (the claim / fix-up of RANK) inside passes"""
    for mode in ("1", "2"):
        monkeypatch.setenv("PVE_EMU_HOME", mode)
        arr = scenarios.symmetric_arrivals(2, gap_s=3.4, rows=40, lane_groups=[[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]])
        scenarios.check_step_many(BACKEND, "zero", n_envs=2, chunks=(1, 30, 90), trajectory_chunk=10, arrivals=arr)


def test_home_block_driver_shape_vs_oracle(monkeypatch):
    """the bench's call shape (prefill + short persistent calls) against the sequential oracle, every tick"""
    monkeypatch.setenv("PVE_EMU_HOME", "1")
    m, _ = scenarios.check_driver_shape_vs_oracle(BACKEND, n_envs=6, n_sub=1, n_sample=6, calls=(30, 30, 5, 20), chunk=12, persistent=True,
                                                  table=True)
    assert m["ctl_steps"] > 0


def test_home_block_items_of_one_and_two_ticks(monkeypatch):
    """every tick is an item's LAST tick (the jerks park in the free act_next[] cells, the state is flushed and re-loaded per item)"""
    import numpy as np
    import torch
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from tests.hip_adapter import make_batch
    monkeypatch.setenv("PVE_EMU_HOME", "1")
    arr = synthetic_arrivals(3, rate=1300.0, horizon_s=60.0, seed=5)
    one, many = (make_batch(arr, 3, 128, BACKEND, outputs=("obs_post", "reward", "flags", "env_out")) for _ in range(2))
    one.reset(); many.reset()
    table = torch.as_tensor(np.random.default_rng(5).uniform(-3, 3, size=(17, 120)))
    one.set_action_table(table); many.set_action_table(table)
    for n, ch in ((150, 25), (7, 1), (9, 2), (5, 1), (40, 3)):
        for _ in range(n):
            one.step(one.actions_from_table())
        many.step_many(n, source="table", chunk=ch, persistent=True)
        assert many.last_launch() == "persistent"
        scenarios.batches_equal(one, many, "items of <= %d ticks" % ch)


def test_home_block_float32_observation_rows(monkeypatch):
    """PVE_CFG_OBS_F32 through the HOME block: rows, state and headers == single ticks, bit for bit"""
    import numpy as np
    import torch
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from tests.hip_adapter import make_batch
    monkeypatch.setenv("PVE_EMU_HOME", "1")
    arr = synthetic_arrivals(4, rate=1200.0, horizon_s=50.0, seed=6)
    one, many = (make_batch(arr, 4, 128, BACKEND, outputs=("obs_post", "reward", "flags", "env_out"), obs_dtype=torch.float32)
                 for _ in range(2))
    one.reset(); many.reset()
    pool = torch.as_tensor(np.random.default_rng(6).uniform(-3, 3, size=(5, 4, 128)))
    many.set_action_pool(pool)
    for n, ch in ((120, 25), (33, 7), (50, 12)):
        for _ in range(n):
            one.step(pool[one.ticks % 5])
        many.step_many(n, source="pool", chunk=ch, persistent=True)
        assert many.last_launch() == "persistent" and many.obs.dtype == torch.float32
        scenarios.batches_equal(one, many, "float32 rows, items of <= %d ticks" % ch)
