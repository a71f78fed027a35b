"""-m gpu: the HOME build of the persistent 128-slot kernel (k_rollout<128, 5, ..>: carried per-slot fields in LDS homes, entry
pool of 304 entries worked in passes, 10 workgroups per CU) through the C ABI -- == single ticks of k_tick bit for bit, which
the parity suite holds to the oracle; the CPU twin (tests/test_home_block_emulated.py) runs the same phase bodies on the emulator
and counts the passes.  Also the table source's LATE spawn-action gather (a full intersection: who spawns is known in FIN only;
ADVICE r5) on both kernel builds."""
import pytest

from tests import scenarios

pytestmark = pytest.mark.gpu
BACKEND = "hip"


@pytest.mark.parametrize("source", ["table", "pool", "zero"])
def test_gpu_home_kernel_equals_single_ticks(source):
    scenarios.check_step_many(BACKEND, source, n_envs=9, chunks=(20, 9, 33, 50), trajectory_chunk=14, seed=281, persistent=True)


def test_gpu_home_kernel_full_intersection_multi_pass_and_late_spawn_gather():
    """3000 veh/h/lane, vm = 3 m/s, braking, (almost) no collisions: 128 slots fill up with controlled vehicles -> more than
    304 list entries (several BUILD .. WALK passes: asserted on the emulator with the same streams' twin) and deferred spawns
    (overflow > 0: the table source gathers the spawned vehicle's first action behind FIN)."""
    st = scenarios.check_step_many(BACKEND, "table", n_envs=6, chunks=(250, 40, 7, 60), trajectory_chunk=10, seed=191, rate=3000.0,
                                   cfg=dict(vm=3.0, collision_thr=0.01), act_lo=-3.0, act_hi=-2.0, persistent=True)
    assert st["max_alive"] >= 120 and st["overflow"] > 0, st


def test_gpu_table_source_late_spawn_gather_capacity_64():
    """the 8-workgroup kernels' late gather: 64 slots at a rate that fills them (queue form and plain launches)"""
    for pers in (True, False):
        st = scenarios.check_step_many(BACKEND, "table", n_envs=6, capacity=64, chunks=(200, 40, 7, 60), trajectory_chunk=10, seed=193,
                                       rate=1500.0, cfg=dict(vm=3.0, collision_thr=0.01), act_lo=-3.0, act_hi=-2.0, persistent=pers)
        assert st["overflow"] > 0, st


def test_gpu_home_kernel_driver_shape_vs_oracle():
    m, _ = scenarios.check_driver_shape_vs_oracle(BACKEND, n_envs=512, n_sub=1, n_sample=8, calls=(50, 50, 50, 5, 20), chunk=12,
                                                  persistent=True, table=True)
    assert m["ctl_steps"] > 0


def test_gpu_home_kernel_items_of_one_and_two_ticks():
    """every tick is an item's LAST tick (the jerks park in the free act_next[] cells, the state is flushed and re-loaded per item)"""
    import numpy as np
    import torch
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from tests.hip_adapter import make_batch
    arr = synthetic_arrivals(40, rate=1300.0, horizon_s=60.0, seed=5)
    one, many = (make_batch(arr, 40, 128, BACKEND, outputs=("obs_post", "reward", "flags", "env_out")) for _ in range(2))
    one.reset(); many.reset()
    table = torch.as_tensor(np.random.default_rng(5).uniform(-3, 3, size=(17, 120)))
    one.set_action_table(table); many.set_action_table(table)
    for n, ch in ((150, 25), (7, 1), (9, 2), (5, 1), (40, 3)):
        for _ in range(n):
            one.step(one.actions_from_table())
        many.step_many(n, source="table", chunk=ch, persistent=True)
        assert many.last_launch() == "persistent"
        scenarios.batches_equal(one, many, "items of <= %d ticks" % ch)


def test_gpu_home_kernel_float32_observation_rows():
    """PVE_CFG_OBS_F32 through the HOME block: rows, state and headers == single ticks, bit for bit"""
    import numpy as np
    import torch
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from tests.hip_adapter import make_batch
    arr = synthetic_arrivals(40, rate=1200.0, horizon_s=50.0, seed=6)
    one, many = (make_batch(arr, 40, 128, BACKEND, outputs=("obs_post", "reward", "flags", "env_out"), obs_dtype=torch.float32)
                 for _ in range(2))
    one.reset(); many.reset()
    pool = torch.as_tensor(np.random.default_rng(6).uniform(-3, 3, size=(5, 40, 128))).to(one.device)
    many.set_action_pool(pool)
    for n, ch in ((120, 25), (33, 7), (50, 12)):
        for _ in range(n):
            one.step(pool[one.ticks % 5])
        many.step_many(n, source="pool", chunk=ch, persistent=True)
        assert many.last_launch() == "persistent" and many.obs.dtype == torch.float32
        scenarios.batches_equal(one, many, "float32 rows, items of <= %d ticks" % ch)
