"""SURVEY.md §8 f4 on CPU: the general-geometry kernel phases (csrc/pve_tick_geo.h, lane_num 4 / 8 / 12) executed by
the test emulator and compared with the golden vectors of the live reference and with the sequential oracle.
The `-m gpu` twin (test_gpu_parity.py) runs the real kernel."""
import pytest
import torch

from tests.parity_util import GEO_CASE_NAMES, GoldenCase, replay_case
from tests.hip_adapter import SplitEnv, make_batch
from tests import scenarios

BACKEND = "emu"


@pytest.mark.parametrize("name", GEO_CASE_NAMES)
def test_geo_split_protocol_matches_golden(name):
    case = GoldenCase(name)
    assert scenarios.check_geo_golden(case, BACKEND) == case.ticks


@pytest.mark.parametrize("name", ["geo_g4_sin3", "geo_g8_sin3"])
def test_geo_matches_oracle_every_field(name):
    scenarios.check_geo_vs_oracle(GoldenCase(name), BACKEND, ticks=400)


@pytest.mark.parametrize("name", ["s1000_sin1", "s400_sin2", "s1000_sin3"])
def test_general_path_with_12_lanes_reproduces_the_12_lane_vectors(name):
    """lane_num = 12 through the general-geometry kernel == the golden tapes that pin the fast path."""
    case = GoldenCase(name)
    b = make_batch(case.arrive, 1, 128, BACKEND, general_path=True, **case.ctor)
    replay_case(case, SplitEnv(b), ftol=1e-9, dtol=1e-9, want_state=True)


def test_geo_fused_equals_split():
    scenarios.check_geo_fused_equals_split(GoldenCase("geo_g4_sin3"), BACKEND, ticks=250)
    scenarios.check_geo_fused_equals_split(GoldenCase("geo_g8_sin3"), BACKEND, ticks=250)


def test_general_path_equals_fast_path_bit_for_bit():
    scenarios.check_general_path_equals_fast_path(BACKEND, n_envs=3, ticks=200)


@pytest.mark.parametrize("lane_num,rate,cap,quant,seed", [(4, 1800.0, 64, None, 31), (8, 1500.0, 128, 1.0, 32),
                                                          (4, 2400.0, 128, 3.0, 33)])
def test_geo_fuzz_random_tapes(lane_num, rate, cap, quant, seed):
    coll, lock = scenarios.check_geo_fuzz_vs_oracle(BACKEND, lane_num, n_envs=4, capacity=cap, ticks=300, rate=rate,
                                                    seed=seed, quantize=quant)
    assert coll > 0 and lock > 0


@pytest.mark.parametrize("rate,quant,scale,seed", [(2400.0, None, 2.0, 61), (2700.0, 0.5, 3.0, 62), (1900.0, 1.0, 0.5, 63), (2200.0, 3.0, 3.0, 64),
                                                   (1500.0, None, 3.0, 65)])
def test_four_lane_far_conflict_key_merge_at_128_slots(rate, quant, scale, seed):
    """Round 6 (VERDICT r5 #2a): TickGeo::walk_merge4 -- the 4-lane layout's window walk + merge of the opposing left-turn
    entries (ref :1301-1319, :1340-1405) as one selection on 32-bit keys, the kernels' form at 128 slots -- against the
    sequential oracle under dense traffic: continuous tapes (the ulp-level near ties of equally spaced platoons take the exact
    window + float64 insertion), quantised tapes (exact distance ties, several opposing entries clamped to the ego's own
    position +- 1), every tick, every field."""
    coll, lock = scenarios.check_geo_fuzz_vs_oracle(BACKEND, 4, n_envs=6, capacity=128, ticks=500, rate=rate, seed=seed, action_scale=scale,
                                                    quantize=quant)
    assert coll > 0 and lock > 0


def test_left_neighbours_one_ulp_apart_share_a_distance():
    """Regression (found by tools/soak.py): two LEFT neighbours whose virtual distances differ by one ulp have the same
    float64 |vd - vd_self|; the reference's stable sort then keeps LIST order (ascending vd), i.e. the farther one first.
    Env 5 of this seed meets that at tick 152 (quantised actions; distances 48.50999999999991 / 48.509999999999906)."""
    scenarios.check_geo_fuzz_vs_oracle(BACKEND, 8, n_envs=24, capacity=128, ticks=160, rate=1600.0, seed=3110, quantize=1.0)


@pytest.mark.parametrize("lane_num", [4, 8])
def test_geo_overflow_empty_exhausted(lane_num):
    scenarios.check_geo_overflow_and_empty(BACKEND, lane_num)


@pytest.mark.parametrize("lane_num,rate,quant", [(8, 1500.0, None), (8, 2400.0, 1.0), (4, 1800.0, 1.0), (12, 1100.0, None)])
def test_geo_list_path_equals_scan_fallback(lane_num, rate, quant):
    m = scenarios.check_geo_lists_equal_scan("emu", lane_num, n_envs=3, ticks=200, rate=rate, quantize=quant)
    assert m["ctl_steps"] > 2000


def test_emulated_step_many_geo_equals_single_ticks():
    """k_rollout_geo's staging sequence (emulated) == one k_tick_geo per tick, 4 and 8 lanes, both capacities."""
    scenarios.check_step_many_geo(BACKEND, 8, n_envs=3, chunks=(1, 7, 30, 3), trajectory_chunk=8)
    scenarios.check_step_many_geo(BACKEND, 4, n_envs=3, capacity=64, chunks=(1, 7, 40, 3), trajectory_chunk=8, quantize=1.0)
    scenarios.check_step_many_geo(BACKEND, 4, n_envs=2, capacity=128, chunks=(25, 30), trajectory_chunk=6)


@pytest.mark.parametrize("lane_num,gap", [(8, 3.0), (4, 1.6)])
def test_geo_symmetric_lanes_equal_distances(lane_num, gap):
    """All lanes spawn in the same tick and nobody steers: runs of equal virtual distances in the per-route lists every
    tick (RANK's claim / fix-up path of the general-geometry kernels) and symmetric collisions inside the box."""
    arr = scenarios.symmetric_arrivals(2, gap_s=gap, rows=60, lane_groups=[list(range(lane_num))], lane_num=lane_num)
    for scale, quant in ((0.0, None), (3.0, 3.0)):
        scenarios.check_geo_fuzz_vs_oracle(BACKEND, lane_num, n_envs=2, capacity=128, ticks=260, rate=0.0, seed=13,
                                           action_scale=scale, quantize=quant, arrivals=arr)


@pytest.mark.parametrize("lane_num,cap,dtype,chunk", [(8, 128, torch.float64, 0), (4, 64, torch.float32, 7), (4, 128, torch.float64, 9)])
def test_emulated_step_many_geo_emits_training_states(lane_num, cap, dtype, chunk):
    """f3 x f4: obs_pre / state_pre / the 7-action vectors of every tick of pve_step_many trajectories for the 4- / 8-lane
    layouts (k_rollout_geo<.., TRAIN>), float64 and float32 rows, against OracleGeoEnv."""
    scenarios.check_step_many_state_rows(BACKEND, n_envs=2, capacity=cap, calls=(30, 12, 25), chunk=chunk, obs_dtype=dtype,
                                         lane_num=lane_num, seed=85 + lane_num, min_ctl_per_tick=3)


@pytest.mark.parametrize("lane_num,cap", [(8, 128), (4, 64)])
def test_emulated_step_many_geo_table_source(lane_num, cap):
    """PVE_SRC_TABLE for the 4- / 8-lane layouts (k_rollout_geo<.., IDT>) == single ticks with the table applied per tick."""
    scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=3, capacity=cap, chunks=(1, 7, 40, 3), trajectory_chunk=8, source="table")


@pytest.mark.parametrize("lane_num,cap", [(8, 128), (4, 64)])
def test_geo_work_queue_item_schedule_emulated(lane_num, cap):
    """The persistent form of the general-geometry roll-out through the emulator's sequential work queue."""
    scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=3, capacity=cap, chunks=(1, 7, 40, 20), trajectory_chunk=8, persistent=True)


@pytest.mark.parametrize("lane_num,cap,dtype", [(4, 128, torch.float64), (8, 64, torch.float32)])
def test_emulated_closed_loop_geo_vs_oracle_and_step_many(lane_num, cap, dtype):
    """The closed loop for lane_num 4 / 8 (main.py:398-441; the shipped checkpoint's args.txt records lane_num = 4): actor
    launch + general-geometry tick against the sequential oracle fed with the same actions, then pve_step_many(PVE_SRC_ACTOR)
    == step_with_actor ticks (the emulator enqueues per-tick launches; the resident form is the GPU suite's)."""
    scenarios.check_step_many_geo_actor(BACKEND, lane_num, n_envs=3, capacity=cap, chunks=(1, 9, 25), trajectory_chunk=7,
                                        obs_dtype=dtype, oracle_ticks=120)


@pytest.mark.parametrize("lane_num,cap", [(8, 64), (4, 128)])
def test_geo_work_queue_table_source_emulated(lane_num, cap):
    """PVE_SRC_TABLE through the work queue for the 4- / 8-lane layouts (k_rollout_geo<.., IDT, PERS>; round 5)."""
    scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=3, capacity=cap, chunks=(1, 7, 40, 20), trajectory_chunk=8, source="table",
                                  persistent=True)


def test_geo_work_queue_training_outputs_emulated():
    """The trainer's roll-out of the 4- / 8-lane layouts through the work queue (k_rollout_geo<.., TRAIN, PERS>; the 4-lane
    variant: round 6): obs_pre / state_pre / 7-action vectors of every tick against the oracle."""
    scenarios.check_step_many_state_rows(BACKEND, n_envs=3, capacity=64, calls=(25, 12, 30), chunk=7, lane_num=8, persistent=True,
                                         obs_dtype=torch.float32, min_ctl_per_tick=1)
    scenarios.check_step_many_state_rows(BACKEND, n_envs=2, capacity=128, calls=(20, 9), chunk=6, lane_num=4, persistent=True,
                                         min_ctl_per_tick=1)
    scenarios.check_step_many_state_rows(BACKEND, n_envs=3, capacity=64, calls=(25, 12, 30), chunk=7, lane_num=4, persistent=True,
                                         obs_dtype=torch.float32, min_ctl_per_tick=1)

