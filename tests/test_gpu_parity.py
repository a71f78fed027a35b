"""Parity tests proper: the real HIP kernels (libpveenv.so, through the C ABI) on an MI355X against
the CPU oracle and the golden vectors. Integers bit-exact; floats within the north-star tolerance
|x-y| <= 1e-5*max(1,|x|) (we assert 1e-9: the only non-IEEE-identical operations are
tanh/log/sin/cos of the device math library, which feed rewards and the XY collision distance)."""
import os

import numpy as np
import pytest
import torch

from tests import scenarios
from tests.hip_adapter import SplitEnv, make_batch
from tests.parity_util import CASE_NAMES, GoldenCase, replay_case

pytestmark = pytest.mark.gpu
BACKEND = "hip"
TOL = 1e-9


def test_native_library_is_the_one_loaded():
    from pve_mcc_amd import _capi
    lib = _capi.load_library()
    assert os.path.basename(lib._name) == "libpveenv.so"
    with open("/proc/self/maps") as f:
        assert any("libpveenv.so" in line for line in f), "HIP library not mapped into the process"


@pytest.mark.parametrize("name", CASE_NAMES)
def test_gpu_split_protocol_matches_golden(name):
    case = GoldenCase(name)
    b = make_batch(case.arrive, 1, 128, BACKEND, **case.ctor)
    env = SplitEnv(b)
    replay_case(case, env, ftol=TOL, dtol=TOL, want_state=False)
    assert b.metrics()["overflow"] == 0


@pytest.mark.parametrize("name", ["s1000_sin1", "s200_sin1", "s1000_sin3", "s1200_zero"])
def test_gpu_matches_oracle_every_field(name):
    scenarios.check_split_vs_oracle(GoldenCase(name), BACKEND, ticks=400, tol=TOL)


def test_gpu_capacity_64():
    scenarios.check_split_vs_oracle(GoldenCase("s400_sin2"), BACKEND, ticks=600, capacity=64, tol=TOL)


def test_gpu_fused_equals_split():
    scenarios.check_fused_equals_split(GoldenCase("s1000_sin3"), BACKEND, ticks=300)
    scenarios.check_fused_equals_split(GoldenCase("s200_sin1"), BACKEND, ticks=300, capacity=64)


def test_gpu_batch_of_independent_envs():
    scenarios.check_batch_independent(BACKEND, n_envs=24, capacity=64, ticks=200)
    scenarios.check_batch_independent(BACKEND, n_envs=6, capacity=128, ticks=300, rate=1100.0)


def test_gpu_overflow_and_empty():
    scenarios.check_overflow(BACKEND)
    scenarios.check_empty_and_exhausted(BACKEND)


def test_gpu_full_size_invariants():
    """BASELINE config sizes (4096 envs x 128 slots): size-independent properties -- slot order is
    (lane, j) sorted with unique ids, lane counts sum to n_alive, clocks advance identically, the
    result is independent of which workgroup simulates an env (env e == env e + 2048 on equal streams),
    and conservation: spawned = alive + deleted."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    half = 2048
    arr = synthetic_arrivals(half, rate=1100.0, horizon_s=40.0, seed=20250213)
    arr = np.concatenate([arr, arr], axis=0)
    b = make_batch(arr, 2 * half, 128, BACKEND, outputs=("obs_post", "reward", "flags", "env_out", "new_slot"))
    b.reset()
    g = torch.Generator().manual_seed(5)
    deleted = torch.zeros(2 * half, dtype=torch.int64, device="cuda")
    for t in range(250):
        a = (torch.rand(half, 128, generator=g, dtype=torch.float64) * 4 - 2)
        a = torch.cat([a, a], 0).cuda()
        out = b.step(a)
        deleted += out["env_out"][:, 4].long()
    for k in ("p", "v", "a", "jerk_sum", "id", "meta", "step"):
        x = b.state_field(k)
        assert torch.equal(x[:half], x[half:]), "state %s depends on the workgroup index" % k
    hdr = b.workspace[:b.n_envs * 0 + 1]  # touch workspace (kept alive)
    ids = b.state_field("id").cpu().numpy()
    meta = b.state_field("meta").cpu().numpy()
    tot_alive = 0
    for e in list(range(0, half, 97)) + [half - 1]:
        info = b.read_env(e)
        n = info.n_alive
        tot_alive += n
        assert sum(info.lane_count) == n
        assert np.all((meta[e, :n] & 0x80) != 0) and np.all(meta[e, n:] == 0)
        assert len(set(ids[e, :n].tolist())) == n
        assert info.id_seq == n + int(deleted[e].item()), "conservation violated in env %d" % e
        assert abs(info.current_time - b.read_env(e + half).current_time) == 0
    m = b.metrics()
    assert m["ticks"] == 250 * 2 * half and m["overflow"] == 0


@pytest.mark.parametrize("name,ticks", [("s1000_sin1", 200), ("s200_sin1", 1300)])
def test_gpu_compat_class_reproduces_golden(name, ticks):
    """The drop-in TrafficInteraction class on the real kernels, driven like main.py drives the reference."""
    from tests.test_compat_class import run_compat
    env = run_compat(name, ticks, "hip")
    assert env.id_seq > 0


@pytest.mark.parametrize("name", ["s1000_sin1", "s1000_sin3"])
def test_gpu_full_state_rows_fresh_and_stale(name):
    """state_pre (7x28, neighbour rows fresh/stale by order, ref :1332) and the 7-action vector."""
    from oracle.oracle import OracleEnv
    from oracle.record import compare_records
    case = GoldenCase(name)
    orc = OracleEnv(case.arrive, **case.ctor)
    env = SplitEnv(make_batch(case.arrive, 1, 128, BACKEND, **case.ctor))
    for t in range(300):
        va, ca, oa = orc.alive_view()
        acts = case.policy(t, va, ca, oa)
        ra, rb = orc.tick(acts, want_state=True), env.tick(acts, want_state=True)
        assert rb["state"] is not None
        compare_records(ra, rb, tol=TOL, label=name)


def test_gpu_actor_matches_numpy_restatement():
    """k_actor (float32, scalar-broadcast weights) vs the pinned NumPy actor on the states of the oracle's
    closed loop: action-level tolerance 5e-4 on a in [-3, 3] (float32 round-off through 3 LayerNorms)."""
    from tests import actor_scenarios as A
    worst = A.check_actions_on_oracle_states("hip", ticks=300)
    assert worst <= A.ACTION_TOL


def test_gpu_closed_loop_actor_plus_tick():
    from tests import actor_scenarios as A
    A.check_closed_loop_on_device("hip", ticks=1000)


@pytest.mark.parametrize("seed,rate,cap,quant", [(11, 1000.0, 128, None), (12, 500.0, 64, 1.0), (13, 1100.0, 128, 3.0)])
def test_gpu_fuzz_random_tapes(seed, rate, cap, quant):
    """Random / quantised action tapes (ties, long dead-lock cycles, collisions) on 16 envs vs 16 oracles."""
    coll, lock = scenarios.check_fuzz_vs_oracle(BACKEND, n_envs=16, capacity=cap, ticks=500, rate=rate, seed=seed,
                                                quantize=quant)
    assert coll > 0 and lock > 0


@pytest.mark.parametrize("groups,scale,quant", [([list(range(12))], 0.0, None),
                                                ([[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]], 3.0, 3.0),
                                                ([[0, 6], [3, 9], [1, 7, 4, 10], [2, 5, 8, 11]], 2.0, 1.0)])
def test_gpu_fuzz_symmetric_lanes_equal_distances(groups, scale, quant):
    """Lanes that spawn in the same tick: runs of 2 .. 4 equal virtual distances in every list, every tick -- the claim /
    fix-up path of RANK (tagged exchange, entries of one run in different waves) and WALK's exact path -- 16 envs vs 16
    oracles every tick."""
    arr = scenarios.symmetric_arrivals(16, gap_s=3.4, rows=48, lane_groups=groups)
    scenarios.check_fuzz_vs_oracle(BACKEND, n_envs=16, capacity=128, ticks=400, rate=0.0, seed=9, action_scale=scale,
                                   quantize=quant, arrivals=arr)


def test_gpu_step_many_symmetric_lanes_equal_distances():
    """The resident kernel on the same streams (zero actions: the runs of equal distances persist until the vehicles
    collide): pve_step_many == single ticks, bit for bit."""
    arr = scenarios.symmetric_arrivals(6, gap_s=3.4, rows=48, lane_groups=[[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]])
    scenarios.check_step_many(BACKEND, "zero", n_envs=6, chunks=(1, 30, 90, 7, 120), trajectory_chunk=20, arrivals=arr)


@pytest.mark.parametrize("lane_num,gap", [(8, 3.0), (4, 1.6)])
def test_gpu_geo_symmetric_lanes_equal_distances(lane_num, gap):
    """The same for the 4- / 8-lane layouts (per-route lists of k_tick_geo): all lanes spawn in the same tick."""
    arr = scenarios.symmetric_arrivals(8, gap_s=gap, rows=70, lane_groups=[list(range(lane_num))], lane_num=lane_num)
    for scale, quant in ((0.0, None), (3.0, 3.0)):
        scenarios.check_geo_fuzz_vs_oracle(BACKEND, lane_num, n_envs=8, capacity=128, ticks=320, rate=0.0, seed=13,
                                           action_scale=scale, quantize=quant, arrivals=arr)


def test_gpu_fuzz_more_than_64_controlled_vehicles():
    """Dense traffic (1400 / 1500 veh/h/lane, gentle actions): more than 64 controlled vehicles per intersection, i.e. the
    second wave takes part in the dense-mapped phases; 16 envs vs 16 oracles every tick, no deferred spawn."""
    for rate, seed in ((1400.0, 5), (1500.0, 6)):
        scenarios.check_fuzz_vs_oracle(BACKEND, n_envs=16, capacity=128, ticks=420, rate=rate, seed=seed, action_scale=0.3)
        assert scenarios.check_fuzz_vs_oracle.max_ctl > 64


def test_gpu_step_many_in_dense_traffic():
    """pve_step_many == single ticks bit for bit at 1500 veh/h/lane (more than 64 controlled vehicles, few still ticks)."""
    scenarios.check_step_many(BACKEND, "pool", n_envs=6, prefill=320, rate=1500.0, chunks=(25, 7), trajectory_chunk=10, seed=9)


def test_gpu_reset_replays_the_same_episode():
    scenarios.check_reset_reproducible(BACKEND)


def test_gpu_side_stream_and_two_handles():
    """Kernels follow torch's current stream (pve_set_stream); two handles on two streams run concurrently and
    reproduce the default-stream result bit for bit."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    arr = synthetic_arrivals(64, rate=900.0, horizon_s=40.0, seed=9)
    g = torch.Generator().manual_seed(3)
    acts = (torch.rand(60, 64, 128, generator=g, dtype=torch.float64) * 4 - 2).cuda()

    def run(stream):
        b = make_batch(arr, 64, 128, BACKEND, outputs=("obs_post", "reward", "flags", "env_out"))
        with torch.cuda.stream(stream):
            b.reset()
            for t in range(60):
                b.step(acts[t])
        return b
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    b1, b2 = run(s1), run(s2)
    b0 = run(torch.cuda.current_stream())
    torch.cuda.synchronize()
    for k in ("p", "v", "a", "id", "meta", "step"):
        assert torch.equal(b0.state_field(k), b1.state_field(k)) and torch.equal(b0.state_field(k), b2.state_field(k)), k
    assert b0.metrics() == b1.metrics() == b2.metrics()


# ---------------------------------------------------------------- 4- / 8-lane layouts (k_tick_geo)
@pytest.mark.parametrize("name", ["geo_g4_zero", "geo_g4_sin2", "geo_g4_sin3", "geo_g8_zero", "geo_g8_sin2", "geo_g8_sin3"])
def test_gpu_geo_split_protocol_matches_golden(name):
    case = GoldenCase(name)
    assert scenarios.check_geo_golden(case, BACKEND) == case.ticks


@pytest.mark.parametrize("name", ["geo_g4_sin3", "geo_g8_sin3"])
def test_gpu_geo_matches_oracle_every_field(name):
    scenarios.check_geo_vs_oracle(GoldenCase(name), BACKEND, ticks=400, tol=TOL)


def test_gpu_geo_fused_equals_split():
    scenarios.check_geo_fused_equals_split(GoldenCase("geo_g4_sin3"), BACKEND, ticks=250)
    scenarios.check_geo_fused_equals_split(GoldenCase("geo_g8_sin3"), BACKEND, ticks=250)


def test_gpu_general_path_equals_fast_path_bit_for_bit():
    """lane_num = 12 through k_tick_geo == k_tick on 16 envs x 400 random-action ticks."""
    scenarios.check_general_path_equals_fast_path(BACKEND, n_envs=16, ticks=400)


@pytest.mark.parametrize("name", ["s1000_sin1", "s1000_sin3"])
def test_gpu_general_path_reproduces_12_lane_golden(name):
    case = GoldenCase(name)
    b = make_batch(case.arrive, 1, 128, BACKEND, general_path=True, **case.ctor)
    replay_case(case, SplitEnv(b), ftol=TOL, dtol=TOL, want_state=True)


@pytest.mark.parametrize("lane_num,rate,cap,quant,seed", [(4, 1800.0, 64, None, 31), (8, 1500.0, 128, 1.0, 32),
                                                          (4, 2400.0, 128, 3.0, 33), (8, 1800.0, 128, None, 34)])
def test_gpu_geo_fuzz_random_tapes(lane_num, rate, cap, quant, seed):
    coll, lock = scenarios.check_geo_fuzz_vs_oracle(BACKEND, lane_num, n_envs=16, capacity=cap, ticks=400, rate=rate,
                                                    seed=seed, quantize=quant)
    assert coll > 0 and lock > 0


@pytest.mark.parametrize("rate,quant,scale,seed", [(2400.0, None, 2.0, 61), (2700.0, 0.5, 3.0, 62), (1900.0, 1.0, 0.5, 63), (2200.0, 3.0, 3.0, 64),
                                                   (1500.0, None, 3.0, 65)])
def test_gpu_four_lane_far_conflict_key_merge_at_128_slots(rate, quant, scale, seed):
    """Round 6 (VERDICT r5 #2a): TickGeo::walk_merge4 (the 4-lane window walk + merge of the opposing left-turn entries on
    32-bit keys; ref :1301-1319, :1340-1405) in k_tick_geo<128, .., FIX4> vs the sequential oracle under dense traffic,
    continuous and quantised tapes, every tick, every field; + the resident kernel == single ticks on the same traffic."""
    coll, lock = scenarios.check_geo_fuzz_vs_oracle(BACKEND, 4, n_envs=24, capacity=128, ticks=500, rate=rate, seed=seed, action_scale=scale,
                                                    quantize=quant)
    assert coll > 0 and lock > 0
    scenarios.check_step_many_geo(BACKEND, 4, n_envs=12, capacity=128, chunks=(1, 7, 60, 25), trajectory_chunk=9, rate=rate, seed=seed,
                                  quantize=quant)


def test_gpu_left_neighbours_one_ulp_apart_share_a_distance():
    """Regression (found by tools/soak.py): two left neighbours whose virtual distances differ by one ulp have the same
    float64 distance to the ego; the reference keeps list order (ascending vd) for them.  8 lanes, seed 3110, tick 152."""
    scenarios.check_geo_fuzz_vs_oracle(BACKEND, 8, n_envs=24, capacity=128, ticks=160, rate=1600.0, seed=3110, quantize=1.0)


@pytest.mark.parametrize("name,ticks", [("geo_g4_sin2", 400), ("geo_g8_sin3", 400)])
def test_gpu_compat_class_4_and_8_lanes(name, ticks):
    from tests.test_compat_class import run_compat
    env = run_compat(name, ticks, "hip")
    assert env.lane_num in (4, 8) and env.id_seq > 0


@pytest.mark.parametrize("lane_num", [4, 8])
def test_gpu_geo_overflow_empty_exhausted(lane_num):
    scenarios.check_geo_overflow_and_empty(BACKEND, lane_num)


def test_gpu_geo_full_size_invariants():
    """4096 8-lane envs x 128 slots: the result does not depend on the workgroup index (env e == env e + 2048 on
    equal streams and draws), ids are unique, lane counts add up, spawned = alive + deleted."""
    from pve_mcc_amd.arrivals import synthetic_arrivals, synthetic_intentions
    half = 2048
    arr = synthetic_arrivals(half, rate=1500.0, horizon_s=40.0, seed=77, lane_num=8)
    ch = synthetic_intentions(half, arr.shape[1], seed=77)
    arr, ch = np.concatenate([arr, arr], 0), np.concatenate([ch, ch], 0)
    b = make_batch(arr, 2 * half, 128, BACKEND, lane_num=8, intentions=ch,
                   outputs=("obs_post", "reward", "flags", "env_out", "new_slot"))
    b.reset()
    g = torch.Generator().manual_seed(6)
    deleted = torch.zeros(2 * half, dtype=torch.int64, device="cuda")
    for t in range(200):
        a = (torch.rand(half, 128, generator=g, dtype=torch.float64) * 4 - 2)
        out = b.step(torch.cat([a, a], 0).cuda())
        deleted += out["env_out"][:, 4].long()
    for k in ("p", "v", "a", "jerk_sum", "id", "meta", "step"):
        x = b.state_field(k)
        assert torch.equal(x[:half], x[half:]), "state %s depends on the workgroup index" % k
    ids = b.state_field("id").cpu().numpy()
    meta = b.state_field("meta").cpu().numpy()
    for e in list(range(0, half, 131)) + [half - 1]:
        info = b.read_env(e)
        n = info.n_alive
        assert sum(info.lane_count) == n and all(c == 0 for c in list(info.lane_count)[8:])
        assert np.all((meta[e, :n] & 0x80) != 0) and np.all(meta[e, n:] == 0)
        assert len(set(ids[e, :n].tolist())) == n
        assert info.id_seq == n + int(deleted[e].item()), "conservation violated in env %d" % e
        assert info.intention_re == info.id_seq                       # one draw per spawn (ref :392)
    m = b.metrics()
    assert m["ticks"] == 200 * 2 * half and m["overflow"] == 0


def test_gpu_float32_observation_rows():
    scenarios.check_obs_f32(BACKEND, lane_num=12, n_envs=8, ticks=200)
    scenarios.check_obs_f32(BACKEND, lane_num=4, n_envs=8, capacity=64, ticks=200)


def test_gpu_closed_loop_on_float32_observations_is_bit_identical():
    from tests import actor_scenarios as A
    A.check_closed_loop_f32_obs_equals_f64("hip", ticks=400, n_envs=16)


def test_gpu_split_actor_long_horizon_drift_is_bounded():
    """Default (split-half) actor vs the exact float32 chain over a 1000-tick closed loop on 16 intersections: actions on
    identical states within 1e-4, no collision in either, every aggregate of the evaluation protocol within 1-2 %."""
    from tests import actor_scenarios as A
    first, worst_a, rel = A.check_split_actor_long_horizon("hip", ticks=1000, n_envs=16)
    print("split vs exact actor: first differing tick %s, max |da| on identical states %.2e, relative aggregate differences %s"
          % (first, worst_a, {k: round(v, 5) for k, v in rel.items()}))


def test_gpu_pipelined_sub_batches_equal_one_batch():
    """Two / three free-running sub-batches on their own HIP streams == one launch over all envs, bit for bit."""
    scenarios.check_pipelined_equals_single(BACKEND, n_envs=64, n_sub=2, ticks=300)
    scenarios.check_pipelined_equals_single(BACKEND, n_envs=37, n_sub=3, ticks=200, actor=True)


@pytest.mark.parametrize("source", ["pool", "zero", "actor", "table"])
def test_gpu_step_many_equals_single_ticks(source):
    """pve_step_many (many ticks per call, action source on the device) == single pve_step_all ticks, bit for bit,
    from a cold start, and again from a filled population (prefill)."""
    scenarios.check_step_many(BACKEND, source, n_envs=6, chunks=(1, 7, 40, 3, 60), trajectory_chunk=12)
    if source == "table":
        scenarios.check_step_many(BACKEND, source, n_envs=9, capacity=64, rate=350.0, prefill=200, chunks=(5, 30, 90), trajectory_chunk=8, seed=6)
    if source == "pool":
        scenarios.check_step_many(BACKEND, source, n_envs=4, prefill=320, chunks=(25,), trajectory_chunk=10, seed=5)
        scenarios.check_step_many(BACKEND, source, n_envs=5, capacity=64, rate=350.0, chunks=(5, 30, 90), trajectory_chunk=8)


@pytest.mark.parametrize("dtype,chunk", [(torch.float64, 0), (torch.float64, 9), (torch.float32, 13)])
def test_gpu_step_many_emits_full_state_rows_fresh_and_stale(dtype, chunk):
    """test_gpu_full_state_rows_fresh_and_stale over pve_step_many trajectories (the resident kernel): 7 x 28 states and
    7-action vectors of every tick against the oracle, across call and launch boundaries; float64 and float32 rows."""
    n = scenarios.check_step_many_state_rows(BACKEND, n_envs=6, calls=(40, 25, 60, 35, 170), chunk=chunk, obs_dtype=dtype)
    assert n > 20000


@pytest.mark.parametrize("lane_num,cap,quant", [(8, 128, None), (4, 64, 1.0), (4, 128, None), (8, 64, 1.0), (12, 128, None)])
def test_gpu_step_many_geo_equals_single_ticks(lane_num, cap, quant):
    """k_rollout_geo (4- / 8-lane layouts resident on the chip; lane_num 12 through the general path as a cross-check) ==
    one k_tick_geo launch per tick, bit for bit, incl. chunked launches and trajectory outputs."""
    rate = {(8, 64): 700.0, (4, 64): 1200.0}.get((lane_num, cap))
    m = scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=12, capacity=cap, chunks=(1, 7, 40, 3, 60, 150), rate=rate,
                                      trajectory_chunk=12, quantize=quant)
    assert m["overflow"] == 0 and m["ctl_steps"] > 5000


@pytest.mark.parametrize("lane_num,cap,dtype,chunk", [(8, 128, torch.float64, 0), (4, 64, torch.float32, 13), (4, 128, torch.float64, 9),
                                                      (8, 64, torch.float32, 0)])
def test_gpu_step_many_geo_emits_full_state_rows(lane_num, cap, dtype, chunk):
    """f3 x f4 on MI355X: k_rollout_geo<.., TRAIN> -- obs_pre / state_pre / 7-action vectors of every tick of pve_step_many
    trajectories for lane_num 4 / 8, float64 and float32 rows, every tick of every env against OracleGeoEnv."""
    scenarios.check_step_many_state_rows(BACKEND, n_envs=4, capacity=cap, calls=(40, 25, 60, 35), chunk=chunk, obs_dtype=dtype,
                                         lane_num=lane_num, seed=85 + lane_num, min_ctl_per_tick=3)


@pytest.mark.parametrize("lane_num,cap,quant", [(8, 128, None), (4, 64, 1.0), (4, 128, None)])
def test_gpu_step_many_geo_table_source(lane_num, cap, quant):
    """PVE_SRC_TABLE inside k_rollout_geo<.., IDT> (lane_num 4 / 8) == single ticks with the same table applied per tick."""
    scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=9, capacity=cap, quantize=quant, source="table")


def test_gpu_step_many_actor_in_dense_traffic():
    """The closed loop inside k_rollout with more than 64 controlled vehicles per intersection (tiles 2 and 3: the second
    wave re-reads the rows of its own dense threads) and with one-wave workgroups (capacity 64: one wave runs every tile):
    bit-identical to actor launch + tick launch per tick."""
    scenarios.check_step_many(BACKEND, "actor", n_envs=8, rate=1450.0, prefill=260, chunks=(30, 7, 45), trajectory_chunk=10, seed=19)
    scenarios.check_step_many(BACKEND, "actor", n_envs=9, capacity=64, rate=420.0, prefill=200, chunks=(25, 3, 40), trajectory_chunk=9, seed=23)


def test_gpu_step_many_pipelined():
    scenarios.check_step_many_pipelined(BACKEND, n_envs=37, n_sub=3, ticks=120)


@pytest.mark.parametrize("cap,rate,many", [(128, 1100.0, 0), (64, 350.0, 0), (128, 1100.0, 30), (64, 350.0, 35)])
def test_gpu_full_size_oracle_spot_checks(cap, rate, many):
    """BASELINE configs 2 / 3 (and config 4's per-GPU shard) at FULL size, steady state: 4096 envs, 420 ticks (past the
    300-tick fill), 16 envs spread over the batch compared with their oracles every tick; overflow == 0."""
    m, peak = scenarios.check_full_size_vs_oracle(BACKEND, 4096, cap, rate, ticks=420, n_sample=16, many=many)
    assert m["alive_steps"] / m["ticks"] > (50 if cap == 128 else 12)      # the batch really is at steady state
    assert peak <= cap


@pytest.mark.parametrize("trajectory", [False, True])
def test_gpu_driver_launch_shape_vs_oracle(trajectory):
    """The exact launch shape of the driver's `bench.py --steps 20 --warmup 5` at 4096 x 128 (PipelinedIntersections,
    2 x 2048 envs, pve_step_many calls in launches of 5 ticks, 325 ticks) against 16 oracles; and the same with the
    per-tick outputs retained in the trajectory ring (every tick compared)."""
    m, peak = scenarios.check_driver_shape_vs_oracle(BACKEND, trajectory=trajectory)
    assert m["alive_steps"] / m["ticks"] > 50 and peak <= 128


@pytest.mark.parametrize("cap,n_envs,chunks,rate", [(128, 37, (1, 7, 40, 3, 60), 1100.0), (64, 300, (30, 5, 47), 400.0),
                                                    (128, 700, (25, 61), 1100.0)])
def test_gpu_persistent_step_many_equals_single_ticks(cap, n_envs, chunks, rate):
    """pve_step_many(persistent = 1): one launch per call, the workgroups pull (intersection, chunk) items from the queue
    (intersections change hands between workgroups and CUs inside the launch) == single ticks, bit for bit."""
    scenarios.check_step_many(BACKEND, "pool", n_envs=n_envs, capacity=cap, chunks=chunks, rate=rate, persistent=True)
    scenarios.check_step_many(BACKEND, "zero", n_envs=max(9, n_envs // 8), capacity=cap, chunks=chunks, rate=rate, persistent=True, seed=5)
    scenarios.check_step_many(BACKEND, "table", n_envs=max(9, n_envs // 4), capacity=cap, chunks=chunks, rate=rate, persistent=True, seed=6,
                              trajectory_chunk=24)
    scenarios.check_step_many(BACKEND, "actor", n_envs=max(9, n_envs // 4), capacity=cap, chunks=chunks, rate=min(rate, 1000.0),
                              persistent=True, seed=7, trajectory_chunk=24)


@pytest.mark.parametrize("lane_num,cap,n_envs,quant", [(8, 128, 300, None), (4, 64, 600, 1.0), (4, 128, 40, None), (8, 64, 9, 1.0)])
def test_gpu_persistent_step_many_geo_equals_single_ticks(lane_num, cap, n_envs, quant):
    """The persistent work-queue launch of the general-geometry kernel (k_rollout_geo<.., PERS>) == k_tick_geo ticks."""
    scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=n_envs, capacity=cap, quantize=quant, persistent=True, chunks=(1, 7, 40, 25),
                                  trajectory_chunk=16)


@pytest.mark.parametrize("lane_num,cap,n_envs,dtype,persistent", [
    (4, 128, 12, torch.float64, False), (8, 128, 10, torch.float32, False), (4, 64, 14, torch.float32, False), (8, 64, 9, torch.float64, False),
    (4, 128, 40, torch.float32, True), (8, 128, 300, torch.float32, True), (4, 64, 600, torch.float64, True), (8, 64, 9, torch.float32, True)])
def test_gpu_closed_loop_geo_resident_equals_two_launches(lane_num, cap, n_envs, dtype, persistent):
    """VERDICT r4 missing #4: the closed loop for lane_num 4 / 8 inside the resident kernel (k_rollout_geo<.., ACT[, PERS]>:
    pve_step_many(PVE_SRC_ACTOR), the actions handed from item to item in the queue form) == actor launch + k_tick_geo per tick,
    bit for bit -- after the two-launch form itself was held to the sequential oracle under the device actor's actions."""
    m = scenarios.check_step_many_geo_actor(BACKEND, lane_num, n_envs=n_envs, capacity=cap, chunks=(1, 9, 30, 4, 45), obs_dtype=dtype,
                                            persistent=persistent, oracle_ticks=60 if n_envs <= 40 else 0)
    print("lane_num %d x %d closed loop (persistent=%s): %s" % (lane_num, cap, persistent, {k: m[k] for k in ("spawned", "passed", "collided", "ctl_steps")}))


@pytest.mark.parametrize("lane_num,cap,n_envs,quant", [(8, 128, 300, None), (4, 64, 600, 1.0), (4, 128, 40, None), (8, 64, 9, 1.0)])
def test_gpu_persistent_geo_table_source(lane_num, cap, n_envs, quant):
    """PVE_SRC_TABLE through the work queue for lane_num 4 / 8 (k_rollout_geo<.., IDT, PERS>) == ticks with the table applied per tick."""
    scenarios.check_step_many_geo(BACKEND, lane_num, n_envs=n_envs, capacity=cap, quantize=quant, persistent=True, chunks=(1, 7, 40, 25),
                                  trajectory_chunk=16, source="table")


@pytest.mark.parametrize("lane_num,cap,dtype,n_envs", [(8, 128, torch.float64, 40), (8, 64, torch.float32, 120), (4, 128, torch.float32, 12),
                                                       (4, 128, torch.float64, 60), (4, 64, torch.float32, 100)])
def test_gpu_persistent_geo_training_outputs(lane_num, cap, dtype, n_envs):
    """The trainer's roll-out of the 4- / 8-lane layouts through the work queue (k_rollout_geo<.., TRAIN, PERS>: the stale rows of
    an item's first tick are another workgroup's stores) vs the oracle at every tick (lane_num 4: round 6)."""
    scenarios.check_step_many_state_rows(BACKEND, n_envs=n_envs, capacity=cap, calls=(30, 17, 40), chunk=7, lane_num=lane_num,
                                         persistent=True, obs_dtype=dtype, min_ctl_per_tick=1)


def test_gpu_two_persistent_launches_share_the_chip():
    """Two handles, each with its own persistent launch on its own stream (2 x 2048 envs, 2 x 2048 workgroups: twice what the
    chip holds at once, so workgroups of both launches wait for slots while others spin on their hand-offs), and a batch of
    3 envs (fewer intersections than shards): same results as the oracle / as single ticks."""
    m, peak = scenarios.check_driver_shape_vs_oracle(BACKEND, n_sub=2, chunk=4, persistent=True, calls=(50, 50, 50, 50, 50, 50, 5, 20))
    assert m["alive_steps"] / m["ticks"] > 50 and peak <= 128
    scenarios.check_step_many(BACKEND, "pool", n_envs=3, capacity=128, chunks=(9, 40, 17), persistent=True, seed=11)


@pytest.mark.parametrize("chunk", [5, 2])
def test_gpu_persistent_driver_shape_vs_oracle(chunk):
    """The persistent launch at full size: ONE batch of 4096 x 128, calls of 50 / 5 / 20 ticks in items of `chunk` ticks
    pulled from the queue by as many workgroups as the chip holds, 16 envs against their oracles after every call."""
    m, peak = scenarios.check_driver_shape_vs_oracle(BACKEND, n_sub=1, chunk=chunk, persistent=True)
    assert m["alive_steps"] / m["ticks"] > 50 and peak <= 128


def test_gpu_full_size_actor_closed_loop_matches_small_batch():
    """step_with_actor on 4096 envs == the same streams stepped as a 16-env batch (config 5 at full size)."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from oracle.actor_np import flat_weights, load_weights
    n, ticks = 4096, 330
    arr = synthetic_arrivals(n, rate=1100.0, horizon_s=ticks * 0.1 + 30, seed=4242)
    sample = np.linspace(0, n - 1, 16).astype(int)
    outs = ("obs_post", "reward", "flags", "env_out", "new_slot")
    big = make_batch(arr, n, 128, BACKEND, outputs=outs)
    small = make_batch(arr[sample], len(sample), 128, BACKEND, outputs=outs)
    w = flat_weights(load_weights())
    for b in (big, small):
        b.reset()
        b.set_actor(w)
    for t in range(ticks):
        big.step_with_actor()
        small.step_with_actor()
    idx = torch.as_tensor(sample, device="cuda")
    for k in ("p", "v", "a", "jerk_sum", "vir_dis", "closer_p", "id", "step", "count", "meta"):
        assert torch.equal(big.state_field(k).index_select(0, idx), small.state_field(k)), k
    assert torch.equal(big.obs.index_select(0, idx), small.obs)
    m = big.metrics()
    assert m["overflow"] == 0 and m["alive_steps"] / m["ticks"] > 40


def test_gpu_full_size_closed_loop_persistent_queue():
    """Config 5 through the persistent launch: ONE batch of 4096 x 128, the actor inside k_rollout<.., ACT, PERS>, items of 6
    ticks pulled from the queue (actions handed from item to item through `actor_actions`) == step_with_actor on a 16-env
    sample batch, bit for bit after every call."""
    m, worst = scenarios.check_closed_loop_rollout_vs_two_launch(BACKEND, n_sub=1, chunk=6, obs_dtype=torch.float32, persistent=True)
    assert m["alive_steps"] / m["ticks"] > 40 and worst <= 5e-4


@pytest.mark.parametrize("chunk,dtype", [(5, torch.float64), (25, torch.float32), (25, torch.float64), (5, torch.float32)])
def test_gpu_full_size_closed_loop_inside_rollout_kernel(chunk, dtype):
    """BASELINE config 5 in its fast form at FULL size: 2 x 2048 envs x 128 slots, 1000 veh/h/lane, the actor inside the
    resident kernel (pve_step_many(PVE_SRC_ACTOR), launches of 5 / 25 ticks, 330 ticks) == step_with_actor on a 16-env
    sample batch, bit for bit after every call (state, rows, outputs); float32 and float64 rows; overflow 0."""
    m, worst = scenarios.check_closed_loop_rollout_vs_two_launch(BACKEND, chunk=chunk, obs_dtype=dtype)
    assert m["alive_steps"] / m["ticks"] > 40
    print("closed loop in k_rollout<ACT>, chunk %d, %s rows: max |a_dev - a_numpy| = %.3e" % (chunk, dtype, worst))


@pytest.mark.parametrize("exact,tol", [(True, 2e-6), (False, 1e-4)])
def test_gpu_actor_kernel_follows_the_canonical_order(exact, tol):
    """exact: k_actor_t (v_mfma_f32_16x16x4_f32, PVE_CFG_ACTOR_F32) computes every dot product and LayerNorm sum in the
    order of csrc/pve_actor.h `actor_canonical` (which the CPU emulator calls): same float32 results up to the final
    tanh's ulps.  Default: k_actor_h (split-half operands on the f16 matrix instruction, float32 accumulation) agrees
    with the float32 chain to 4e-5 on actions in [-3, 3] for random rows of scale up to 150 (harsher than real states) --
    an order of magnitude inside the 5e-4 parity bar."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from oracle.actor_np import flat_weights, load_weights
    n, ticks = 24, 120
    arr = synthetic_arrivals(n, rate=1100.0, horizon_s=ticks * 0.1 + 30, seed=77)
    w = flat_weights(load_weights())
    outs = ("obs_post", "reward", "flags", "env_out")
    dev, emu = make_batch(arr, n, 128, BACKEND, outputs=outs, actor_f32=exact), make_batch(arr, n, 128, "emu", outputs=outs)
    for b in (dev, emu):
        b.reset()
        b.set_actor(w)
        for t in range(ticks):
            b.step(None)
    assert np.array_equal(dev.state_field("meta").cpu().numpy(), emu.state_field("meta").numpy())
    rng = np.random.default_rng(5)
    worst = 0.0
    for scale in (1.0, 30.0, 150.0):
        obs = rng.normal(0, scale, size=(n, 128, 28))
        dev.obs.copy_(torch.as_tensor(obs).cuda())
        emu.obs.copy_(torch.as_tensor(obs))
        a_dev, a_emu = dev.act().cpu().numpy(), emu.act().numpy()
        ctl = (emu.state_field("meta").numpy() & 1) != 0
        assert ctl.sum() > 200 and np.all(a_dev[~ctl] == 0)
        worst = max(worst, np.abs(a_dev - a_emu).max())
    print("actor (exact=%s): max |a_dev - a_canonical| = %.3e" % (exact, worst))
    assert worst <= tol, worst


@pytest.mark.parametrize("lane_num,rate,quant", [(8, 1500.0, None), (8, 2400.0, 1.0), (4, 1800.0, 1.0), (12, 1100.0, None)])
def test_gpu_geo_list_path_equals_scan_fallback(lane_num, rate, quant):
    """General-geometry kernel: sorted per-route lists (incl. the exact-size second chance at 2400 veh/h/lane) == the
    membership-scan fallback (PVE_CFG_GEO_SCAN), bit for bit."""
    m = scenarios.check_geo_lists_equal_scan(BACKEND, lane_num, n_envs=12, ticks=300, rate=rate, quantize=quant)
    assert m["ctl_steps"] > 20000


@pytest.mark.parametrize("exact", [False, True])
def test_gpu_actor_kernels_vs_reference_graph(exact):
    """f1, graph-level pin: k_actor_h (default, split-half f16 matrix instructions) and k_actor_t (PVE_CFG_ACTOR_F32) on
    the golden rows against the actions of the reference's OWN graph (model_data/baseline/66.cptk.meta decoded and
    evaluated op by op, tests/golden/gen_actor_golden.py): |a - a_graph| <= 5e-4 against both the float32 and the
    float64 evaluation, on closed-loop states, random rows of scale 1 / 30 / 150 and the rows where the variance
    epsilon 1e-12 decides (the all-zero row of a fresh vehicle to 1e-5)."""
    from tests import actor_scenarios as A
    worst = A.check_actor_entry_point_vs_graph(BACKEND, n_envs=16, actor_f32=exact)
    print("actor (exact=%s) vs graph, max (|a - f32|, |a - f64|) per kind: %s" % (exact, worst))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_gpu_fused_rollout_actor_vs_reference_graph(dtype):
    """The actor inside the resident kernel (k_rollout<.., ACT>) applies the reference graph's actions: golden rows
    planted in the observation buffer, one tick through pve_step_many(PVE_SRC_ACTOR) == one plain tick with the graph's
    actions as the tape, accelerations to 5e-4, everything discrete identical."""
    from tests import actor_scenarios as A
    worst, seen = A.check_fused_rollout_actor_vs_graph(BACKEND, n_envs=72, obs_dtype=dtype)
    print("k_rollout<ACT> vs graph (%s rows): max |da| = %.3e over %d planted actions seen unclipped" % (dtype, worst, seen))


@pytest.mark.parametrize("source,cap,dtype,chunk", [("pool", 128, torch.float64, 7), ("pool", 128, torch.float32, 10), ("table", 128, torch.float64, 6),
                                                    ("zero", 64, torch.float64, 9), ("table", 64, torch.float32, 5)])
def test_gpu_trainer_rollout_through_the_work_queue(source, cap, dtype, chunk):
    """Round 5 (VERDICT r4 missing #3): what MADDPG training consumes -- re_state (7 x 28, fresh / stale neighbour rows) and the
    7-action vectors (ref :288-292, :1325-1337; main.py:243-266) -- out of the persistent launch, k_rollout<.., TRAIN, PERS> and
    k_rollout<.., TRAIN, IDT, PERS>: intersections change hands between workgroups inside the launch, the stale rows of an
    item's first tick are what the previous item's workgroup stored.  Every tick of every env against the oracle."""
    rate = 1100.0 if cap == 128 else 420.0
    scenarios.check_step_many_state_rows(BACKEND, n_envs=40, capacity=cap, rate=rate, calls=(40, 25, 60, 35), chunk=chunk, obs_dtype=dtype,
                                         source=source, persistent=True, min_ctl_per_tick=(1 if source == "zero" else 5))


@pytest.mark.parametrize("persistent,chunk", [(False, 0), (False, 7), (True, 7)])
def test_gpu_closed_loop_training_rollout(persistent, chunk):
    """k_rollout<.., ACT, TRAIN[, PERS]>: the closed loop with the training outputs == actor launch + k_tick (with its STATE
    phase) per tick, bit for bit."""
    scenarios.check_closed_loop_state_rows(BACKEND, n_envs=48, chunk=chunk, persistent=persistent)
    scenarios.check_closed_loop_state_rows(BACKEND, n_envs=20, capacity=64, rate=420.0, chunk=chunk, persistent=persistent, obs_dtype=torch.float64, seed=3)


@pytest.mark.parametrize("lane_num,persistent,chunk", [(4, False, 0), (4, False, 7), (8, False, 7), (8, True, 7), (4, True, 7)])
def test_gpu_closed_loop_training_rollout_lanes_4_and_8(lane_num, persistent, chunk):
    """Round 6: the closed loop WITH the training outputs for the 4- / 8-lane layouts inside the resident kernel
    k_rollout_geo<.., TRAIN, .., ACT> (asked through the queue it runs as chunked launches: bound by the state writes, the queue
    form measured slower) == actor launch + k_tick_geo with its STATE phase per tick (main.py:243-266, :398-441;
    ref :288-292, :1301-1319), bit for bit: rows, 7 x 28 states, rewards, flags of every tick."""
    rate = {4: 1500.0, 8: 1300.0}[lane_num]
    scenarios.check_closed_loop_state_rows(BACKEND, n_envs=40, rate=rate, chunk=chunk, persistent=persistent, lane_num=lane_num,
                                           want_launch=("resident",), seed=11 + lane_num)
    scenarios.check_closed_loop_state_rows(BACKEND, n_envs=12, capacity=64, rate=rate * 0.6, chunk=chunk, persistent=persistent,
                                           lane_num=lane_num, obs_dtype=torch.float64, want_launch=("resident",), seed=13 + lane_num)


@pytest.mark.parametrize("trajectory", [False, True])
def test_gpu_headline_kernel_id_tape_driver_shape_vs_oracle(trajectory):
    """The bench headline as the driver runs it (round 5): ONE batch of 4096 x 128, BASELINE.md 3's tape by vehicle id gathered on
    the device (k_rollout<.., IDT, PERS>), calls of 50 / 5 / 20 ticks as persistent launches with items of <= 12 ticks -- 16
    intersections shadowed by their oracles after every call (trajectory: at EVERY tick, the retained-outputs leg), final state
    field by field, overflow 0 over the whole batch."""
    m, peak = scenarios.check_driver_shape_vs_oracle(BACKEND, n_sub=1, chunk=12, persistent=True, table=True, trajectory=trajectory,
                                                     calls=(50, 50, 50, 50, 50, 50, 5, 20) if not trajectory else (50, 50, 50, 50, 50, 50, 25, 20))
    assert m["alive_steps"] / m["ticks"] > 50 and peak <= 128
