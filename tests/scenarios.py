"""Parity scenarios shared by the CPU (emulated kernel) and GPU (real kernel) test files."""
import numpy as np
import torch

from oracle.oracle import OracleEnv
from oracle.record import close, compare_records, get_policy
from pve_mcc_amd.arrivals import synthetic_arrivals
from tests.hip_adapter import SplitEnv, _np, make_batch, state_snapshot

STATE_F = ("p", "v", "a", "jerk", "jerk_sum", "vir_dis", "closer_p")
STATE_I = ("id", "seq", "vnum", "step", "count", "meta", "hdr")


def check_split_vs_oracle(case, backend, ticks, capacity=128, tol=1e-9):
    orc = OracleEnv(case.arrive, **case.ctor)
    b = make_batch(case.arrive, 1, capacity, backend, **case.ctor)
    env = SplitEnv(b)
    for t in range(min(ticks, case.ticks)):
        va, ca, oa = orc.alive_view()
        vb, cb, ob = env.alive_view()
        assert np.array_equal(va, vb) and np.array_equal(ca, cb), "alive set differs at tick %d" % t
        assert close(np.where(ca[:, None] != 0, oa, 0), ob, tol), "stored observation differs at tick %d" % t
        acts = case.policy(t, va, ca, oa)
        compare_records(orc.tick(acts), env.tick(acts), tol=tol, label=case.name)
    assert b.metrics()["overflow"] == 0
    return b


def batches_equal(b1, b2, what):
    """Persistent state + headers + observations of two batches are bit-identical."""
    for e in range(b1.n_envs):
        i1, i2 = b1.read_env(e), b2.read_env(e)
        for f, _ in i1._fields_:
            x, y = getattr(i1, f), getattr(i2, f)
            x = list(x) if hasattr(x, "__len__") else x
            y = list(y) if hasattr(y, "__len__") else y
            assert x == y, "%s: env %d header field %s: %s vs %s" % (what, e, f, x, y)
        n = i1.n_alive
        for k in STATE_F + STATE_I:
            x, y = _np(b1.state_field(k)[e, :n]), _np(b2.state_field(k)[e, :n])
            assert np.array_equal(x, y), "%s: env %d state %s differs" % (what, e, k)
        ctl = (_np(b1.state_field("meta")[e, :n]) & 1) != 0
        assert np.array_equal(_np(b1.obs[e, :n])[ctl], _np(b2.obs[e, :n])[ctl]), "%s: env %d obs differs" % (what, e)


def check_fused_equals_split(case, backend, ticks, capacity=128, mk=None):
    """pve_step_all == pve_scene_update + pve_compact, bit for bit (same kernels, one launch)."""
    if mk is None:
        mk = lambda: make_batch(case.arrive, 1, capacity, backend, **case.ctor)   # noqa: E731
    bs, bf = mk(), mk()
    bs.reset(); bf.reset()
    batches_equal(bs, bf, "after reset")
    for t in range(min(ticks, case.ticks)):
        n = bs.read_env(0).n_alive
        ids = _np(bs.state_field("id")[0, :n]).astype(np.int64)
        ctl = _np(bs.state_field("meta")[0, :n]) & 1
        acts = torch.zeros(1, capacity, dtype=torch.float64)
        acts[0, :n] = torch.as_tensor(case.policy(t, ids, ctl))
        acts = acts.to(bs.device)
        o1 = {k: v.clone() for k, v in bs.scene_update(acts).items()}
        bs.compact()
        o2 = bf.step(acts)
        for k in ("reward", "flags", "lanej", "nbr", "obs_pre", "env_out"):
            x, y = _np(o1[k]), _np(o2[k])
            if k == "env_out":
                x, y = x[:, :7], y[:, :7]       # n_post differs by design (split counts before compaction)
            if k == "obs_pre":
                c = (_np(o1["flags"]) & 2) != 0
                x, y = x[c], y[c]
            assert np.array_equal(x, y), "tick %d: output %s differs between fused and split" % (t, k)
        batches_equal(bs, bf, "tick %d" % t)


def check_batch_independent(backend, n_envs, capacity, ticks, rate=500.0):
    """Every env of a batch (own arrival stream, own action tape) evolves exactly like a lone oracle."""
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=11)
    b = make_batch(arr, n_envs, capacity, backend, outputs=("obs_post", "reward", "flags", "env_out", "new_slot"))
    b.reset()
    oracles = [OracleEnv(arr[e]) for e in range(n_envs)]
    pols = [get_policy(p) for p in ("sin1", "zero", "sin2", "sin0.5", "sin3")]
    for t in range(ticks):
        acts = np.zeros((n_envs, capacity))
        for e, o in enumerate(oracles):
            vid, ctl, _ = o.alive_view()
            acts[e, :len(vid)] = pols[e % len(pols)](t, vid, ctl)
        out = b.step(torch.as_tensor(acts).to(b.device))
        rew, flags, eo = _np(out["reward"]), _np(out["flags"]), _np(out["env_out"])
        for e, o in enumerate(oracles):
            n = o.n_alive
            rec = o.tick(acts[e, :n])
            ctl = (flags[e, :n] & 2) != 0
            assert int(eo[e, 0]) == n and int(eo[e, 1]) == len(rec["ids"]), (t, e)
            assert close(rec["reward"], rew[e, :n][ctl], 1e-9), "reward: tick %d env %d" % (t, e)
            assert int(eo[e, 2]) == rec["collisions"] and int(eo[e, 3]) == rec["lock"], (t, e)
            assert np.array_equal((flags[e, :n][ctl] >> 8), rec["coll_pv"]), (t, e)
    for e, o in enumerate(oracles):
        info, vi, vf = state_snapshot(b, e)
        ovi, ovf, _ = o.vehicles()
        assert np.array_equal(vi[:, :13], ovi[:, :13]), "final state ints, env %d" % e   # hdr is pre-compaction, skip
        assert close(ovf, vf, 1e-9), "final state floats, env %d" % e
    m = b.metrics()
    assert m["ticks"] == ticks * n_envs and m["overflow"] == 0
    return b


def check_overflow(backend):
    """A full env defers spawns (cursor not advanced, overflow counted) instead of corrupting state."""
    rows = 400
    arr = np.full((rows, 12), np.inf)
    arr[:rows - 1, :] = (np.arange(rows - 1)[:, None] * 1.0 + 1.0)      # one vehicle per lane per second
    b = make_batch(arr, 1, 64, backend, outputs=("obs_post", "env_out", "flags"))
    b.reset()
    seen_over = False
    for t in range(200):
        out = b.step(None)
        info = b.read_env(0)
        assert 0 <= info.n_alive <= 64
        assert sum(info.lane_count) == info.n_alive
        seen_over = seen_over or info.overflow > 0
    assert seen_over, "the scenario was meant to overflow a 64-slot env"
    n = b.read_env(0).n_alive
    ids = _np(b.state_field("id")[0, :n])
    assert len(set(ids.tolist())) == n, "duplicate vehicle ids after overflow"
    assert b.metrics()["overflow"] > 0


def check_empty_and_exhausted(backend):
    """No arrivals at all: reset leaves the env empty and ticks only advance the clock; a stream that
    runs dry stops spawning (cursor == rows is not an error)."""
    arr = np.full((4, 12), np.inf)
    b = make_batch(arr, 2, 64, backend, outputs=("obs_post", "env_out"))
    b.reset()
    assert b.read_env(0).n_alive == 0
    t0 = b.read_env(0).current_time
    for _ in range(5):
        out = b.step(None)
    assert b.read_env(1).n_alive == 0 and abs(b.read_env(1).current_time - (t0 + 0.5)) < 1e-9
    arr2 = np.full((2, 12), np.inf)
    arr2[0, 4] = 0.35
    arr2[1, 4] = 0.95
    b2 = make_batch(arr2, 1, 64, backend, outputs=("obs_post", "env_out"))
    b2.reset()
    info = b2.read_env(0)
    assert info.n_alive == 1 and list(info.lane_count)[4] == 1 and abs(info.current_time - 0.4) < 1e-9
    orc = OracleEnv(np.vstack([arr2, np.full((1, 12), np.inf)]))
    for t in range(420):
        b2.step(None)
        orc.tick(np.zeros(orc.n_alive))
    info = b2.read_env(0)
    assert list(info.veh_rec)[4] == 2 and info.id_seq == 2
    assert info.n_alive == orc.n_alive and info.passed_veh == 2


class CapacityOverflow(Exception):
    """The workload filled every slot of an intersection: the batched env defers the spawn (documented deviation,
    counted in metrics()['overflow']), the reference / oracle does not -- the comparison ends there."""


def _overflow_guard(b, fn):
    try:
        fn()
    except (AssertionError, ValueError) as ex:
        if b.metrics()["overflow"] > 0:
            raise CapacityOverflow(str(ex))
        raise


def symmetric_arrivals(n_envs, gap_s, rows, lane_groups, lane_num=12):
    """Arrival streams in which the lanes of a group spawn in the SAME tick (identical arrival times): while nobody steers
    them apart, vehicles of symmetric lanes keep identical positions, i.e. identical virtual distances in every list they
    share -- runs of 2 .. 4 equal keys in RANK (the claim / fix-up path) in every tick."""
    out = np.full((n_envs, rows, lane_num), np.inf, dtype=np.float64)
    for e in range(n_envs):
        for g, lanes in enumerate(lane_groups):
            t = 1.0 + 0.3 * g + 0.7 * e + gap_s * np.arange(rows - 1)
            for l in lanes:
                out[e, :rows - 1, l] = t
    return out


def check_fuzz_vs_oracle(backend, n_envs, capacity, ticks, rate, seed, action_scale=3.0, quantize=None, arrivals=None):
    """Random action tapes (uniform in [-scale, scale], optionally quantised to provoke exact ties), every env
    compared with its own oracle every tick: controlled set, rewards, collision counters, lock counts, and the
    full persistent state at the end. Exercises vd ties, long dead-lock cycles, collisions, mid-lane deletions."""
    rng = np.random.default_rng(seed)
    arr = arrivals if arrivals is not None else synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=seed)
    b = make_batch(arr, n_envs, capacity, backend, outputs=("obs_post", "obs_pre", "reward", "flags", "nbr",
                                                            "env_out", "new_slot", "lanej"))
    b.reset()
    oracles = [OracleEnv(arr[e]) for e in range(n_envs)]
    tot_coll = tot_lock = 0
    check_fuzz_vs_oracle.max_ctl = 0          # (most controlled vehicles any env held: the dense mapping's second wave)
    def run_ticks():
        nonlocal tot_coll, tot_lock
        for t in range(ticks):
            acts = rng.uniform(-action_scale, action_scale, size=(n_envs, capacity)).astype(np.float32).astype(np.float64)
            if quantize:
                acts = np.round(acts / quantize) * quantize
            out = b.step(torch.as_tensor(acts).to(b.device))
            rew, flags, eo = _np(out["reward"]), _np(out["flags"]), _np(out["env_out"])
            nbr, obs = _np(out["nbr"]), _np(out["obs_pre"])
            for e, o in enumerate(oracles):
                n = o.n_alive
                _vid, ctlm, _ = o.alive_view()
                rec = o.tick(np.where(ctlm != 0, acts[e, :n], 0.0))
                ctl = (flags[e, :n] & 2) != 0
                check_fuzz_vs_oracle.max_ctl = max(check_fuzz_vs_oracle.max_ctl, int(ctl.sum()))
                assert int(eo[e, 0]) == n and int(ctl.sum()) == len(rec["ids"]), "controlled set: tick %d env %d" % (t, e)
                assert int(eo[e, 2]) == rec["collisions"] and int(eo[e, 3]) == rec["lock"], "counters: tick %d env %d" % (t, e)
                assert np.array_equal(flags[e, :n][ctl] >> 8, rec["coll_pv"]), "coll_pv: tick %d env %d" % (t, e)
                nb = nbr[e, :n][ctl].astype(np.int64)
                nb = np.stack([np.where(nb < 0, -1, nb >> 16), np.where(nb < 0, -1, nb & 0xFFFF)], -1)
                assert np.array_equal(nb, rec["nbr"]), "neighbours: tick %d env %d" % (t, e)
                assert close(rec["reward"], rew[e, :n][ctl], 1e-9), "reward: tick %d env %d" % (t, e)
                assert close(rec["obs0"], obs[e, :n][ctl], 1e-9), "obs: tick %d env %d" % (t, e)
                tot_coll += rec["collisions"]
                tot_lock += rec["lock"]
    _overflow_guard(b, run_ticks)
    for e, o in enumerate(oracles):
        info, vi, vf = state_snapshot(b, e)
        ovi, ovf, _ = o.vehicles()
        assert np.array_equal(vi[:, :13], ovi[:, :13]), "final state ints, env %d" % e
        assert close(ovf[:, :5], vf[:, :5], 1e-9), "final state floats, env %d" % e
    assert b.metrics()["overflow"] == 0
    return tot_coll, tot_lock


def check_reset_reproducible(backend):
    """A second reset() replays exactly the same episode (the reference builds a fresh object per episode)."""
    arr = synthetic_arrivals(3, rate=900.0, horizon_s=60.0, seed=5)
    b = make_batch(arr, 3, 128, backend, outputs=("obs_post", "reward", "flags", "env_out"))
    g = torch.Generator().manual_seed(7)
    acts = (torch.rand(80, 3, 128, generator=g, dtype=torch.float64) * 4 - 2)

    def episode():
        b.reset()
        rew = []
        for t in range(80):
            out = b.step(acts[t].contiguous().to(b.device))
            rew.append(_np(out["reward"]).copy())
        return np.stack(rew), {k: _np(b.state_field(k)).copy() for k in ("p", "v", "id", "meta")}, b.metrics()
    r1, s1, m1 = episode()
    r2, s2, m2 = episode()
    assert np.array_equal(r1, r2) and m1 == m2
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k


# ---------------------------------------------------------------- general geometry
def make_geo_batch(case, backend, capacity=128, n_envs=1, **kw):
    cfg = dict(case.ctor)
    cfg.update(kw)
    return make_batch(case.arrive, n_envs, capacity, backend, lane_num=case.lane_num,
                      intentions=case.choice if case.lane_num == 8 else None, **cfg)


def check_geo_golden(case, backend, ticks=None, capacity=128, want_state=True, **kw):
    """Split protocol on the general-geometry kernel vs the golden vectors of the live reference (digest every
    tick, every field on the dense ticks, the 7x28 state on the state ticks)."""
    from tests.parity_util import replay_case
    b = make_geo_batch(case, backend, capacity, **kw)
    env = SplitEnv(b)
    n = replay_case(case, env, ticks=ticks, ftol=1e-9, dtol=1e-9, want_state=want_state)
    assert b.metrics()["overflow"] == 0
    return n


def check_geo_vs_oracle(case, backend, ticks, capacity=128, tol=1e-9):
    """Every tick, every field (incl. the full state) against the sequential general-geometry oracle."""
    from oracle.oracle_geo import OracleGeoEnv
    orc = OracleGeoEnv(case.arrive, case.lane_num, choice=case.choice, **case.ctor)
    b = make_geo_batch(case, backend, capacity)
    env = SplitEnv(b)
    for t in range(min(ticks, case.ticks)):
        va, ca, oa = orc.alive_view()
        vb, cb, ob = env.alive_view()
        assert np.array_equal(va, vb) and np.array_equal(ca, cb), "alive set differs at tick %d" % t
        assert close(np.where(ca[:, None] != 0, oa, 0), ob, tol), "stored observation differs at tick %d" % t
        acts = case.policy(t, va, ca, oa)
        ra, rb = orc.tick(acts, want_state=True), env.tick(acts, want_state=True)
        compare_records(ra, rb, tol=tol, label=case.name)
        assert np.array_equal(ra["intent"], rb["intent"]), "intent differs at tick %d" % t
        assert ra["intention_re"] == rb["intention_re"]
    assert b.metrics()["overflow"] == 0


def check_geo_fused_equals_split(case, backend, ticks, capacity=128):
    check_fused_equals_split(case, backend, ticks, capacity, mk=lambda: make_geo_batch(case, backend, capacity))


def check_general_path_equals_fast_path(backend, n_envs=6, capacity=128, ticks=300, rate=1100.0, seed=21):
    """lane_num = 12 through the general-geometry kernel == the optimised 12-lane kernel, bit for bit, on random
    action tapes (state, headers, observations and every per-tick output)."""
    rng = np.random.default_rng(seed)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=seed)
    outs = ("obs_post", "obs_pre", "reward", "flags", "nbr", "env_out", "new_slot", "lanej")
    bf = make_batch(arr, n_envs, capacity, backend, outputs=outs)
    bg = make_batch(arr, n_envs, capacity, backend, outputs=outs, general_path=True)
    bf.reset(); bg.reset()
    batches_equal(bf, bg, "after reset")
    for t in range(ticks):
        acts = rng.uniform(-3, 3, size=(n_envs, capacity)).astype(np.float32).astype(np.float64)
        a = torch.as_tensor(acts).to(bf.device)
        o1, o2 = bf.step(a), bg.step(a)
        for k in ("reward", "flags", "lanej", "nbr", "env_out", "new_slot"):
            x, y = _np(o1[k]), _np(o2[k])
            if k == "flags":
                y = y & ~(3 << 6)             # the general path also reports the intention (lane % 3)
            assert np.array_equal(x, y), "tick %d: output %s differs between the fast and the general path" % (t, k)
        c = (_np(o1["flags"]) & 2) != 0
        assert np.array_equal(_np(o1["obs_pre"])[c], _np(o2["obs_pre"])[c]), "tick %d: obs_pre differs" % t
        if t % 25 == 0 or t == ticks - 1:
            batches_equal(bf, bg, "tick %d" % t)


def check_geo_fuzz_vs_oracle(backend, lane_num, n_envs, capacity, ticks, rate, seed, action_scale=3.0, quantize=None,
                             arrivals=None):
    """Random action tapes on a batch of 4- or 8-lane envs (own arrival + intention streams), fused ticks, every env
    against its own sequential oracle every tick: processing order, controlled set, neighbours, rewards,
    observations, collision counters, lock counts; full persistent state at the end."""
    from oracle.oracle_geo import OracleGeoEnv
    from pve_mcc_amd.arrivals import synthetic_intentions
    rng = np.random.default_rng(seed)
    arr = arrivals if arrivals is not None else synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=seed,
                                                                   lane_num=lane_num)
    ch = synthetic_intentions(n_envs, arr.shape[1], seed=seed, lane_num=lane_num) if lane_num == 8 else None
    b = make_batch(arr, n_envs, capacity, backend, lane_num=lane_num, intentions=ch,
                   outputs=("obs_post", "obs_pre", "reward", "flags", "nbr", "env_out", "new_slot", "lanej"))
    b.reset()
    oracles = [OracleGeoEnv(arr[e], lane_num, choice=None if ch is None else ch[e]) for e in range(n_envs)]
    tot_coll = tot_lock = 0
    def run_ticks():
        nonlocal tot_coll, tot_lock
        for t in range(ticks):
            acts = rng.uniform(-action_scale, action_scale, size=(n_envs, capacity)).astype(np.float32).astype(np.float64)
            if quantize:
                acts = np.round(acts / quantize) * quantize
            out = b.step(torch.as_tensor(acts).to(b.device))
            rew, flags, eo = _np(out["reward"]), _np(out["flags"]).astype(np.int64), _np(out["env_out"])
            nbr, obs, lanej = _np(out["nbr"]), _np(out["obs_pre"]), _np(out["lanej"]).astype(np.int64)
            for e, o in enumerate(oracles):
                n = o.n_alive
                _vid, ctlm, _ = o.alive_view()
                rec = o.tick(np.where(ctlm != 0, acts[e, :n], 0.0))
                f = flags[e, :n]
                order = np.lexsort((lanej[e, :n] & 0xFFFF, (f >> 6) & 3, lanej[e, :n] >> 16))
                ctl = order[((f & 2) != 0)[order]]
                assert int(eo[e, 0]) == n and len(ctl) == len(rec["ids"]), "controlled set: tick %d env %d" % (t, e)
                ids = np.stack([lanej[e, ctl] >> 16, lanej[e, ctl] & 0xFFFF], -1)
                assert np.array_equal(ids, rec["ids"]), "ids order: tick %d env %d" % (t, e)
                assert int(eo[e, 2]) == rec["collisions"] and int(eo[e, 3]) == rec["lock"], "counters: tick %d env %d" % (t, e)
                assert np.array_equal(f[ctl] >> 8, rec["coll_pv"]), "coll_pv: tick %d env %d" % (t, e)
                nb = nbr[e, ctl].astype(np.int64)
                nb = np.stack([np.where(nb < 0, -1, nb >> 16), np.where(nb < 0, -1, nb & 0xFFFF)], -1)
                assert np.array_equal(nb, rec["nbr"]), "neighbours: tick %d env %d" % (t, e)
                assert close(rec["reward"], rew[e, ctl], 1e-9), "reward: tick %d env %d" % (t, e)
                assert close(rec["obs0"], obs[e, ctl], 1e-9), "obs: tick %d env %d" % (t, e)
                tot_coll += rec["collisions"]
                tot_lock += rec["lock"]
    _overflow_guard(b, run_ticks)
    for e, o in enumerate(oracles):
        info, vi, vf = state_snapshot(b, e)
        ovi, ovf, _, ointent = o.vehicles()
        assert np.array_equal(vi[:, :13], ovi[:, :13]), "final state ints, env %d" % e
        assert close(ovf[:, :5], vf[:, :5], 1e-9), "final state floats, env %d" % e
        got = np.array([[v.intention, v.route] for v in b.read_vehicles(e)], np.int32).reshape(-1, 2)
        assert np.array_equal(got, ointent), "intentions / routes, env %d" % e
    assert b.metrics()["overflow"] == 0
    return tot_coll, tot_lock


def check_geo_overflow_and_empty(backend, lane_num):
    """4- / 8-lane edge cases: a full env defers spawns (cursor, intention counter and ids stay consistent); an
    empty stream only advances the clock; an exhausted stream stops spawning."""
    from oracle.oracle_geo import OracleGeoEnv
    rows = 300
    arr = np.full((rows, lane_num), np.inf)
    arr[:rows - 1, :] = (np.arange(rows - 1)[:, None] * 0.6 + 1.0)       # one vehicle per lane every 0.6 s
    ch = (np.arange(rows * lane_num).reshape(rows, lane_num) % 2).astype(np.int32) if lane_num == 8 else None
    b = make_batch(arr, 1, 64, backend, lane_num=lane_num, intentions=ch, outputs=("obs_post", "env_out", "flags"))
    b.reset()
    seen_over = False
    for t in range(260):
        b.step(None)
        info = b.read_env(0)
        assert 0 <= info.n_alive <= 64 and sum(info.lane_count) == info.n_alive
        assert all(c == 0 for c in list(info.lane_count)[lane_num:])
        seen_over = seen_over or info.overflow > 0
    assert seen_over, "the scenario was meant to overflow a 64-slot env"
    vs = b.read_vehicles(0)
    assert len({v.id for v in vs}) == len(vs)
    assert all(0 <= v.route < b.dir_num and 0 <= v.intention <= 2 for v in vs)
    # empty stream
    e = make_batch(np.full((4, lane_num), np.inf), 2, 64, backend, lane_num=lane_num, outputs=("obs_post", "env_out"))
    e.reset()
    t0 = e.read_env(0).current_time
    for _ in range(5):
        e.step(None)
    assert e.read_env(1).n_alive == 0 and abs(e.read_env(1).current_time - (t0 + 0.5)) < 1e-9
    # exhausted stream: two vehicles on lane 1, then nothing; same life cycle as the oracle
    arr2 = np.full((3, lane_num), np.inf)
    arr2[0, 1], arr2[1, 1] = 0.35, 0.95
    ch2 = np.ones((3, lane_num), np.int32) if lane_num == 8 else None
    x = make_batch(arr2, 1, 64, backend, lane_num=lane_num, intentions=ch2, outputs=("obs_post", "env_out"))
    x.reset()
    orc = OracleGeoEnv(arr2, lane_num, choice=ch2)
    for t in range(450):
        x.step(None)
        orc.tick(np.zeros(orc.n_alive))
    info = x.read_env(0)
    assert list(info.veh_rec)[1] == 2 and info.id_seq == 2 and info.n_alive == orc.n_alive
    assert info.passed_veh == 2 or lane_num == 8      # 8-lane lane 1 = straight / right: both pass as well
    assert info.passed_veh == 2


def check_obs_f32(backend, lane_num=12, n_envs=3, capacity=128, ticks=120, seed=41):
    """PVE_CFG_OBS_F32: the float32 observation rows are exactly float32(float64 rows), the dynamics do not change."""
    from pve_mcc_amd.arrivals import synthetic_intentions
    from pve_mcc_amd._capi import PveError
    rng = np.random.default_rng(seed)
    rate = {12: 1100.0, 8: 1500.0, 4: 1800.0}[lane_num]
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=seed, lane_num=lane_num)
    ch = synthetic_intentions(n_envs, arr.shape[1], seed=seed) if lane_num == 8 else None
    outs = ("obs_post", "reward", "flags", "env_out", "new_slot")
    b64 = make_batch(arr, n_envs, capacity, backend, outputs=outs, lane_num=lane_num, intentions=ch)
    b32 = make_batch(arr, n_envs, capacity, backend, outputs=outs, lane_num=lane_num, intentions=ch,
                     obs_dtype=torch.float32)
    assert b32.obs.dtype == torch.float32 and b32.obs.shape == (n_envs, capacity, 28)
    b64.reset(); b32.reset()
    for t in range(ticks):
        a = torch.as_tensor(rng.uniform(-2, 2, size=(n_envs, capacity))).to(b64.device)
        o64, o32 = b64.step(a), b32.step(a)
        assert np.array_equal(_np(o64["reward"]), _np(o32["reward"])) and np.array_equal(_np(o64["flags"]), _np(o32["flags"]))
        ctl = (_np(b64.state_field("meta")) & 1) != 0
        x64, x32 = _np(b64.obs)[ctl], _np(b32.obs)[ctl]
        assert np.array_equal(x64.astype(np.float32), x32), "tick %d: float32 rows are not float32(float64 rows)" % t
    for k in STATE_F + STATE_I:
        assert np.array_equal(_np(b64.state_field(k)), _np(b32.state_field(k))), k
    # the training outputs follow the row type (every layout since round 4): obs_pre == float32(float64 obs_pre)
    kw = dict(lane_num=lane_num, intentions=ch)
    p64 = make_batch(arr, n_envs, capacity, backend, outputs=("obs_post", "obs_pre", "flags"), **kw)
    p32 = make_batch(arr, n_envs, capacity, backend, outputs=("obs_post", "obs_pre", "flags"), obs_dtype=torch.float32, **kw)
    p64.reset(); p32.reset()
    for t in range(60):
        a = torch.as_tensor(rng.uniform(-2, 2, size=(n_envs, capacity))).to(b64.device)
        o64, o32 = p64.step(a), p32.step(a)
        ctl = (_np(o64["flags"]) & 2) != 0
        assert o32["obs_pre"].dtype == torch.float32
        assert np.array_equal(_np(o64["obs_pre"])[ctl].astype(np.float32), _np(o32["obs_pre"])[ctl]), "tick %d: obs_pre" % t


def check_pipelined_equals_single(backend, n_envs=7, n_sub=3, capacity=128, ticks=150, seed=61, actor=False):
    """PipelinedIntersections (sub-batches on their own streams) evolves every env exactly like one batch."""
    from pve_mcc_amd.batched import PipelinedIntersections
    from tests.hip_adapter import emulator_lib
    rng = np.random.default_rng(seed)
    arr = synthetic_arrivals(n_envs, rate=1100.0, horizon_s=ticks * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "env_out", "new_slot")
    one = make_batch(arr, n_envs, capacity, backend, outputs=outs)
    kw = dict(device="cpu", _lib=emulator_lib()) if backend == "emu" else dict(device="cuda")
    pipe = PipelinedIntersections(n_envs, capacity, arr, n_sub=n_sub, outputs=outs, **kw)
    assert pipe.bounds[-1] == n_envs and len(pipe.subs) == n_sub
    one.reset(); pipe.reset()
    if actor:
        from oracle.actor_np import flat_weights, load_weights
        w = flat_weights(load_weights())
        one.set_actor(w); pipe.set_actor(w)
    for t in range(ticks):
        if actor:
            one.step_with_actor(); pipe.step_with_actor()
        else:
            a = torch.as_tensor(rng.uniform(-2, 2, size=(n_envs, capacity))).to(one.device)
            pipe.wait_stream()                     # the actions were produced on the current stream
            one.step(a); pipe.step(a)
    pipe.synchronize(); one.synchronize()
    for k in STATE_F + STATE_I:
        x = _np(one.state_field(k))
        y = np.concatenate([_np(sub.state_field(k)) for sub in pipe.subs], 0)
        assert np.array_equal(x, y), "state %s differs between the pipelined and the single batch" % k
    assert np.array_equal(_np(one.obs), np.concatenate([_np(sub.obs) for sub in pipe.subs], 0))
    m1, m2 = one.metrics(), pipe.metrics()
    for k in m1:                                   # float sums differ in summation order only
        assert abs(m1[k] - m2[k]) <= 1e-9 * max(1.0, abs(m1[k])), (k, m1[k], m2[k])
    assert pipe.sub_of(n_envs - 1) == (n_sub - 1, pipe.subs[-1].n_envs - 1)


def check_step_many(backend, source, n_envs=5, capacity=128, seed=71, chunks=(1, 7, 40, 3, 60), rate=1100.0,
                    prefill=0, trajectory_chunk=12, arrivals=None, persistent=False, cfg=None, act_lo=-3.0, act_hi=3.0):
    """pve_step_many (n ticks per call, action source on the device) == n single-tick calls, bit for bit: persistent
    state, headers, observation rows, last-tick outputs and -- trajectory mode -- the outputs of every tick.
    persistent=True: the calls that are split into several chunks run as ONE launch whose workgroups pull (intersection,
    chunk) items from the work queue (pve_rollout.persistent)."""
    from pve_mcc_amd._capi import PveError
    n_pool = 5
    rng = np.random.default_rng(seed)
    total = prefill + sum(chunks) + trajectory_chunk
    arr = arrivals if arrivals is not None else synthetic_arrivals(n_envs, rate=rate, horizon_s=total * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out", "lanej")
    cfg = cfg or {}                                         # constructor arguments (vm, dis_ctl, ...: ref :21-23)
    one = make_batch(arr, n_envs, capacity, backend, outputs=outs, **cfg)
    many = make_batch(arr, n_envs, capacity, backend, outputs=outs, **cfg)
    one.reset(); many.reset()
    pool = torch.as_tensor(rng.uniform(act_lo, act_hi, size=(n_pool, n_envs, capacity))).to(one.device)
    if source == "pool":
        many.set_action_pool(pool)
    if source == "actor":
        from oracle.actor_np import flat_weights, load_weights
        w = flat_weights(load_weights())
        one.set_actor(w); many.set_actor(w)
    if source == "table":                                   # actions by (tick, vehicle id); few columns: ids beyond them share the last
        table = torch.as_tensor(rng.uniform(act_lo, act_hi, size=(23, 150)))
        one.set_action_table(table); many.set_action_table(table)

    def single():
        if source == "table":
            return one.step(one.actions_from_table())
        if source == "pool":
            return one.step(pool[one.ticks % n_pool])
        if source == "actor":
            return one.step_with_actor()
        return one.step(None)

    def same_outputs(o1, o2, what):
        f = _np(o1["flags"])
        assert np.array_equal(f, _np(o2["flags"])), what + ": flags"
        alive, ctl = (f & 1) != 0, (f & 2) != 0
        for k in ("reward", "new_slot", "lanej"):
            assert np.array_equal(_np(o1[k])[alive], _np(o2[k])[alive]), what + ": " + k
        assert np.array_equal(_np(o1["nbr"])[ctl], _np(o2["nbr"])[ctl]), what + ": nbr"
        assert np.array_equal(_np(o1["env_out"]), _np(o2["env_out"])), what + ": env_out"

    if prefill:
        for _ in range(prefill):
            single()
        many.step_many(prefill, source=source)
        batches_equal(one, many, "after prefill")
    for n in chunks:
        for _ in range(n):
            o1 = single()
        ch = 0 if n < 6 else (n // 3 + 1)
        o2 = many.step_many(n, source=source, chunk=ch, persistent=persistent)     # also in several launches
        one.synchronize(); many.synchronize()
        assert many.ticks == one.ticks
        # which path ran (a regressed eligibility test would silently fall back to chunked launches and still pass)
        want = "persistent" if (persistent and ch > 0 and n > ch and not (backend == "emu" and source == "actor")) else \
            ("tick" if (backend == "emu" and source == "actor") else "resident")
        assert many.last_launch() == want, (many.last_launch(), want, source, n)
        batches_equal(one, many, "%s, chunk of %d" % (source, n))
        same_outputs(o1, o2, "%s, chunk of %d" % (source, n))
    # trajectory mode: every tick's outputs
    traj = many.step_many(trajectory_chunk, source=source, trajectory=True, chunk=trajectory_chunk // 2 + 1, persistent=persistent)
    for k in range(trajectory_chunk):
        o1 = single()
        same_outputs(o1, {n: traj[n][k] for n in traj}, "trajectory tick %d" % k)
        post_ctl = (_np(one.state_field("meta")) & 1) != 0
        assert np.array_equal(_np(one.obs)[post_ctl], _np(traj["obs_post"][k])[post_ctl]), "trajectory obs, tick %d" % k
    batches_equal(one, many, "after the trajectory chunk")
    assert np.array_equal(_np(many.out["flags"]), _np(one.out["flags"]))
    m1, m2 = one.metrics(), many.metrics()
    for k in m1:
        assert m1[k] == m2[k], (k, m1[k], m2[k])
    assert m1["ctl_steps"] > 0
    stats = dict(m1, max_alive=int(max(one.read_env(e).n_alive for e in range(n_envs))))
    assert many.step_many(0, source=source) is not None          # zero ticks: a no-op
    batches_equal(one, many, "after a zero-tick call")
    # misuse
    bad = make_batch(arr, n_envs, capacity, backend, outputs=("obs_post", "obs_pre", "state_pre", "flags"), **cfg)
    bad.reset()
    for fn in (lambda: bad.step_many(2, source="zero"), lambda: one.step_many(2, source="pool"),
               lambda: many.step_many(2, source="nope")):
        try:
            fn()
            raise AssertionError("misuse accepted")
        except PveError:
            pass
    return stats


def check_step_many_geo(backend, lane_num, n_envs=5, capacity=128, seed=75, chunks=(1, 7, 40, 3, 60), rate=None,
                        trajectory_chunk=12, quantize=None, source="pool", persistent=False):
    """pve_step_many for the 4- / 8-lane layouts (k_rollout_geo: the general-geometry tick resident on the chip) == single
    pve_step_all ticks of k_tick_geo, bit for bit: persistent state, headers (incl. the spawn counter intention_re and
    the stale list heads), observation rows, last-tick outputs and, in trajectory mode, every tick's outputs."""
    from pve_mcc_amd.arrivals import synthetic_intentions
    n_pool = 5
    rng = np.random.default_rng(seed)
    rate = rate or {12: 1100.0, 8: 1500.0, 4: (1800.0 if capacity == 128 else 1200.0)}[lane_num]
    total = sum(chunks) + trajectory_chunk
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=total * 0.1 + 30, seed=seed, lane_num=lane_num)
    ch = synthetic_intentions(n_envs, arr.shape[1], seed=seed, lane_num=lane_num) if lane_num == 8 else None
    outs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out", "lanej")
    kw = dict(lane_num=lane_num, intentions=ch, outputs=outs)
    if lane_num == 12:
        kw["general_path"] = True
    one = make_batch(arr, n_envs, capacity, backend, **kw)
    many = make_batch(arr, n_envs, capacity, backend, **kw)
    one.reset(); many.reset()
    acts = rng.uniform(-3, 3, size=(n_pool, n_envs, capacity))
    if quantize:
        acts = np.round(acts / quantize) * quantize
    pool = torch.as_tensor(acts).to(one.device)
    many.set_action_pool(pool)
    if source == "table":                                   # actions by (tick, vehicle id), gathered inside the resident kernel
        tv = rng.uniform(-3, 3, size=(23, 150))
        if quantize:
            tv = np.round(tv / quantize) * quantize
        table = torch.as_tensor(tv)
        one.set_action_table(table); many.set_action_table(table)

    def single():
        if source == "table":
            return one.step(one.actions_from_table())
        return one.step(pool[one.ticks % n_pool])

    def same_outputs(o1, o2, what):
        f = _np(o1["flags"])
        assert np.array_equal(f, _np(o2["flags"])), what + ": flags"
        alive, ctl = (f & 1) != 0, (f & 2) != 0
        for k in ("reward", "new_slot", "lanej"):
            assert np.array_equal(_np(o1[k])[alive], _np(o2[k])[alive]), what + ": " + k
        assert np.array_equal(_np(o1["nbr"])[ctl], _np(o2["nbr"])[ctl]), what + ": nbr"
        assert np.array_equal(_np(o1["env_out"]), _np(o2["env_out"])), what + ": env_out"

    def same_headers(what):
        for e in range(n_envs):
            i1, i2 = one.read_env(e), many.read_env(e)
            for f in ("current_time", "n_alive", "id_seq", "passed_veh", "passed_veh_step_total", "overflow", "intention_re"):
                assert getattr(i1, f) == getattr(i2, f), "%s: header %s of env %d" % (what, f, e)
            for f in ("lane_count", "veh_rec", "head_valid", "head_lane", "head_j"):
                assert list(getattr(i1, f)) == list(getattr(i2, f)), "%s: header %s of env %d" % (what, f, e)

    for n in chunks:
        for _ in range(n):
            o1 = single()
        ch = 0 if n < 6 else (n // 3 + 1)
        o2 = many.step_many(n, source=source, chunk=ch, persistent=persistent)
        want = "persistent" if (persistent and ch > 0 and n > ch) else "resident"
        assert many.last_launch() == want, (many.last_launch(), want, source, n)
        one.synchronize(); many.synchronize()
        batches_equal(one, many, "lane_num %d, chunk of %d" % (lane_num, n))
        same_outputs(o1, o2, "lane_num %d, chunk of %d" % (lane_num, n))
        same_headers("lane_num %d, chunk of %d" % (lane_num, n))
    traj = many.step_many(trajectory_chunk, source=source, trajectory=True, chunk=trajectory_chunk // 2 + 1, persistent=persistent)
    for k in range(trajectory_chunk):
        o1 = single()
        same_outputs(o1, {n: traj[n][k] for n in traj}, "trajectory tick %d" % k)
        post_ctl = (_np(one.state_field("meta")) & 1) != 0
        assert np.array_equal(_np(one.obs)[post_ctl], _np(traj["obs_post"][k])[post_ctl]), "trajectory obs, tick %d" % k
    batches_equal(one, many, "after the trajectory chunk")
    same_headers("after the trajectory chunk")
    m1, m2 = one.metrics(), many.metrics()
    for k in m1:
        assert m1[k] == m2[k], (k, m1[k], m2[k])
    assert m1["ctl_steps"] > 0
    return m1


def check_step_many_geo_actor(backend, lane_num, n_envs=6, capacity=128, seed=77, chunks=(1, 9, 30, 4, 45), rate=None,
                              trajectory_chunk=11, obs_dtype=torch.float64, persistent=False, oracle_ticks=0, strict=True):
    """The closed loop for the 4- / 8-lane layouts (main.py:398-441 drives every lane_num; the shipped checkpoint's args.txt
    records lane_num = 4): pve_step_many(PVE_SRC_ACTOR) -- the actor inside k_rollout_geo<.., ACT[, PERS]> -- == step_with_actor
    ticks (actor launch + k_tick_geo), bit for bit: persistent state, headers, observation rows, last-tick outputs and, in
    trajectory mode, every tick's outputs.  oracle_ticks > 0: first, for that many ticks, the two-launch form itself against
    the sequential general-geometry oracle, which is handed the actions the device actor produced (rewards + state)."""
    from oracle.actor_np import flat_weights, load_weights
    from pve_mcc_amd.arrivals import synthetic_intentions
    rate = rate or {8: 1300.0, 4: (1500.0 if capacity == 128 else 1000.0)}[lane_num]
    total = sum(chunks) + trajectory_chunk + oracle_ticks
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=total * 0.1 + 30, seed=seed, lane_num=lane_num)
    ch = synthetic_intentions(n_envs, arr.shape[1], seed=seed, lane_num=lane_num) if lane_num == 8 else None
    outs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out", "lanej")
    kw = dict(lane_num=lane_num, intentions=ch, outputs=outs, obs_dtype=obs_dtype)
    one = make_batch(arr, n_envs, capacity, backend, **kw)
    many = make_batch(arr, n_envs, capacity, backend, **kw)
    w = flat_weights(load_weights())
    for b in (one, many):
        b.reset()
        b.set_actor(w)
    if oracle_ticks:
        from oracle.oracle_geo import OracleGeoEnv
        oracles = [OracleGeoEnv(arr[e], lane_num, choice=None if ch is None else ch[e]) for e in range(n_envs)]
        for t in range(oracle_ticks):
            acts = _np(one.act()).copy()
            out = one.step_with_actor()
            many.step_with_actor()
            rew, flags, lanej = _np(out["reward"]), _np(out["flags"]).astype(np.int64), _np(out["lanej"]).astype(np.int64)
            for e, o in enumerate(oracles):
                n = o.n_alive
                _vid, ctlm, _ = o.alive_view()
                assert np.all(acts[e, :n][ctlm == 0] == 0) and np.all(acts[e, n:] == 0), "uncontrolled slots get 0 (main.py:401)"
                rec = o.tick(acts[e, :n])
                f = flags[e, :n]
                order = np.lexsort((lanej[e, :n] & 0xFFFF, (f >> 6) & 3, lanej[e, :n] >> 16))
                ctl = order[((f & 2) != 0)[order]]
                assert len(ctl) == len(rec["ids"]), "controlled set: tick %d env %d" % (t, e)
                assert close(rec["reward"], rew[e, ctl], 1e-9), "reward: tick %d env %d" % (t, e)
        for e, o in enumerate(oracles):
            info, vi, vf = state_snapshot(one, e)
            ovi, ovf, _, _ = o.vehicles()
            assert np.array_equal(vi[:, :13], ovi[:, :13]), "state ints after the oracle ticks, env %d" % e
            assert close(ovf[:, :5], vf[:, :5], 1e-9), "state floats after the oracle ticks, env %d" % e

    def same_outputs(o1, o2, what):
        f = _np(o1["flags"])
        assert np.array_equal(f, _np(o2["flags"])), what + ": flags"
        alive, ctl = (f & 1) != 0, (f & 2) != 0
        for k in ("reward", "new_slot", "lanej"):
            assert np.array_equal(_np(o1[k])[alive], _np(o2[k])[alive]), what + ": " + k
        assert np.array_equal(_np(o1["nbr"])[ctl], _np(o2["nbr"])[ctl]), what + ": nbr"
        assert np.array_equal(_np(o1["env_out"]), _np(o2["env_out"])), what + ": env_out"

    def same_rows(what):
        post_ctl = (_np(one.state_field("meta")) & 0x81) == 0x81
        assert np.array_equal(_np(one.obs)[post_ctl], _np(many.obs)[post_ctl]), what + ": observation rows"

    n_ctl = 0
    for n in chunks:
        for _ in range(n):
            o1 = one.step_with_actor()
        ch_t = 0 if n < 6 else (n // 3 + 1)
        o2 = many.step_many(n, actor=True, chunk=ch_t, persistent=persistent)
        want = "tick" if backend == "emu" else ("persistent" if (persistent and ch_t > 0 and n > ch_t) else "resident")
        assert many.last_launch() == want, (many.last_launch(), want, n)
        one.synchronize(); many.synchronize()
        what = "lane_num %d closed loop, call of %d" % (lane_num, n)
        batches_equal(one, many, what)
        same_outputs(o1, o2, what)
        same_rows(what)
        n_ctl += int(_np(o1["env_out"])[:, 1].sum())
    traj = many.step_many(trajectory_chunk, actor=True, trajectory=True, chunk=trajectory_chunk // 2 + 1, persistent=persistent)
    for k in range(trajectory_chunk):
        o1 = one.step_with_actor()
        same_outputs(o1, {n: traj[n][k] for n in traj}, "trajectory tick %d" % k)
        post_ctl = (_np(one.state_field("meta")) & 0x81) == 0x81
        assert np.array_equal(_np(one.obs)[post_ctl], _np(traj["obs_post"][k])[post_ctl]), "trajectory obs, tick %d" % k
    batches_equal(one, many, "after the trajectory chunk")
    m1, m2 = one.metrics(), many.metrics()
    for k in m1:
        assert m1[k] == m2[k], (k, m1[k], m2[k])
    if strict:                                            # (a randomised soak may draw an empty or an overfull scene: both forms agree there too)
        assert n_ctl >= 3 * len(chunks) * n_envs and m1["overflow"] == 0, (n_ctl, m1)
    return m1


def check_step_many_state_rows(backend, n_envs=3, capacity=128, rate=None, calls=(40, 25, 60, 35), seed=81, n_pool=7,
                               obs_dtype=torch.float64, chunk=0, source="pool", min_ctl_per_tick=5, lane_num=12, persistent=False):
    """Training outputs on the fast path (lane_num 4 / 8: f3 x f4):
    pve_step_many trajectory roll-outs with state_pre -- the 7 x 28 states with fresh / stale neighbour rows (ref
    :1325-1337) and the 7-action vectors (column 2, ref :290) -- compared with the oracle at EVERY tick of every env (ids,
    neighbours, rewards, row 0, full state), across call boundaries (the first tick of a call reads the rows the previous
    call stored) and chunked launches.  float32 rows: the same within float32 round-off of the stored rows.  The 4- / 8-lane
    layouts process (and list) the vehicles in (lane, intention, j) order (ref :233-275).
    persistent=True (round 5; lane_num 12): the trainer's roll-out through the work queue -- k_rollout<.., TRAIN, PERS>: the stale
    rows of an item's first tick are what the previous item of the intersection (another workgroup) stored.
    source="table": actions by (tick, vehicle id) (k_rollout<.., TRAIN, IDT>)."""
    from pve_mcc_amd import _capi
    from pve_mcc_amd.arrivals import synthetic_intentions
    rng = np.random.default_rng(seed)
    total = sum(calls)
    rate = rate or {12: 1100.0, 8: 1500.0, 4: (1800.0 if capacity == 128 else 1200.0)}[lane_num]
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=total * 0.1 + 30, seed=seed, lane_num=lane_num)
    ch = synthetic_intentions(n_envs, arr.shape[1], seed=seed, lane_num=lane_num) if lane_num == 8 else None
    outs = ("obs_post", "obs_pre", "state_pre", "reward", "flags", "nbr", "lanej", "env_out", "new_slot")
    kw = dict(lane_num=lane_num, intentions=ch) if lane_num != 12 else {}
    b = make_batch(arr, n_envs, capacity, backend, outputs=outs, obs_dtype=obs_dtype, **kw)
    b.reset()
    pool_np = rng.uniform(-3, 3, size=(n_pool, n_envs, capacity)).astype(np.float32).astype(np.float64)
    if source == "pool":
        b.set_action_pool(torch.as_tensor(pool_np))
    table_np = rng.uniform(-3, 3, size=(19, 140)).astype(np.float32).astype(np.float64)      # (few columns: later ids share the last)
    if source == "table":
        b.set_action_table(torch.as_tensor(table_np))
    if lane_num == 12:
        oracles = [OracleEnv(arr[e]) for e in range(n_envs)]
    else:
        from oracle.oracle_geo import OracleGeoEnv
        oracles = [OracleGeoEnv(arr[e], lane_num, choice=None if ch is None else ch[e]) for e in range(n_envs)]
    tol = 1e-9 if obs_dtype == torch.float64 else 2e-6          # float32 rows: relative round-off of the stored values
    t, n_ctl_total = 0, 0
    ring = [b.alloc_trajectory(max(calls)) for _ in range(2)]
    for ci, n in enumerate(calls):
        traj = b.step_many(n, source=source, trajectory=ring[ci & 1] if ci else True, chunk=chunk, persistent=persistent)
        b.synchronize()
        if persistent:      # (round 6: lane_num 4 through the queue too)
            assert b.last_launch() == ("persistent" if 0 < chunk < n else "resident"), (b.last_launch(), chunk, n)
        host = {x: _np(traj[x][:n]) for x in ("flags", "reward", "nbr", "lanej", "obs_pre", "state_pre", "env_out")}
        for k in range(n):
            for e, o in enumerate(oracles):
                vid, ctlm, _ = o.alive_view()
                na = o.n_alive
                if source == "pool":
                    acts = np.where(ctlm != 0, pool_np[(t + k) % n_pool, e, :na], 0.0)
                elif source == "table":
                    acts = np.where(ctlm != 0, table_np[(t + k) % table_np.shape[0], np.minimum(vid, table_np.shape[1] - 1)], 0.0)
                else:
                    acts = np.zeros(na)
                rec = o.tick(acts, want_state=True)
                f = host["flags"][k, e, :na].astype(np.int64)
                lja = host["lanej"][k, e, :na].astype(np.int64)
                # controlled slots in PROCESSING order: (lane, intention, j); == slot order for lane_num 12
                order = np.lexsort((lja & 0xFFFF, (f >> _capi.F_INTENT_SHIFT) & 3, lja >> 16)) if na else np.zeros(0, np.int64)
                ctl = order[((f & 2) != 0)[order]]
                assert int(host["env_out"][k, e, 0]) == na and len(ctl) == len(rec["ids"]), "controlled set: tick %d env %d" % (t + k, e)
                lj = lja[ctl]
                assert np.array_equal(np.stack([lj >> 16, lj & 0xFFFF], -1), rec["ids"]), "ids: tick %d env %d" % (t + k, e)
                nb = host["nbr"][k, e, :na][ctl].astype(np.int64)
                nb = np.stack([np.where(nb < 0, -1, nb >> 16), np.where(nb < 0, -1, nb & 0xFFFF)], -1)
                assert np.array_equal(nb, rec["nbr"]), "neighbours: tick %d env %d" % (t + k, e)
                assert close(rec["reward"], host["reward"][k, e, :na][ctl], 1e-9), "reward: tick %d env %d" % (t + k, e)
                assert close(rec["obs0"], host["obs_pre"][k, e, :na][ctl].astype(np.float64), tol), "row 0: tick %d env %d" % (t + k, e)
                st = host["state_pre"][k, e, :na][ctl].astype(np.float64)
                assert close(rec["state"], st, tol), "7 x 28 state: tick %d env %d" % (t + k, e)
                assert close(rec["act7"], st[:, :, 2], tol), "7-action vector: tick %d env %d" % (t + k, e)
                n_ctl_total += len(rec["ids"])
        t += n
    assert n_ctl_total >= min_ctl_per_tick * total            # (the scenario did exercise controlled vehicles)
    assert b.metrics()["overflow"] == 0
    return n_ctl_total


def check_closed_loop_state_rows(backend, n_envs=6, capacity=128, rate=1000.0, calls=(30, 17, 40), chunk=7, seed=87,
                                 obs_dtype=torch.float32, persistent=False, lane_num=12, want_launch=None):
    """The trainer's closed-loop roll-out: pve_step_many(PVE_SRC_ACTOR, trajectory = 1) with the training outputs (obs_pre,
    state_pre: k_rollout<.., ACT, TRAIN[, PERS]>) == step_with_actor ticks (actor launch + k_tick with its own STATE phase), bit
    for bit: rows, 7 x 28 states, rewards, flags of every tick, and the persistent state after every call."""
    from oracle.actor_np import flat_weights, load_weights
    from pve_mcc_amd.arrivals import synthetic_intentions
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=sum(calls) * 0.1 + 30, seed=seed, lane_num=lane_num)
    geo = dict(lane_num=lane_num, intentions=synthetic_intentions(n_envs, arr.shape[1], seed=seed, lane_num=lane_num) if lane_num == 8 else None) \
        if lane_num != 12 else {}
    outs = ("obs_post", "obs_pre", "state_pre", "reward", "flags", "nbr", "new_slot", "env_out")
    one = make_batch(arr, n_envs, capacity, backend, outputs=outs, obs_dtype=obs_dtype, **geo)
    many = make_batch(arr, n_envs, capacity, backend, outputs=outs, obs_dtype=obs_dtype, **geo)
    w = flat_weights(load_weights())
    n_ctl = 0
    for b in (one, many):
        b.reset()
        b.set_actor(w)
    for n in calls:
        traj = many.step_many(n, actor=True, trajectory=True, chunk=chunk, persistent=persistent)
        many.synchronize()
        if want_launch is not None and backend != "emu":      # (lane_num 4 / 8, round 6: the resident kernel, no longer two launches per tick)
            assert many.last_launch() in want_launch, (many.last_launch(), want_launch)
        elif persistent and backend != "emu":
            assert many.last_launch() == ("persistent" if 0 < chunk < n else "resident"), many.last_launch()
        for k in range(n):
            o = one.step_with_actor()
            f = _np(o["flags"])
            assert np.array_equal(f, _np(traj["flags"][k])), "flags, tick %d" % k
            ctl = (f & 2) != 0
            n_ctl += int(ctl.sum())
            for x in ("reward", "obs_pre", "state_pre", "nbr"):
                assert np.array_equal(_np(o[x])[ctl], _np(traj[x][k])[ctl]), "%s, tick %d of a call of %d" % (x, k, n)
        batches_equal(one, many, "closed-loop training roll-out, call of %d" % n)
    assert n_ctl >= sum(calls)
    return n_ctl


def check_step_many_pipelined(backend, n_envs=5, n_sub=2, capacity=128, ticks=40, seed=73):
    """PipelinedIntersections.step_many (one call per sub-batch, each on its own stream) == one batch stepped tick by tick."""
    from pve_mcc_amd.batched import PipelinedIntersections
    from tests.hip_adapter import emulator_lib
    rng = np.random.default_rng(seed)
    arr = synthetic_arrivals(n_envs, rate=1100.0, horizon_s=ticks * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "env_out", "new_slot")
    pool = torch.as_tensor(rng.uniform(-3, 3, size=(4, n_envs, capacity)))
    one = make_batch(arr, n_envs, capacity, backend, outputs=outs)
    kw = dict(device="cpu", _lib=emulator_lib()) if backend == "emu" else dict(device="cuda")
    pipe = PipelinedIntersections(n_envs, capacity, arr, n_sub=n_sub, outputs=outs, **kw)
    one.reset(); pipe.reset()
    pipe.set_action_pool(pool)
    pool_d = pool.to(one.device)
    for t in range(ticks):
        one.step(pool_d[t % 4])
    pipe.step_many(ticks // 2)
    pipe.step_many(ticks - ticks // 2, chunk=7)
    pipe.synchronize(); one.synchronize()
    # live slots only: a multi-tick call rewrites the slots that hold a vehicle when it ends, single ticks leave the
    # values of vehicles deleted earlier behind in the (unused) slots above n_alive
    meta = _np(one.state_field("meta"))
    assert np.array_equal(meta, np.concatenate([_np(sub.state_field("meta")) for sub in pipe.subs], 0))
    live = meta != 0
    assert live.sum() > 10 * n_envs
    for k in STATE_F + STATE_I:
        x = _np(one.state_field(k))
        y = np.concatenate([_np(sub.state_field(k)) for sub in pipe.subs], 0)
        assert np.array_equal(x[live], y[live]), "state %s differs" % k
    ctl = (meta & 1) != 0
    assert np.array_equal(_np(one.obs)[ctl], np.concatenate([_np(sub.obs) for sub in pipe.subs], 0)[ctl])
    m1, m2 = one.metrics(), pipe.metrics()
    for k in ("alive_steps", "ctl_steps", "spawned", "passed", "collided", "locks", "passed_steps", "ticks"):
        assert m1[k] == m2[k], (k, m1[k], m2[k])
    # trajectory roll-outs of the sub-batches: buffers allocated, filled and copied back on each sub-batch's OWN stream
    # (a zero-fill / copy-back on torch's current stream is not ordered with the kernels), twice in a row so
    # that the second call's allocations can recycle the first call's memory
    for rep in range(2):
        n = 9
        trajs = pipe.step_many(n, trajectory=True, chunk=4)
        for k in range(n):
            o1 = one.step(pool_d[one.ticks % 4])
            f1 = _np(o1["flags"])
            f2 = np.concatenate([_np(tr["flags"][k]) for tr in trajs], 0)
            assert np.array_equal(f1, f2), "pipelined trajectory, flags of tick %d" % k
            alive = (f1 & 1) != 0
            r2 = np.concatenate([_np(tr["reward"][k]) for tr in trajs], 0)
            assert np.array_equal(_np(o1["reward"])[alive], r2[alive]), "pipelined trajectory, reward of tick %d" % k
            post_ctl = (_np(one.state_field("meta")) & 1) != 0
            ob2 = np.concatenate([_np(tr["obs_post"][k]) for tr in trajs], 0)
            assert np.array_equal(_np(one.obs)[post_ctl], ob2[post_ctl]), "pipelined trajectory, rows of tick %d" % k
        # the handles' single-tick views show the last tick (copied back on the sub-batch stream)
        pipe.synchronize()
        post_ctl = (_np(one.state_field("meta")) & 1) != 0
        assert np.array_equal(_np(one.obs)[post_ctl], np.concatenate([_np(sub.obs) for sub in pipe.subs], 0)[post_ctl])
        assert np.array_equal(_np(one.out["flags"]), np.concatenate([_np(sub.out["flags"]) for sub in pipe.subs], 0))


def check_full_size_vs_oracle(backend, n_envs, capacity, rate, ticks=420, n_sample=16, seed=20250213, many=0,
                              n_pool=8):
    """BASELINE-size batch (thousands of envs) past the 300-tick fill: `n_sample` envs spread over the batch are each
    shadowed by their own CPU oracle on the same arrival stream and the same slot-indexed action tape, compared EVERY
    tick (controlled set, rewards, collision / lock counters, neighbour ids, observation rows) and field by field at the
    end; no deferred spawn anywhere in the batch.  many > 0: the batch is advanced by pve_step_many in chunks of `many`
    ticks (trajectory outputs) instead of one pve_step_all per tick."""
    rng = np.random.default_rng(seed)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "nbr", "env_out", "new_slot", "lanej")
    b = make_batch(arr, n_envs, capacity, backend, outputs=outs)
    b.reset()
    sample = sorted(set(np.linspace(0, n_envs - 1, n_sample).astype(int).tolist()))
    idx = torch.as_tensor(sample, device=b.device)
    oracles = {e: OracleEnv(arr[e]) for e in sample}
    pool_np = rng.uniform(-3, 3, size=(n_pool, n_envs, capacity)).astype(np.float32).astype(np.float64)
    pool = torch.as_tensor(pool_np).to(b.device)
    if many:
        b.set_action_pool(pool)
    peak = 0

    def compare(t, rew, flags, eo, nbr, obs_post, new_slot):
        nonlocal peak
        for q, e in enumerate(sample):
            o = oracles[e]
            n = o.n_alive
            peak = max(peak, n)
            _vid, ctlm, _ = o.alive_view()
            rec = o.tick(np.where(ctlm != 0, pool_np[t % n_pool, e, :n], 0.0))
            ctl = (flags[q, :n] & 2) != 0
            assert int(eo[q, 0]) == n and int(ctl.sum()) == len(rec["ids"]), "controlled set: tick %d env %d" % (t, e)
            assert int(eo[q, 2]) == rec["collisions"] and int(eo[q, 3]) == rec["lock"], "counters: tick %d env %d" % (t, e)
            assert np.array_equal(flags[q, :n][ctl] >> 8, rec["coll_pv"]), "coll_pv: tick %d env %d" % (t, e)
            nb = nbr[q, :n][ctl].astype(np.int64)
            nb = np.stack([np.where(nb < 0, -1, nb >> 16), np.where(nb < 0, -1, nb & 0xFFFF)], -1)
            assert np.array_equal(nb, rec["nbr"]), "neighbours: tick %d env %d" % (t, e)
            assert close(rec["reward"], rew[q, :n][ctl], 1e-9), "reward: tick %d env %d" % (t, e)
            ns = new_slot[q, :n][ctl]
            kept = ns >= 0                                      # rows are stored at the post-compaction slot
            assert close(rec["obs0"][kept], obs_post[q][ns[kept]], 1e-9), "obs rows: tick %d env %d" % (t, e)

    t = 0
    while t < ticks:
        if many:
            n = min(many, ticks - t)
            traj = b.step_many(n, source="pool", trajectory=True)
            host = {k: _np(traj[k].index_select(1, idx)) for k in ("reward", "flags", "env_out", "nbr", "obs_post", "new_slot")}
            for k in range(n):
                compare(t + k, *(host[x][k] for x in ("reward", "flags", "env_out", "nbr", "obs_post", "new_slot")))
            t += n
        else:
            out = b.step(pool[t % n_pool])
            compare(t, *(_np(out[x].index_select(0, idx)) for x in ("reward", "flags", "env_out", "nbr")),
                    _np(b.obs.index_select(0, idx)), _np(out["new_slot"].index_select(0, idx)))
            t += 1
    for e in sample:
        info, vi, vf = state_snapshot(b, e)
        ovi, ovf, _ = oracles[e].vehicles()
        assert np.array_equal(vi[:, :13], ovi[:, :13]), "final state ints, env %d" % e
        assert close(ovf[:, :5], vf[:, :5], 1e-9), "final state floats, env %d" % e
    m = b.metrics()
    assert m["ticks"] == ticks * n_envs
    assert m["overflow"] == 0, "deferred spawns at %g veh/h/lane x %d slots (peak of the sampled envs: %d)" % (rate, capacity, peak)
    return m, peak


def check_driver_shape_vs_oracle(backend, n_envs=4096, n_sub=2, capacity=128, rate=1100.0, chunk=5, n_sample=16,
                                 calls=(50, 50, 50, 50, 50, 50, 5, 20), n_pool=16, seed=20250213, trajectory=False, persistent=False,
                                 table=False):
    """Exactly the launch shape the driver's `bench.py --steps 20 --warmup 5` runs:
    PipelinedIntersections, `n_sub` sub-batches on their own streams, pve_step_many calls of 50 (prefill) / 5 (warm-up) /
    20 (timed) ticks split into launches of `chunk` ticks, action pool of 16 slot-indexed entries.  `n_sample` envs spread
    over the batch are shadowed by their own oracles: after EVERY call the last tick's outputs (controlled set, rewards,
    counters, neighbour ids, observation rows) and at the end the state field by field; overflow == 0 over the whole
    batch.  trajectory=True: the calls write into a ring of two trajectory buffers (bench.py --trajectory 1) and EVERY
    tick's outputs are compared.  table=True (round 5: the bench headline): BASELINE.md 3's tape a = float32(sin(0.37 id + 0.05
    tick)) by vehicle id through PVE_SRC_TABLE instead of the slot-indexed pool."""
    from pve_mcc_amd.batched import PipelinedIntersections
    from tests.hip_adapter import emulator_lib
    rng = np.random.default_rng(seed)
    total = sum(calls)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=total * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")
    kw = dict(device="cpu", _lib=emulator_lib()) if backend == "emu" else dict(device="cuda")
    pipe = PipelinedIntersections(n_envs, capacity, arr, n_sub=n_sub, outputs=outs, **kw)
    pipe.reset()
    pool_np = rng.uniform(-3, 3, size=(n_pool, n_envs, capacity)).astype(np.float32).astype(np.float64)
    src = None
    if table:
        cols = int(12 * (total * 0.1 + 4)) + 64
        tab_np = np.sin(0.37 * np.arange(cols, dtype=np.float64)[None, :] + 0.05 * np.arange(total + 8, dtype=np.float64)[:, None])
        tab_np = tab_np.astype(np.float32).astype(np.float64)
        pipe.set_action_table(torch.as_tensor(tab_np))
        src = "table"
    else:
        pipe.set_action_pool(torch.as_tensor(pool_np))
    sample = sorted(set(np.linspace(0, n_envs - 1, n_sample).astype(int).tolist()))
    oracles = {e: OracleEnv(arr[e]) for e in sample}
    ring = [pipe.alloc_trajectory(max(calls)) for _ in range(2)] if trajectory else None
    t, peak = 0, 0

    def oracle_actions(e, o, tk):
        vid, ctlm, _ = o.alive_view()
        if table:
            return np.where(ctlm != 0, tab_np[tk % tab_np.shape[0], np.minimum(vid, tab_np.shape[1] - 1)], 0.0)
        return np.where(ctlm != 0, pool_np[tk % n_pool, e, :len(vid)], 0.0)

    def compare_tick(e, o, tk, flags, rew, eo, nbr, new_slot, obs_post):
        nonlocal peak
        n = o.n_alive
        peak = max(peak, n)
        rec = o.tick(oracle_actions(e, o, tk))
        ctl = (flags[:n] & 2) != 0
        assert int(eo[0]) == n and int(ctl.sum()) == len(rec["ids"]), "controlled set: tick %d env %d" % (tk, e)
        assert int(eo[2]) == rec["collisions"] and int(eo[3]) == rec["lock"], "counters: tick %d env %d" % (tk, e)
        nb = nbr[:n][ctl].astype(np.int64)
        nb = np.stack([np.where(nb < 0, -1, nb >> 16), np.where(nb < 0, -1, nb & 0xFFFF)], -1)
        assert np.array_equal(nb, rec["nbr"]), "neighbours: tick %d env %d" % (tk, e)
        assert close(rec["reward"], rew[:n][ctl], 1e-9), "reward: tick %d env %d" % (tk, e)
        ns = new_slot[:n][ctl]
        kept = ns >= 0
        assert close(rec["obs0"][kept], obs_post[ns[kept]], 1e-9), "obs rows: tick %d env %d" % (tk, e)

    for ci, n in enumerate(calls):
        if trajectory:
            trajs = pipe.step_many(n, source=src, trajectory=ring[ci & 1], chunk=chunk, update_views=False, persistent=persistent)
        else:
            pipe.step_many(n, source=src, chunk=chunk, persistent=persistent)
        pipe.synchronize()
        for e in sample:
            k, le = pipe.sub_of(e)
            o = oracles[e]
            if trajectory:
                tr = trajs[k]
                host = {x: _np(tr[x][:n, le]) for x in ("flags", "reward", "env_out", "nbr", "new_slot", "obs_post")}
                for q in range(n):
                    compare_tick(e, o, t + q, *(host[x][q] for x in ("flags", "reward", "env_out", "nbr", "new_slot", "obs_post")))
            else:
                sub = pipe.subs[k]
                if n > 1 and not table:
                    o.run_pool(n - 1, pool_np[:, e, :], t)
                elif n > 1:
                    for q in range(n - 1):
                        o.tick(oracle_actions(e, o, t + q))
                compare_tick(e, o, t + n - 1, *(_np(sub.out[x][le]) for x in ("flags", "reward", "env_out", "nbr", "new_slot")),
                             _np(sub.obs[le]))
        t += n
    for e in sample:
        k, le = pipe.sub_of(e)
        info, vi, vf = state_snapshot(pipe.subs[k], le)
        ovi, ovf, _ = oracles[e].vehicles()
        assert np.array_equal(vi[:, :13], ovi[:, :13]), "final state ints, env %d" % e
        assert close(ovf[:, :5], vf[:, :5], 1e-9), "final state floats, env %d" % e
    m = pipe.metrics()
    assert m["ticks"] == total * n_envs and m["overflow"] == 0, (m["ticks"], m["overflow"], peak)
    return m, peak


def check_closed_loop_rollout_vs_two_launch(backend, n_envs=4096, n_sub=2, capacity=128, rate=1000.0, chunk=5, n_sample=16,
                                            calls=(50, 50, 50, 50, 50, 50, 5, 25), seed=4243, obs_dtype=torch.float64,
                                            persistent=False):
    """BASELINE config 5 at full size through the product's fast path: PipelinedIntersections,
    `n_sub` sub-batches on their own streams, pve_step_many(PVE_SRC_ACTOR) -- the actor INSIDE the resident kernel
    (k_rollout<.., ACT>) -- in launches of `chunk` ticks, against `n_sample` of the same arrival streams stepped as ONE
    small batch with step_with_actor (actor launch + tick launch per tick).  Both forms run the same actor_tile32 on
    the same float32 rows, so after every call the persistent state of the sampled envs (live slots, every field), the
    observation rows of their controlled vehicles and the last tick's outputs must be BIT-equal; overflow == 0 over the
    whole batch.  Also bounds the on-device actions against the NumPy restatement (action-level bar 5e-4) on the final
    rows."""
    from oracle.actor_np import actor_forward, flat_weights, load_weights
    from pve_mcc_amd.batched import PipelinedIntersections
    from tests.hip_adapter import emulator_lib
    total = sum(calls)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=total * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "env_out", "new_slot")
    kw = dict(device="cpu", _lib=emulator_lib()) if backend == "emu" else dict(device="cuda")
    sample = sorted(set(np.linspace(0, n_envs - 1, n_sample).astype(int).tolist()))
    big = PipelinedIntersections(n_envs, capacity, arr, n_sub=n_sub, outputs=outs, obs_dtype=obs_dtype, **kw)
    small = make_batch(arr[sample], len(sample), capacity, backend, outputs=outs, obs_dtype=obs_dtype)
    wd = load_weights()
    w = flat_weights(wd)
    for b in (big, small):
        b.reset()
        b.set_actor(w)
    t = 0
    for n in calls:
        big.step_many(n, actor=True, chunk=chunk, persistent=persistent)
        for _ in range(n):
            small.step_with_actor()
        big.synchronize(); small.synchronize()
        t += n
        for i, e in enumerate(sample):
            k, le = big.sub_of(e)
            sub = big.subs[k]
            live = _np(small.state_field("meta")[i]) != 0
            assert np.array_equal(_np(sub.state_field("meta")[le]) != 0, live), "alive set: env %d after %d ticks" % (e, t)
            for f in STATE_F + STATE_I:
                x, y = _np(sub.state_field(f)[le]), _np(small.state_field(f)[i])
                assert np.array_equal(x[live], y[live]), "%s: env %d after %d ticks" % (f, e, t)
            ctl = live & ((_np(small.state_field("meta")[i]) & 1) != 0)
            assert np.array_equal(_np(sub.obs[le])[ctl], _np(small.obs[i])[ctl]), "obs rows: env %d after %d ticks" % (e, t)
            for o in ("flags", "env_out"):
                assert np.array_equal(_np(sub.out[o][le]), _np(small.out[o][i])), "%s: env %d after %d ticks" % (o, e, t)
            fl = _np(small.out["flags"][i])
            was_ctl = (fl & 2) != 0
            assert np.array_equal(_np(sub.out["reward"][le])[was_ctl], _np(small.out["reward"][i])[was_ctl]), "reward: env %d" % e
    # action-level bar against the NumPy restatement on the rows the loop ended with
    a_dev = _np(small.act())
    meta = _np(small.state_field("meta"))
    ctl = (meta & 1) != 0
    a_np = actor_forward(wd, _np(small.obs)).astype(np.float64)
    worst = float(np.abs(a_dev - a_np)[ctl].max()) if ctl.any() else 0.0
    assert ctl.sum() > 10 * len(sample) * (capacity // 64) / 4 and worst <= 5e-4, (int(ctl.sum()), worst)
    m = big.metrics()
    assert m["ticks"] == total * n_envs and m["overflow"] == 0, (m["ticks"], m["overflow"])
    return m, worst


def check_geo_lists_equal_scan(backend, lane_num, n_envs=6, capacity=128, ticks=250, rate=None, seed=91, quantize=None):
    """General-geometry kernel: the per-route list path == its membership-scan fallback (PVE_CFG_GEO_SCAN), bit for bit,
    on random (optionally quantised: exact ties) action tapes; dense traffic makes some intersections overflow the list
    pool, so one batch mixes both paths."""
    from pve_mcc_amd.arrivals import synthetic_intentions
    rng = np.random.default_rng(seed)
    rate = rate or {12: 1100.0, 8: 1500.0, 4: 1800.0}[lane_num]
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 30, seed=seed, lane_num=lane_num)
    ch = synthetic_intentions(n_envs, arr.shape[1], seed=seed, lane_num=lane_num) if lane_num == 8 else None
    outs = ("obs_post", "obs_pre", "reward", "flags", "nbr", "env_out", "new_slot", "lanej")
    kw = dict(lane_num=lane_num, intentions=ch, outputs=outs)
    if lane_num == 12:
        kw["general_path"] = True
    bl = make_batch(arr, n_envs, capacity, backend, **kw)
    bs = make_batch(arr, n_envs, capacity, backend, geo_scan=True, **kw)
    bl.reset(); bs.reset()
    for t in range(ticks):
        acts = rng.uniform(-3, 3, size=(n_envs, capacity)).astype(np.float32).astype(np.float64)
        if quantize:
            acts = np.round(acts / quantize) * quantize
        a = torch.as_tensor(acts).to(bl.device)
        o1, o2 = bl.step(a), bs.step(a)
        f = _np(o1["flags"])
        assert np.array_equal(f, _np(o2["flags"])), "tick %d: flags" % t
        alive, c = (f & 1) != 0, (f & 2) != 0
        for k in ("reward", "lanej", "new_slot"):
            assert np.array_equal(_np(o1[k])[alive], _np(o2[k])[alive]), "tick %d: %s" % (t, k)
        assert np.array_equal(_np(o1["nbr"])[c], _np(o2["nbr"])[c]), "tick %d: nbr" % t
        assert np.array_equal(_np(o1["obs_pre"])[c], _np(o2["obs_pre"])[c]), "tick %d: obs_pre" % t
        assert np.array_equal(_np(o1["env_out"]), _np(o2["env_out"])), "tick %d: env_out" % t
        if t % 50 == 0 or t == ticks - 1:
            batches_equal(bl, bs, "tick %d" % t)
    return bl.metrics()
