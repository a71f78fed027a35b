"""Generate the golden vectors under tests/golden/ by importing the UNMODIFIED reference.

Run in the build container only:   python tests/golden/gen_golden.py
Outputs (committed): tests/golden/<case>.npz, one per (arrival stream, action tape), plus
tests/golden/geometry.npz (known answers of get_p / get_virtual_distance) and
tests/golden/streams/*.mat (the reference's own arrival-stream data files, MIT-licensed data).

Per case:
  meta        JSON: stream, policy, ticks, ctor kwargs, numpy/scipy versions
  arrive      float64 [R,12]  the rows of the stream the run can reach (+2)
  dig_i/dig_f per-tick digests (oracle/record.py: DIGEST_I_COLS / DIGEST_F_COLS)
  dense_ticks ticks with a full-precision dump; arrays named t<tick>_<field>
  state_ticks subset that also carries the full 7x28 state / 7-action rows
  guard_hits  how many ticks needed the :371-375 crash guard
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.record import digest, get_policy, DIGEST_I_COLS, DIGEST_F_COLS  # noqa: E402
from tests.golden import ref_harness as rh  # noqa: E402

CASES = [
    # name, stream, policy, ticks, ctor kwargs
    ("s1000_zero", "1000", "zero", 1000, {}),
    ("s1000_sin1", "1000", "sin1", 1000, {}),
    ("s200_sin1", "200", "sin1", 1500, {}),
    ("s1200_sin1", "1200", "sin1", 1000, {}),
    ("s1200_zero", "1200", "zero", 1000, {}),
    ("s400_sin2", "400", "sin2", 1500, {}),
    ("s1000_sin3", "1000", "sin3", 600, {}),
    ("s1000_sin1_vm6", "1000", "sin1", 600, {"vm": 6}),
    # every constructor argument the C ABI exposes (pve_config; ref :21-23) away from its default, together, under a
    # pseudo-random +-3 tape
    # the pretrained actor (the reference's own graph) driving the LIVE reference env, main.py:397-441
    ("s1000_actor", "1000", "actor_tape", 1000, {}),
    ("s1000_rand_kw", "1000", "rand3", 300, {"dis_ctl": 120, "lane_cw": 3, "collision_thr": 3, "vM": 15, "v0": 9, "am": -2.5,
                                             "aM": 2.5, "deltaT": 0.2, "vm": 6}),
]
DENSE_FIELDS = ("ids", "nbr", "reward", "obs0", "coll_pv", "deleted", "jerks", "veh_i", "veh_f",
                "heads", "veh_num", "veh_rec")


def graph_actor_policy():
    """SURVEY 8(c) policy (iii): the PRETRAINED actor -- the reference's own shipped graph (model_data/baseline/66.cptk.meta,
    decoded and evaluated op by op in float32 by tests/golden/gen_actor_golden.py:GraphActor) -- fed with veh["state"][0] of
    every controlled vehicle as main.py:398-406 does.  The ACTIONS are recorded as a tape; parity tests replay the tape."""
    from tests.golden.gen_actor_golden import GraphActor
    actor = GraphActor()

    def pol(tick, veh_id, control, obs0=None):
        a = np.zeros(len(veh_id), np.float64)
        c = np.asarray(control) != 0
        if c.any():
            a[c] = actor.run(np.asarray(obs0, np.float64)[c].astype(np.float32), np.float32).astype(np.float64).reshape(-1)
        return a
    return pol


def dense_tick_set(ticks):
    s = set(range(0, 6))
    s.update(t for t in range(ticks) if t % 125 == 124)
    s.add(ticks - 1)
    return sorted(s)


def gen_case(name, stream, pol, ticks, kw):
    arr = rh.load_stream(stream)
    policy = graph_actor_policy() if pol == "actor_tape" else get_policy(pol)
    ref = rh.RefRunner(arr, policy, want_state=True, **kw)
    tape_vals, tape_off = [], [0]                  # "actor_tape": the actions fed on every tick, alive vehicles in (lane, j) order
    agg = dict(alive_steps=0, ctl_steps=0, collided=0, locks=0, sum_reward=0.0, n_reward=0, sum_jerks=0.0)
    dense = dense_tick_set(ticks)
    state_ticks = [t for t in (3, ticks // 2, ticks - 1) if t in dense or True][:3]
    out = {}
    dig_i = np.zeros((ticks, len(DIGEST_I_COLS)), np.int64)
    dig_f = np.zeros((ticks, len(DIGEST_F_COLS)), np.float64)
    for t in range(ticks):
        rec = ref.tick()
        dig_i[t], dig_f[t] = digest(rec)
        tape_vals.append(np.asarray(ref.tape, np.float64)); tape_off.append(tape_off[-1] + len(ref.tape))
        agg["alive_steps"] += len(rec["veh_i"]); agg["ctl_steps"] += len(rec["ids"])
        agg["collided"] += int((rec["coll_pv"] > 0).sum()); agg["locks"] += int(rec["lock"])     # main.py:410-412, :413-415
        agg["sum_reward"] += float(np.sum(rec["reward"])); agg["n_reward"] += len(rec["reward"])
        agg["sum_jerks"] += float(np.sum(rec["jerks"]))
        if t in dense or t in state_ticks:
            for f in DENSE_FIELDS:
                out["t%d_%s" % (t, f)] = rec[f]
            out["t%d_scalars" % t] = np.array([rec["collisions"], rec["lock"], rec["id_seq"], rec["passed"],
                                               rec["passed_step_total"]], np.int64)
            out["t%d_time" % t] = np.array(rec["time"], np.float64)
            out["t%d_tape" % t] = np.asarray(ref.tape, np.float64)
        if t in state_ticks:
            out["t%d_state" % t] = rec["state"]
            out["t%d_act7" % t] = rec["act7"]
    rows = int(np.max(ref.env.veh_rec)) + 2
    out["arrive"] = arr[:rows].copy()
    out["dig_i"], out["dig_f"] = dig_i, dig_f
    out["dense_ticks"] = np.array(sorted(set(dense) | set(state_ticks)), np.int32)
    out["state_ticks"] = np.array(state_ticks, np.int32)
    out["guard_hits"] = np.array(ref.guard_hits, np.int32)
    if pol == "actor_tape":
        tv = np.concatenate(tape_vals)
        assert np.array_equal(tv.astype(np.float32).astype(np.float64), tv), "actor outputs are float32 values"
        out["tape_vals"] = tv.astype(np.float32)
        out["tape_off"] = np.asarray(tape_off, np.int64)
        # the aggregates main.py:test() prints (main.py:523-526), from the LIVE reference under its own policy
        agg.update(id_seq=int(rec["id_seq"]), passed=int(rec["passed"]), passed_step_total=int(rec["passed_step_total"]),
                   pT_m=float(rec["passed_step_total"] / (rec["passed"] + 1e-4) * 0.1),
                   reward_mean=agg["sum_reward"] / max(1, agg["n_reward"]), jerk_per_veh=agg["sum_jerks"] / max(1, int(rec["passed"])))
        out["aggregates"] = np.array(json.dumps(agg))
        print("   aggregates:", agg)
    import scipy
    out["meta"] = np.array(json.dumps(dict(name=name, stream=stream, policy=pol, ticks=ticks, ctor=kw,
                                           numpy=np.__version__, scipy=scipy.__version__,
                                           t_init=repr(float(dig_f[0, 0] - 0.1)))))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-16s ticks %4d  alive-steps %6d ctl-steps %6d id_seq %3d passed %3d guard %3d  -> %d KB" % (
        name, ticks, dig_i[:, 0].sum() , dig_i[:, 1].sum(), dig_i[-1, 2], dig_i[-1, 3], ref.guard_hits,
        os.path.getsize(path) // 1024))


def gen_geometry():
    """Known answers of the two geometry helpers, sampled through the reference object."""
    tis = rh.import_reference()
    arr = rh.load_stream("200")
    env = tis.TrafficInteraction(arr, 150, rh.default_args(), show_col=False, virtual_l=True, lane_num=12)
    ps = np.concatenate([np.linspace(-140, 170, 63), np.array([0.0, 1e-9, 3.926875, 27.488125, 30.0, 7.5, 22.5])])
    gp = np.zeros((12, len(ps), 2))
    for lane in range(12):
        for k, p in enumerate(ps):
            q = env.get_p(float(p), lane, lane % 3)
            gp[lane, k] = (q[0], q[1])
    vd = np.full((12, 12, len(ps)), np.nan)
    for ego in range(12):
        for other in env.lane2lane[ego]:
            for k, p in enumerate(ps):
                d, ch = env.get_virtual_distance(other, ego, float(p))
                if ch:
                    vd[ego, other, k] = d[0]
    consts = np.array([env.cita, env.alpha, env.beta, env.gama, env._gama], np.float64)
    li = np.array(env.lane_info, np.float64)
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), ps=ps, get_p=gp, vd=vd, consts=consts, lane_info=li)
    print("geometry.npz written")


def copy_streams():
    dst = os.path.join(HERE, "streams")
    os.makedirs(dst, exist_ok=True)
    for s in ("200", "1000"):
        src = os.path.join(rh.REF_DIR, "data", "test", "arvTimeNewVeh_new_%s_12.mat" % s)
        shutil.copyfile(src, os.path.join(dst, os.path.basename(src)))
        os.chmod(os.path.join(dst, os.path.basename(src)), 0o644)


if __name__ == "__main__":
    only = sys.argv[1:]
    gen_geometry()
    copy_streams()
    for c in CASES:
        if not only or c[0] in only:
            gen_case(*c)
